"""Streaming front end (pdt_stream_*): pushing a capture block by block yields, in order and exactly
once, the frames of one whole-capture demodulation -- the guarantee stated in include/pdt.h -- with every stage's state
carried from segment to segment (cost per push independent of the stream's length, bounded device window).  The expected
frames come from the ORACLE (CPU restatement of the reference's chunk loop), not from another run of the HIP path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def stream_all(d, iq, block):
    d.stream_begin()
    parts, when = [], []
    for i in range(0, len(iq), block):
        new = d.stream_push(iq[i:i + block])
        if len(new):
            parts.append(new)
            when.append(i + block)
    tail = d.stream_end()
    parts.append(tail)
    return np.concatenate(parts) if parts else tail, when, len(tail)


@pytest.mark.parametrize("block", [2400, 10000, 7777, 50001])
def test_clip_in_blocks_equals_one_shot(pdt, clip, block):
    rate, iq = clip
    with pdt.Demodulator(pdt.MODE_POES, rate) as ref:
        ref.demod(iq)
        want = ref.frames_array()
        want_text = ref.text()
    with pdt.Demodulator(pdt.MODE_POES, rate) as d:
        got, when, n_tail = stream_all(d, iq, block)
        assert got.tobytes() == want.tobytes()
        assert d.text() == want_text                      # after stream_end the context describes the whole capture
        if block <= 10000:
            assert len(when) >= 5 and n_tail <= 3         # frames arrive while the capture is still being pushed
        got2, _, _ = stream_all(d, iq, block)             # a context can stream again
        assert got2.tobytes() == want.tobytes()


def test_synthetic_minute_in_portaudio_blocks(pdt):
    """2 400-frame blocks as the reference's real-time twin reads them (POESTIPdemodPortAudio/main.c:326)."""
    fs = 50000
    iq = pdt.synth_capture(0, fs, 20.0, seed=17)
    with pdt.Demodulator(pdt.MODE_POES, fs) as ref:
        ref.demod(iq)
        want = ref.frames_array()
    with pdt.Demodulator(pdt.MODE_POES, fs) as d:
        got, when, n_tail = stream_all(d, iq, 2400)
    assert got.tobytes() == want.tobytes() and len(want) >= 198
    # latency: every frame is reported within ~3 chunks (0.6 s) of its end
    assert n_tail <= 4


def test_raw_float_and_argos_streams(pdt):
    fs = 50000
    iq = pdt.synth_capture(0, fs, 6.0, seed=3)
    raw = iq.astype(np.float32) / np.float32(32768.0)
    with pdt.Demodulator(pdt.MODE_POES, fs) as ref:
        ref.demod_raw(raw)
        want = ref.frames_array()
    with pdt.Demodulator(pdt.MODE_POES, fs) as d:
        got, _, _ = stream_all(d, raw, 4096)
        assert got.tobytes() == want.tobytes() and len(want) >= 55
        d.stream_begin()
        d.stream_push(raw[:100])
        with pytest.raises(pdt.PdtError):
            d.stream_push(iq[:100])                       # the sample format is fixed by the first push
    a = pdt.synth_capture(1, 32000, 12.0, f0_hz=130.0, seed=9)
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000) as ref:
        ref.demod(a)
        want = ref.frames_array()
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000) as d:
        got, _, _ = stream_all(d, a, 2400)
        assert got.tobytes() == want.tobytes() and len(want) >= 5
        with pytest.raises(pdt.PdtError):
            d.stream_begin().stream_push(np.zeros((10, 2), dtype=np.float32))   # ARGOS refuses RAW input


def test_empty_stream(pdt):
    with pdt.Demodulator(pdt.MODE_POES, 50000) as d:
        d.stream_begin()
        assert len(d.stream_end()) == 0 and d.stats().frames == 0


def oracle_frames(pdt, orc, mode, omode, fs, iq, **kw):
    o = orc.Oracle(omode, fs, iq, **kw)
    return o, o.text()


@pytest.mark.parametrize("blocks", [[2400], [10000], [1, 77, 9999, 30011], [250000], [12345, 3]])
def test_stream_against_the_oracle(pdt, orc, clip, blocks):
    """The reference's own capture pushed in various block patterns (cycled): text and totals equal the oracle's."""
    rate, iq = clip
    o = orc.Oracle(orc.POES, rate, iq)
    with pdt.Demodulator(pdt.MODE_POES, rate) as d:
        d.stream_begin()
        i = k = 0
        got = []
        while i < len(iq):
            b = blocks[k % len(blocks)]
            got.append(d.stream_push(iq[i:i + b]))
            i += b
            k += 1
        got.append(d.stream_end())
        fr = np.concatenate(got)
        assert pdt.format_frames(fr) == o.text() == d.text()
        s = d.stats()
        ns, nsym, nbits, nfr = o.totals()
        assert (s.samples, s.symbols, s.bits, s.frames) == (ns, nsym, nbits, nfr)
        assert s.lock_sample == o.lock_sample and f"{s.lock_freq_hz:0.2f}" == f"{o.lock_freq_hz:0.2f}"
        assert np.float32(s.norm_factor) == np.float32(o.norm_factor)


def test_long_stream_has_flat_cost_and_a_bounded_window(pdt, orc):
    """Four minutes at 50 ksps in sound-card blocks of 2 400 frames (5 000 pushes): identical to the oracle; the device
    window stops growing after the first second; the time per push does not grow with the position in the stream."""
    import time
    fs, secs = 50000, 240.0
    iq = pdt.synth_capture(0, fs, secs, seed=77)
    o = orc.Oracle(orc.POES, fs, iq, keep_stages=False)
    with pdt.Demodulator(pdt.MODE_POES, fs) as d:
        d.stream_begin()
        got, retained, dt = [], [], []
        for i in range(0, len(iq), 2400):
            t0 = time.perf_counter()
            got.append(d.stream_push(iq[i:i + 2400]))
            dt.append(time.perf_counter() - t0)
            retained.append(d.stream_retained())
        got.append(d.stream_end())
        assert pdt.format_frames(np.concatenate(got)) == o.text()
    assert max(retained) < 0.7 * fs + 200000 and max(retained[len(retained) // 2:]) <= max(retained[:len(retained) // 2])
    q = len(dt) // 4
    first, last = np.median(dt[q:2 * q]), np.median(dt[3 * q:])
    assert last < 1.5 * first, (first, last)


def test_lock_in_a_later_segment_weak_signal_and_mm(pdt, orc):
    import ctypes as C
    fs = 50000
    # (a) two seconds of noise before the signal starts: the acquisition state crosses many segment boundaries
    sig = pdt.synth_capture(0, fs, 12.0, seed=5)
    rng = np.random.default_rng(3)
    noise = rng.integers(-300, 300, size=(2 * fs + 1234, 2)).astype(np.int16)
    iq = np.concatenate([noise, sig])
    o = orc.Oracle(orc.POES, fs, iq)
    assert o.lock_sample > len(noise)
    with pdt.Demodulator(pdt.MODE_POES, fs) as d:
        got, _, _ = stream_all(d, iq, 2400)
        assert pdt.format_frames(got) == o.text() and d.stats().lock_sample == o.lock_sample
    # (b) weak signal (seam repairs in every segment that is long enough to have seams), big and small pushes mixed
    p = pdt.synth_params(0, fs, 1000.0, 78)
    p.noise_gain = int(p.noise_gain * 6)
    n = 20 * fs
    weak = np.zeros((n, 2), dtype="<i2")
    pdt.synth_lib().pdt_synth_fill(C.byref(p), 0, n, weak.ctypes.data)
    o = orc.Oracle(orc.POES, fs, weak)
    with pdt.Demodulator(pdt.MODE_POES, fs) as d:
        d.stream_begin()
        parts, i = [], 0
        for b in [300000, 2400, 2400, 150001, 7, 400000, 99999]:
            parts.append(d.stream_push(weak[i:i + b])); i += b
        parts.append(d.stream_push(weak[i:]))
        parts.append(d.stream_end())
        assert pdt.format_frames(np.concatenate(parts)) == o.text()
    # (c) the M&M sampler's state (nextSample, stepSize, sampleLast) is carried as well
    iq = pdt.synth_capture(0, fs, 8.0, seed=9)
    o = orc.Oracle(orc.POES, fs, iq, chunk=3333, sampler=1, mm_range=3.0, mm_kp=0.15)
    with pdt.Demodulator(pdt.MODE_POES, fs, chunk=3333, sampler=pdt.SAMPLER_MM) as d:
        got, _, _ = stream_all(d, iq, 5000)
        assert pdt.format_frames(got) == o.text()


def test_argos_stream_against_the_oracle(pdt, orc):
    a = pdt.synth_capture(1, 32000, 20.0, f0_hz=130.0, seed=19)
    o = orc.Oracle(orc.ARGOS, 32000, a, math_mode=orc.MATH_LIBM)
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000) as d:
        got, _, _ = stream_all(d, a, 2400)
        assert pdt.format_frames(got) == o.text() and len(got) >= 10
        got, _, _ = stream_all(d, a, 50000)
        assert pdt.format_frames(got) == o.text()


def test_tiny_segments_do_not_invent_sync_words(pdt, orc):
    """Segments of one 777-sample chunk (ten bits each): many segments begin in the middle of a sync word.  The frames are
    those of the oracle (a sync word is recognised in the segment that holds its last bit, with its real first bits)."""
    a = pdt.synth_capture(1, 32000, 12.0, f0_hz=-57.0, seed=298)
    o = orc.Oracle(orc.ARGOS, 32000, a, chunk=777, math_mode=orc.MATH_LIBM)
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000, chunk=777) as d:
        for block in (777, 1554, 3000):
            got, _, _ = stream_all(d, a, block)
            assert pdt.format_frames(got) == o.text() and len(got) >= 5
    p = pdt.synth_capture(0, 50000, 4.0, seed=299)
    o = orc.Oracle(orc.POES, 50000, p, chunk=260)
    with pdt.Demodulator(pdt.MODE_POES, 50000, chunk=260) as d:
        got, _, _ = stream_all(d, p, 260)
        assert pdt.format_frames(got) == o.text() and len(got) >= 30


def test_large_pushes_take_the_table_sampler(pdt, orc):
    """Pushes of many chunks go through the boundary-state tables, entered with the carried sampler state (the chain walks the
    segment's first chunk and takes to the tables from its exit on): same frames as the oracle, and much faster than the
    sequential sampler of small pushes (PDT_SEG_SEQUENTIAL keeps that one)."""
    import os, time
    iq = pdt.synth_capture(0, 50000, 40.0, seed=41)
    o = orc.Oracle(orc.POES, 50000, iq)
    times = {}
    for env in ("", "1"):
        if env:
            os.environ["PDT_SEG_SEQUENTIAL"] = env
        try:
            with pdt.Demodulator(pdt.MODE_POES, 50000) as d:
                for block in (500000, 333333, 123457):
                    t0 = time.perf_counter()
                    got, _, _ = stream_all(d, iq, block)
                    times[(env, block)] = time.perf_counter() - t0
                    assert pdt.format_frames(got) == o.text() and len(got) >= 390
        finally:
            os.environ.pop("PDT_SEG_SEQUENTIAL", None)
    assert times[("", 500000)] < 0.6 * times[("1", 500000)], times


def test_file_ingest_overlapped_with_the_chain(pdt, orc, tmp_path):
    """pdt_demod_fd on a large file runs the chain in a few segments over the part of the capture that has arrived (the streaming
    path, in place).  PDT_OVERLAP_MIN_MB brings the threshold down to this test's size: same text as the oracle, and as the plain call."""
    import os
    iq = pdt.synth_capture(0, 250000, 12.0, seed=61)                     # 3 M samples, 300 chunks
    o = orc.Oracle(orc.POES, 250000, iq)
    wav = str(tmp_path / "cap.wav")
    pdt.write_wav(wav, 250000, iq)
    texts = {}
    for segs in ("", "3", "7"):
        if segs:
            os.environ["PDT_OVERLAP_MIN_MB"] = "1"
            os.environ["PDT_OVERLAP_SEGMENTS"] = segs
        try:
            with pdt.Demodulator(pdt.MODE_POES, 250000) as d:
                fd = os.open(wav, os.O_RDONLY)
                try:
                    d.demod_file(fd, 44, len(iq), 0)
                finally:
                    os.close(fd)
                texts[segs] = d.text()
                assert d.stats().samples == len(iq) and d.stats().frames == len(o.frames())
        finally:
            os.environ.pop("PDT_OVERLAP_MIN_MB", None)
            os.environ.pop("PDT_OVERLAP_SEGMENTS", None)
    assert texts[""] == o.text() and texts["3"] == o.text() and texts["7"] == o.text()


def _with_env(env, fn):
    import os
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k in env:
            os.environ.pop(k, None)


@pytest.mark.parametrize("fs,block,extra", [(250000, 1040000, {"PDT_GSPAN": "13"}), (250000, 260000, {})])
def test_aligned_segments_take_the_whole_capture_kernels(pdt, orc, fs, block, extra):
    """Round 5: a segment whose first new sample lies on the grid of the whole-capture kernels' units (a multiple of the mix + FIR
    kernel's 208-output runs and of a 128-byte line; the table rows' span dividing its first chunk) runs k_mix_fir /
    k_agc_block_tr / rows of several chunks instead of the stream path's kernels.  Same frames as the oracle; PDT_SEG_PLAIN keeps
    the stream path for the comparison, and the profile says which kernels ran."""
    secs = 24.0
    iq = pdt.synth_capture(0, fs, secs, seed=71)
    o = orc.Oracle(orc.POES, fs, iq)

    def run():
        with pdt.Demodulator(pdt.MODE_POES, fs, profile=True) as d:
            got, _, _ = stream_all(d, iq, block)
            return got, d.text(), d.kernel_times()

    got, text, kt = _with_env(extra, run)
    assert text == o.text() and pdt.format_frames(got) == o.text() and len(got) >= int(secs * 10) - 12
    assert "mix_fir" in kt, kt                              # (the last segment's groups: the fused kernel at INTERP 1)

    # stage by stage: after every push the window's newest FIR / AGC outputs are the oracle's, bit for bit (the window is
    # local: its last element is the stream's newest)
    def stages():
        ofir, oagc = o.stage(orc.ST_FIR), o.stage(orc.ST_AGC)
        with pdt.Demodulator(pdt.MODE_POES, fs) as d:
            d.stream_begin()
            for i in range(0, len(iq), block):
                d.stream_push(iq[i:i + block])
                done = min(i + block, len(iq)) // 10000 * 10000          # whole reference chunks so far
                # (the window has already slid when the push returns: the tails the next segment looks back on were copied down over
                # the window's older part, at least 700 000 elements from its end for these block sizes)
                new = min(done - (i // 10000 * 10000), 700000)
                for st, want in ((pdt.ST_FIR, ofir), (pdt.ST_AGC, oagc)):
                    a = d.stage(st)
                    assert len(a) >= new and a[len(a) - new:].tobytes() == want[done - new:done].tobytes(), (st, i)
            d.stream_end()
    _with_env(extra, stages)
    got_p, text_p, kt_p = _with_env(dict(extra, PDT_SEG_PLAIN="1"), run)
    assert text_p == o.text() and "mix_fir" not in kt_p


@pytest.mark.parametrize("fs,secs", [(250000, 75.0), (50000, 1010.0)])
def test_overlapped_file_in_unequal_segments(pdt, orc, tmp_path, fs, secs):
    """pdt_demod_file on a large file (threshold brought down with PDT_OVERLAP_MIN_MB): three unequal segments cut on the
    whole-capture kernels' grid (2 080 000 samples at chunk 10 000 and INTERP 1; 6 240 000 at INTERP 3, where the unit is a FIR
    tile of 4 992 outputs), each segment's text written while the next one runs -- the file's bytes are the oracle's text,
    whatever the split, and the same as the plain call's."""
    import os
    iq = pdt.synth_capture(0, fs, secs, seed=73)                          # 9 (8) grid units
    o = orc.Oracle(orc.POES, fs, iq)
    wav = str(tmp_path / "cap.wav")
    pdt.write_wav(wav, fs, iq)
    outs = {}
    for name, env in (("plain", {"PDT_NO_OVERLAP": "1"}), ("default", {"PDT_OVERLAP_MIN_MB": "1", "PDT_GSPAN": "13"}),
                      ("split", {"PDT_OVERLAP_MIN_MB": "1", "PDT_OVERLAP_SPLIT": "0.35,0.25,0.25,0.15", "PDT_GSPAN": "16"}),
                      ("stream_kernels", {"PDT_OVERLAP_MIN_MB": "1", "PDT_SEG_PLAIN": "1"})):
        def run():
            with pdt.Demodulator(pdt.MODE_POES, fs, profile=True).keep_pll(False) as d:
                fd = os.open(wav, os.O_RDONLY)
                outp = str(tmp_path / f"{name}.txt")
                fo = os.open(outp, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
                try:
                    nb = d.demod_file_text(fd, 44, len(iq), fo, 0)
                finally:
                    os.close(fd)
                    os.close(fo)
                data = open(outp, "rb").read()
                assert nb == len(data) and data == d.text()
                assert d.stats().samples == len(iq) and d.stats().frames == len(o.frames())
                return data, d.kernel_times()
        outs[name] = _with_env(env, run)
    for name, (data, kt) in outs.items():
        assert data == o.text(), name
    if fs == 250000:
        assert "mix_fir" in outs["default"][1] and "mix_fir" in outs["split"][1] and "mix_fir" not in outs["stream_kernels"][1]


@pytest.mark.parametrize("lead_noise", [0, 7_000_000])
def test_overlapped_segments_keep_the_chunk_reports(pdt, tmp_path, lead_noise):
    """pdt_keep_quality with the overlapped ingest: the segments end on chunk boundaries, averagePhase (which goes on behind
    the lock) and the symbol / bit counts are carried from one to the next -- the per-chunk reports are those of the plain
    whole-capture call, bit for bit; pdt_set_progress hands them on segment by segment (in chunk order, each chunk once, the
    frame whose sync word lies in one segment and whose last byte in the next counted where ByteSync counts it).  With seven
    million samples of faint noise in front the PLL locks in the second of four equal segments: the first one reports the
    acquisition's averagePhase."""
    import os
    fs = 250000
    iq = pdt.synth_capture(0, fs, 75.0 - lead_noise / fs, seed=74)
    if lead_noise:
        # (a few LSB of noise rather than zeros: StaticGain normalises by the first chunk's mean magnitude)
        iq = np.concatenate([np.random.default_rng(5).integers(-3, 4, size=(lead_noise, 2), dtype=np.int16), iq])
    wav = str(tmp_path / "cap.wav")
    pdt.write_wav(wav, fs, iq)
    res = {}
    for name, env in (("plain", {"PDT_NO_OVERLAP": "1"}), ("overlapped", {"PDT_OVERLAP_MIN_MB": "1"}),
                      ("split", {"PDT_OVERLAP_MIN_MB": "1", "PDT_OVERLAP_SPLIT": "0.25,0.25,0.25,0.25"})):
        def run():
            calls = []
            with pdt.Demodulator(pdt.MODE_POES, fs, profile=True).keep_pll(False) as d:
                d.keep_quality().set_progress(lambda first, rep, st: calls.append((first, rep, st.lock_sample, st.norm_factor)))
                fd = os.open(wav, os.O_RDONLY)
                fo = os.open(str(tmp_path / f"{name}.txt"), os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
                try:
                    d.demod_file_text(fd, 44, len(iq), fo, 0)
                finally:
                    os.close(fd)
                    os.close(fo)
                return d.chunk_reports(), calls, d.text(), d.stats(), d.kernel_times()
        res[name] = _with_env(env, run)
    rep0, calls0, text0, st0, _ = res["plain"]
    assert len(rep0) == (len(iq) + 9999) // 10000 and len(calls0) == 1 and calls0[0][0] == 0
    assert calls0[0][1].tobytes() == rep0.tobytes()
    assert rep0["symbols"].sum() == st0.symbols and rep0["bits"].sum() == st0.bits and rep0["frames"].sum() == st0.frames
    for name, ncalls in (("overlapped", 3), ("split", 4)):
        rep, calls, text, st, kt = res[name]
        assert text == text0 and "mix_fir" in kt and "quality" in kt
        assert rep.tobytes() == rep0.tobytes(), name
        assert len(calls) == ncalls
        at = 0
        for first, r, lock_sample, norm in calls:
            assert first == at and len(r) > 0 and norm == st0.norm_factor
            at += len(r)
        assert at == len(rep0)
        assert np.concatenate([c[1] for c in calls]).tobytes() == rep0.tobytes()
        assert calls[-1][2] == st0.lock_sample
        if lead_noise:                                 # nothing to lock on in the first quarter (the first 55 % hold the lock)
            assert st0.lock_sample >= lead_noise and calls[0][2] == (-1 if name == "split" else st0.lock_sample)


@pytest.mark.timeout(600)
def test_capture_that_does_not_fit_goes_through_the_bounded_window(pdt, tmp_path):
    """The reference's chunk loop takes a file of any length in O(chunk) memory (POESTIPdemod/main.c:373, while(!feof)).  The
    one-piece path keeps every stage's stream of the capture resident; when that does not fit what the device has free, the
    file entries feed the capture through the bounded window of the streaming path instead of failing.  BASELINE configs[1]
    (50 ksps, 10 min) on a device that pretends to have 2 GB: same text, same per-chunk reports handed on piece by piece, the
    window really bounded; the host-memory entry likewise; and the C host program (-D: its developer switch)."""
    import os
    import subprocess
    fs = 50000
    iq = pdt.synth_capture(0, fs, 600.0, seed=76)
    wav = str(tmp_path / "c2.wav")
    pdt.write_wav(wav, fs, iq)

    def run(name, use_mem=False):
        calls = []
        with pdt.Demodulator(pdt.MODE_POES, fs).keep_pll(False) as d:
            d.keep_quality().set_progress(lambda first, rep, st: calls.append((first, rep)))
            if use_mem:
                d.demod(iq)
                text = d.text()
            else:
                fd = os.open(wav, os.O_RDONLY)
                fo = os.open(str(tmp_path / f"{name}.txt"), os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
                try:
                    d.demod_file_text(fd, 44, len(iq), fo, 0)
                finally:
                    os.close(fd)
                    os.close(fo)
                text = open(str(tmp_path / f"{name}.txt"), "rb").read()
                assert text == d.text()
            return text, d.chunk_reports(), calls, d.stats(), d.stage_len(pdt.ST_AGC)

    text0, rep0, calls0, st0, agc0 = run("whole")
    assert st0.segments == 1 and st0.windowed == 0 and agc0 == 3 * len(iq) and len(text0) > 1_000_000
    for use_mem in (False, True):
        text, rep, calls, st, agc = _with_env({"PDT_HBM_LIMIT_MB": "2048"}, lambda: run("windowed", use_mem))
        assert st.windowed == 1 and st.segments >= 3, (st.windowed, st.segments)
        assert text == text0
        assert rep.tobytes() == rep0.tobytes()
        assert len(calls) == st.segments and [c[0] for c in calls] == list(np.cumsum([0] + [len(c[1]) for c in calls[:-1]]))
        assert np.concatenate([c[1] for c in calls]).tobytes() == rep0.tobytes()
        assert agc < 3 * len(iq) // 2                                  # (the last piece's window, not the capture)
        assert (st.samples, st.symbols, st.bits, st.frames, st.lock_sample) == (st0.samples, st0.symbols, st0.bits, st0.frames, st0.lock_sample)
    # small pieces (a piece of 37 chunks: many slides of the window), ARGOS in double precision likewise
    text, rep, calls, st, _ = _with_env({"PDT_WINDOW_PIECE": "370000"}, lambda: run("pieces"))
    assert text == text0 and rep.tobytes() == rep0.tobytes() and st.segments == (len(iq) + 369999) // 370000
    aq = pdt.synth_capture(1, 32000, 40.0, seed=77)
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000) as d:
        d.demod(aq)
        want = d.text()
    def argos():
        with pdt.Demodulator(pdt.MODE_ARGOS, 32000) as d:
            d.demod(aq)
            return d.text(), d.stats()
    got, st = _with_env({"PDT_WINDOW_PIECE": "240000"}, argos)
    assert got == want and len(want) > 100 and st.windowed == 1
    # the host program: demodPOES -D PDT_HBM_LIMIT_MB=2048
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "demodPOES")
    outp = str(tmp_path / "cli.txt")
    r = subprocess.run([exe, "-D", "PDT_HBM_LIMIT_MB=2048", "-o", outp, wav], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert open(outp, "rb").read() == text0
    os.unlink(outp)
    r2 = subprocess.run([exe, "-D", "PDT_HBM_LIMIT_MB=64", "-P", "-o", outp, wav], capture_output=True, text=True)   # (a window of 31 chunks)
    assert r2.returncode == 0 and open(outp, "rb").read() == text0
    r3 = subprocess.run([exe, "-D", "PDT_HBM_LIMIT_MB=8", "-o", outp, wav], capture_output=True, text=True)
    assert r3.returncode != 0 and "Demodulation failed" in r3.stdout        # (not even a window of a few chunks fits 8 MB)


def test_uncached_file_is_read_directly_into_the_pinned_staging(pdt, tmp_path):
    """Round 6 (the host-memory budget at N = 8): a capture file whose pages are NOT in the page cache is read with
    O_DIRECT straight into the pinned slots -- two passes over host memory per byte instead of three; one that is cached (or lives
    on tmpfs, which has no direct I/O) keeps the buffered reads.  The 44-byte header makes every span start off a block boundary:
    the reads start on the 4 KiB boundary below and the copy to the GPU starts inside the slot.  Same text either way."""
    import os
    fs = 250000
    iq = pdt.synth_capture(0, fs, 80.0, seed=79)                     # 80 MB: above the size where the ingest probes at all
    wav = str(tmp_path / "cold.wav")
    pdt.write_wav(wav, fs, iq)
    with pdt.Demodulator(pdt.MODE_POES, fs) as ref:
        ref.demod(iq)
        want = ref.text()

    def run(evict):
        fd = os.open(wav, os.O_RDONLY)
        try:
            if evict:
                os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            with pdt.Demodulator(pdt.MODE_POES, fs).keep_pll(False) as d:
                d.demod_file(fd, 44, len(iq))
                return d.text(), d.stats()
        finally:
            os.close(fd)

    wfd = os.open(wav, os.O_RDWR)
    os.fsync(wfd)                                                     # (clean pages can be dropped)
    os.close(wfd)
    text, st = _with_env({"PDT_INGEST_DIRECT": "1"}, lambda: run(False))          # forced: whatever the page cache holds
    assert text == want
    if not st.ingest_direct:
        pytest.skip("no O_DIRECT on the file system under tmp_path")
    text, st = run(True)                                              # evicted: the probe finds it cold
    assert text == want and st.ingest_direct == 1
    open(wav, "rb").read()                                            # cached again: buffered reads
    text, st = run(False)
    assert text == want and st.ingest_direct == 0
    text, st = _with_env({"PDT_INGEST_NUMA": "1"}, lambda: run(False))            # readers + staging bound to the GPU's node
    assert text == want


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["plain", "overlapped"])
def test_file_entry_errors_leave_the_context_usable(pdt, tmp_path, mode):
    """A capture file that ends before the claimed number of frames (PDT_ERR_FORMAT: the reader threads, the submitter, the
    segments that have already run and the marks the later ones wait on all have to wind down), and a text descriptor that
    cannot be written (PDT_ERR_IO) -- ingested first and in overlapped segments; the same context then demodulates the file
    properly, with the same text as a fresh one."""
    import os
    fs = 250000
    iq = pdt.synth_capture(0, fs, 75.0, seed=75)
    wav = str(tmp_path / "cap.wav")
    pdt.write_wav(wav, fs, iq)
    with pdt.Demodulator(pdt.MODE_POES, fs) as ref:
        ref.demod(iq)
        want = ref.text()
    assert len(want) > 1000

    def run():
        with pdt.Demodulator(pdt.MODE_POES, fs).keep_pll(False) as d:
            fd = os.open(wav, os.O_RDONLY)
            outp = str(tmp_path / f"{mode}.txt")
            fo = os.open(outp, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
            ro = os.open(outp, os.O_RDONLY)
            try:
                for claimed in (len(iq) + 5_000_000, len(iq) + 1, 4 * len(iq)):
                    with pytest.raises(pdt.PdtError, match="WAV|format|ends"):
                        d.demod_file_text(fd, 44, claimed, fo, 0)
                with pytest.raises(pdt.PdtError):
                    d.demod_file_text(fd, 44, len(iq), ro, 0)                   # text cannot be written
                os.ftruncate(fo, 0)
                os.lseek(fo, 0, os.SEEK_SET)
                nb = d.demod_file_text(fd, 44, len(iq), fo, 0)
                assert open(outp, "rb").read() == want and nb == len(want) and d.text() == want
                d.demod(iq)                                                     # and the other entries work as ever
                assert d.text() == want
            finally:
                os.close(fd)
                os.close(fo)
                os.close(ro)

    _with_env({"PDT_OVERLAP_MIN_MB": "1"} if mode == "overlapped" else {"PDT_NO_OVERLAP": "1"}, run)


def test_stage_and_whole_capture_entries_are_refused_while_a_stream_is_open(pdt, clip):
    """The stage buffers hold the tails an open stream continues from: every pdt_stage_* / pdt_demod_* entry
    returns PDT_ERR_STATE between the first push and pdt_stream_end, and the stream is not disturbed by the attempt."""
    rate, iq = clip
    with pdt.Demodulator(pdt.MODE_POES, rate) as ref:
        ref.demod(iq)
        want = ref.frames_array()
    with pdt.Demodulator(pdt.MODE_POES, rate) as d:
        parts = [d.stream_push(iq[:60000])]                           # no pdt_stream_begin: the first push opens the stream
        for call in (lambda: d.demod(iq[:20000]), lambda: d.stage_fir(np.zeros(100, np.float32)),
                     lambda: d.stage_agc(np.ones(100, np.float32), 1.0), lambda: d.stage_pll(iq[:1000]),
                     lambda: d.stage_manchester(np.ones(64, np.float32), 1.0), lambda: d.bytesync(np.full(100, 48, np.uint8)),
                     lambda: d.stage_static_gain(iq[:1000]), lambda: d.stage_squelch(np.ones(8, np.float32), np.ones(8, np.float32), 0.1)):
            with pytest.raises(pdt.PdtError, match="call sequence"):
                call()
        parts.append(d.stream_push(iq[60000:]))
        parts.append(d.stream_end())
        assert np.concatenate(parts).tobytes() == want.tobytes()
        d.demod(iq[:20000])                                           # the stream is over: whole-capture calls work again
        parts = [d.stream_push(iq[:123456]), d.stream_push(iq[123456:]), d.stream_end()]     # and so does a new stream, without begin
        assert np.concatenate(parts).tobytes() == want.tobytes()
