"""Stage-level entry points: the oracle's per-chunk dumps replayed stage by stage -- each chunk handed to ONE stage's
kernels with the reference function's statics as an explicit state record (pdt_manchester_state, pdt_fir_state), chunk after
chunk; the outputs must be the oracle's stage streams, the carried records what the reference's statics hold."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def chunks_of(n, chunk):
    return [(a, min(a + chunk, n)) for a in range(0, n, chunk)]


@pytest.mark.parametrize("fs,chunk", [(50000, 10000), (250000, 10000), (32000, 777), (100000, 1)])
def test_fir_stage_replays_the_oracle_chunk_by_chunk(pdt, orc, fs, chunk):
    secs = 2.0 if chunk > 1 else 0.004
    iq = pdt.synth_capture(0, fs, secs, seed=41)
    o = orc.Oracle(orc.POES, fs, iq, chunk=max(chunk, 64))
    x = o.stage(orc.ST_PLL)                       # the FIR's input stream; the filter does not care where its chunks end
    want = o.stage(orc.ST_FIR)
    interp = o.interp
    with pdt.Demodulator(pdt.MODE_POES, fs) as d:
        st = pdt.FirState()
        got = [d.stage_fir(x[a:b], st) for a, b in chunks_of(len(x), chunk)]
        got = np.concatenate(got)
        assert got.dtype == want.dtype and got.tobytes() == want.tobytes()
        assert st.count == len(x)
        K = 26
        assert np.array_equal(np.array(st.history[:K], dtype=np.float32), x[-K:])
        # without a state record every call is a fresh filter: the first chunk again
        a, b = chunks_of(len(x), chunk)[0]
        assert d.stage_fir(x[a:b]).tobytes() == want[:b * interp].tobytes()
        assert len(d.stage_fir(x[:0], st)) == 0 and st.count == len(x)


def test_fir_stage_argos(pdt, orc):
    iq = pdt.synth_capture(1, 32000, 4.0, f0_hz=160.0, seed=42)
    o = orc.Oracle(orc.ARGOS, 32000, iq)
    x = o.stage(orc.ST_PLL)
    want = o.stage(orc.ST_FIR)
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000) as d:
        st = pdt.FirState()
        got = np.concatenate([d.stage_fir(x[a:b], st) for a, b in chunks_of(len(x), 2400)])
        assert got.dtype == want.dtype and got.tobytes() == want.tobytes()


def manchester_replay(pdt, d, sym, symidx, chunk_out, thr):
    """the symbols of every reference chunk (by pick index) through pdt_stage_manchester, statics carried in the record"""
    st = pdt.ManchesterState()
    which = symidx // chunk_out
    bits, bsym, base = [], [], 0
    for c in range(int(which.max()) + 1 if len(which) else 0):
        s = sym[which == c]
        b, k = d.stage_manchester(s, thr, st)
        bits.append(b)
        bsym.append(k.astype(np.int64) + base)
        base += len(s)
    return np.concatenate(bits), np.concatenate(bsym), st


@pytest.mark.parametrize("fs,chunk", [(50000, 10000), (50000, 333), (250000, 10000)])
def test_manchester_stage_replays_the_oracle_chunk_by_chunk(pdt, orc, fs, chunk):
    iq = pdt.synth_capture(0, fs, 3.0, seed=43)
    o = orc.Oracle(orc.POES, fs, iq, chunk=chunk)
    sym, symidx = o.stage(orc.ST_SYM), o.stage(orc.ST_SYMIDX)
    want_bits, symt, bitt = o.stage(orc.ST_BITS), o.stage(orc.ST_SYMT), o.stage(orc.ST_BITT)
    with pdt.Demodulator(pdt.MODE_POES, fs, chunk=chunk) as d:
        bits, bsym, st = manchester_replay(pdt, d, sym, symidx, chunk * o.interp, 1.0)
        assert bits.tobytes() == want_bits.tobytes()
        assert symt[bsym].tobytes() == bitt.tobytes()        # the time stamp each bit inherits (ManchesterDecode.c:86)
        assert st.even_odd == len(sym) % 256 and st.current == float(sym[-1]) and st.previous == float(sym[-2])
        # one call over everything = the same bits
        b1, k1 = d.stage_manchester(sym, 1.0)
        assert b1.tobytes() == want_bits.tobytes() and np.array_equal(k1.astype(np.int64), bsym)
        # symbols one at a time across a stretch (every parity / pad combination)
        st1 = pdt.ManchesterState()
        few = np.concatenate([d.stage_manchester(sym[i:i + 1], 1.0, st1)[0] for i in range(300)])
        b300, _ = d.stage_manchester(sym[:300], 1.0)
        assert few.tobytes() == b300.tobytes()


def test_manchester_stage_argos(pdt, orc):
    iq = pdt.synth_capture(1, 32000, 8.0, f0_hz=160.0, seed=44)
    o = orc.Oracle(orc.ARGOS, 32000, iq)
    sym, symidx = o.stage(orc.ST_SYM), o.stage(orc.ST_SYMIDX)
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000) as d:
        bits, bsym, _ = manchester_replay(pdt, d, sym, symidx, 2400, 0.5)
        assert bits.tobytes() == o.stage(orc.ST_BITS).tobytes()
        assert o.stage(orc.ST_SYMT)[bsym].tobytes() == o.stage(orc.ST_BITT).tobytes()


@pytest.mark.parametrize("fs,chunk", [(50000, 10000), (250000, 10000), (50000, 777)])
def test_agc_stage_replays_the_oracle_chunk_by_chunk(pdt, orc, fs, chunk):
    iq = pdt.synth_capture(0, fs, 3.0, seed=45)
    o = orc.Oracle(orc.POES, fs, iq, chunk=chunk)
    x, want = o.stage(orc.ST_FIR), o.stage(orc.ST_AGC)
    step = chunk * o.interp
    with pdt.Demodulator(pdt.MODE_POES, fs, chunk=chunk) as d:
        st = pdt.AgcState()
        got = np.concatenate([d.stage_agc(x[a:b], o.norm_factor, st) for a, b in chunks_of(len(x), step)])
        assert got.dtype == want.dtype and got.tobytes() == want.tobytes()
        assert st.started == 1
        # the carried gain is the reference's static: one more sample from it = the same sample in one go
        whole = pdt.AgcState()
        ref = d.stage_agc(x, o.norm_factor, whole)
        assert ref.tobytes() == want.tobytes() and whole.gain == st.gain
        # `initial` counts on the first call only (AGC.c:91-95)
        st2 = pdt.AgcState()
        a = d.stage_agc(x[:step], o.norm_factor, st2)
        b = d.stage_agc(x[step:2 * step], 123.0, st2)
        assert np.concatenate([a, b]).tobytes() == want[:2 * step].tobytes()
    # tiny blocks with a warm-up far too short: every seam is repaired, same result
    with pdt.Demodulator(pdt.MODE_POES, fs, chunk=chunk, agc_block=64, agc_warm=64) as d:
        st = pdt.AgcState()
        n = 20 * step
        got = np.concatenate([d.stage_agc(x[a:b], o.norm_factor, st) for a, b in chunks_of(n, step)])
        assert got.tobytes() == want[:n].tobytes()


def test_agc_and_squelch_stages_argos(pdt, orc):
    iq = pdt.synth_capture(1, 32000, 8.0, f0_hz=160.0, seed=46)
    o = orc.Oracle(orc.ARGOS, 32000, iq)
    x, raw, want, lock = o.stage(orc.ST_FIR), o.stage(orc.ST_AGC_RAW), o.stage(orc.ST_AGC), o.stage(orc.ST_LOCK)
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000) as d:
        st = pdt.AgcState()
        got = np.concatenate([d.stage_agc(x[a:b], o.norm_factor, st) for a, b in chunks_of(len(x), 2400)])
        assert got.tobytes() == raw.tobytes()                                   # NormalizingAGC alone: what -r dumps
        sq = np.concatenate([d.stage_squelch(got[a:b], lock[a:b], 0.15) for a, b in chunks_of(len(x), 2400)])
        assert sq.tobytes() == want.tobytes()                                   # ... then Squelch (ARGOSdemod/main.c:276)


def as_complex_float(iq):
    """the reference's sample conversion: value / 32768 in float (wave.c:127-172)"""
    return (iq.astype(np.float32) / np.float32(32768.0)).astype("<f4")


def pll_replay(pdt, d, x, chunk):
    st = pdt.PllState()
    outs, locks, rets = [], [], []
    for a, b in chunks_of(len(x), chunk):
        o, l, r = d.stage_pll(x[a:b], st)
        outs.append(o); locks.append(l); rets.append(r)
    return np.concatenate(outs), np.concatenate(locks), np.array(rets), st


@pytest.mark.parametrize("chunk", [10000, 1000])
def test_pll_stage_replays_the_clip_chunk_by_chunk(pdt, orc, clip, chunk):
    """CarrierTrackPLL chunk after chunk with its statics in the caller's record: realDataOut = the oracle's PLL stream, the return
    values = the reference's own (golden, made by its objects), the lock happens in the call and at the index the oracle says"""
    import os
    from conftest import GOLDEN
    rate, iq = clip
    x = as_complex_float(iq)
    o = orc.Oracle(orc.POES, rate, iq, chunk=chunk)
    want = o.stage(orc.ST_PLL)
    avg_ref = np.fromfile(os.path.join(GOLDEN, f"clip.c{chunk}.avg.f32"), dtype="<f4")
    with pdt.Demodulator(pdt.MODE_POES, rate, chunk=chunk) as d:
        out, lock, rets, st = pll_replay(pdt, d, x, chunk)
        assert out.tobytes() == want.tobytes()
        assert rets.astype("<f4").tobytes() == avg_ref[:len(rets)].tobytes()
        assert st.started == 1 and st.locked == 1
        assert st.lock_index == o.lock_sample % chunk
        assert f"{st.lock_freq_hz:0.2f}" == f"{o.lock_freq_hz:0.2f}"
        # any other cut of the same stream: same outputs, same statics at the end
        out2, lock2, rets2, st2 = pll_replay(pdt, d, x, 33333)
        assert out2.tobytes() == want.tobytes() and lock2.tobytes() == lock.tobytes()
        for f in ("phase", "freq", "avg_phase", "locksig", "sweep"):
            assert getattr(st, f) == getattr(st2, f), f
        # the whole-capture path afterwards is unaffected by the stage calls
        d.demod(iq)
        assert d.text() == o.text()


def test_pll_stage_lock_stream_live_chain(pdt, orc):
    """the twin's constants and its lock stream (POESTIPdemodPortAudio/main.c:367): lockSignalStreamOut chunk by chunk"""
    fs = 48000
    iq = pdt.synth_capture(0, fs, 4.0, f0_hz=-1500.0, seed=47)
    x = as_complex_float(iq)
    o = orc.Oracle(orc.POES, fs, x.reshape(-1, 2), chunk=2400, chain=1)       # float32 capture, as the sound card delivers
    with pdt.Demodulator(pdt.MODE_POES, fs, chunk=2400, chain=1) as d:
        out, lock, rets, st = pll_replay(pdt, d, x, 2400)
        assert lock.tobytes() == o.stage(orc.ST_LOCK).tobytes()
        # the oracle's PLL stream is the one after Squelch there; before it the two agree wherever the lock stream is above 0.05
        want = o.stage(orc.ST_PLL)
        keep = lock >= np.float32(0.05)
        assert np.array_equal(out[keep].view(np.uint32), want[keep].view(np.uint32)) and not want[~keep].any()


def test_pll_stage_never_locks_on_noise(pdt, orc):
    rng = np.random.default_rng(48)
    iq = np.clip(rng.normal(0, 3000, (60000, 2)), -32768, 32767).astype("<i2")
    x = as_complex_float(iq)
    o = orc.Oracle(orc.POES, 50000, iq)
    with pdt.Demodulator(pdt.MODE_POES, 50000) as d:
        out, lock, rets, st = pll_replay(pdt, d, x, 7777)
        assert o.lock_sample < 0 and st.locked == 0
        assert out.tobytes() == o.stage(orc.ST_PLL).tobytes()
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000) as d:
        with pytest.raises(pdt.PdtError):
            d.stage_pll(x[:100])                                      # no `double complex` sample source


@pytest.mark.parametrize("chunk", [2400, 1001])
def test_pll_stage_argos_from_pcm16(pdt, orc, chunk):
    """the double-precision build, fed the int16 pairs the WAV holds: realDataOut, the lock stream and the return values"""
    iq = pdt.synth_capture(1, 32000, 9.0, f0_hz=-140.0, seed=53)[:-7]
    o = orc.Oracle(orc.ARGOS, 32000, iq, chunk=chunk)
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000, chunk=chunk) as d:
        out, lock, rets, st = pll_replay(pdt, d, iq, chunk)
        assert out.tobytes() == o.stage(orc.ST_PLL).tobytes()
        assert lock.tobytes() == o.stage(orc.ST_LOCK).tobytes()
        avg = o.stage(orc.ST_AVG)
        assert rets.tobytes() == avg[:len(rets)].tobytes()
        assert st.locked == 1 and st.lock_index == o.lock_sample % chunk
        out2, lock2, _, st2 = pll_replay(pdt, d, iq, 50000)
        assert out2.tobytes() == out.tobytes() and lock2.tobytes() == lock.tobytes() and st2.phase == st.phase and st2.freq == st.freq


def test_pll_stage_pcm16_poes(pdt, orc, clip):
    rate, iq = clip
    o = orc.Oracle(orc.POES, rate, iq)
    with pdt.Demodulator(pdt.MODE_POES, rate) as d:
        out, _, rets, _ = pll_replay(pdt, d, np.ascontiguousarray(iq), 10000)
        assert out.tobytes() == o.stage(orc.ST_PLL).tobytes()
        assert rets.astype("<f4").tobytes() == o.stage(orc.ST_AVG)[:len(rets)].tobytes()


def gardner_replay(pdt, d, x, C, lock=None):
    """the reference's chunk loop around the sampler: one persistent buffer of C elements that every chunk overwrites from
    index 0 (what lies behind the chunk's end stays), statics in the record"""
    st = pdt.GardnerState()
    buf = np.zeros(C, dtype=x.dtype)
    nbuf = np.zeros(C, dtype=x.dtype) if lock is not None else None
    syms, picks = [], []
    for a, b in chunks_of(len(x), C):
        buf[:b - a] = x[a:b]
        if nbuf is not None:
            nbuf[:b - a] = lock[a:b]
        s, p = d.stage_gardner(buf, b - a, st, nbuf)
        syms.append(s)
        picks.append(p.astype(np.int64) + a)
    return np.concatenate(syms), np.concatenate(picks), st


@pytest.mark.parametrize("chunk", [10000, 333])
def test_gardner_stage_replays_the_clip_chunk_by_chunk(pdt, orc, clip, chunk):
    rate, iq = clip                                                   # 250 195 samples: a short last chunk (Q3's stale reads)
    o = orc.Oracle(orc.POES, rate, iq, chunk=chunk)
    x = o.stage(orc.ST_AGC)
    with pdt.Demodulator(pdt.MODE_POES, rate, chunk=chunk) as d:
        sym, pick, st = gardner_replay(pdt, d, x, chunk * o.interp)
        assert sym.tobytes() == o.stage(orc.ST_SYM).tobytes()
        assert np.array_equal(pick, o.stage(orc.ST_SYMIDX))
        assert st.prev_bit == float(sym[-1])


@pytest.mark.parametrize("fs", [250000, 32000])
def test_gardner_stage_other_rates(pdt, orc, fs):
    iq = pdt.synth_capture(0, fs, 2.0, seed=49)[:-123]
    o = orc.Oracle(orc.POES, fs, iq)
    with pdt.Demodulator(pdt.MODE_POES, fs) as d:
        sym, pick, _ = gardner_replay(pdt, d, o.stage(orc.ST_AGC), 10000 * o.interp)
        assert sym.tobytes() == o.stage(orc.ST_SYM).tobytes() and np.array_equal(pick, o.stage(orc.ST_SYMIDX))


@pytest.mark.parametrize("chunk", [2400, 1001])
def test_gardner_stage_argos_heap_neighbour(pdt, orc, chunk):
    """ARGOS: the reads past the buffer land in the lock-signal array behind it (Q16), for both alignments of the size field"""
    iq = pdt.synth_capture(1, 32000, 8.0, f0_hz=160.0, seed=50)[:-77]
    o = orc.Oracle(orc.ARGOS, 32000, iq, chunk=chunk)
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000, chunk=chunk) as d:
        sym, pick, _ = gardner_replay(pdt, d, o.stage(orc.ST_AGC), chunk, lock=o.stage(orc.ST_LOCK))
        assert sym.tobytes() == o.stage(orc.ST_SYM).tobytes() and np.array_equal(pick, o.stage(orc.ST_SYMIDX))


def test_static_gain_stage(pdt, orc, clip):
    rate, iq = clip
    o = orc.Oracle(orc.POES, rate, iq)
    with pdt.Demodulator(pdt.MODE_POES, rate) as d:
        assert np.float32(d.stage_static_gain(iq[:10000])) == np.float32(o.norm_factor)
        assert np.float32(d.stage_static_gain(as_complex_float(iq[:10000]))) == np.float32(o.norm_factor)
        o2 = orc.Oracle(orc.POES, rate, iq, chunk=777)
        assert np.float32(d.stage_static_gain(iq[:777])) == np.float32(o2.norm_factor)
    a = pdt.synth_capture(1, 32000, 1.0, seed=51)
    oa = orc.Oracle(orc.ARGOS, 32000, a)
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000) as d:
        assert d.stage_static_gain(a[:2400]) == oa.norm_factor


@pytest.mark.parametrize("mode,fs,chunk", [(0, 50000, 10000), (0, 50000, 777), (1, 32000, 2400)])
def test_mm_stage_replays_the_oracle_chunk_by_chunk(pdt, orc, mode, fs, chunk):
    iq = pdt.synth_capture(mode, fs, 3.0 if mode == 0 else 8.0, seed=52)[:-31]
    o = orc.Oracle(mode, fs, iq, chunk=chunk, sampler=1)
    x = o.stage(orc.ST_AGC)
    step = chunk * o.interp
    with pdt.Demodulator(mode, fs, chunk=chunk, sampler=1) as d:
        st = pdt.MmState()
        syms, picks = [], []
        for a, b in chunks_of(len(x), step):
            s, p = d.stage_mm(x[a:b], st)
            syms.append(s)
            picks.append(p.astype(np.int64) + a)
        assert np.concatenate(syms).tobytes() == o.stage(orc.ST_SYM).tobytes()
        assert np.array_equal(np.concatenate(picks), o.stage(orc.ST_SYMIDX))
        assert st.started == 1 and st.sample_last == float(syms[-1][-1])
