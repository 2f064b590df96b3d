"""the sound-card twin's chain (POESTIPdemodPortAudio/main.c:324-393) on the GPU -- float32 blocks
of 2400 frames at 48 kHz, the twin's PLL constants, Squelch between PLL and FIR, Manchester threshold 0.75 --
against the oracle (which tests/test_oracle_ref.py pins on the reference's stage objects called in the twin's order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STAGES = ["ST_PLL", "ST_LOCK", "ST_FIR", "ST_AGC", "ST_SYM", "ST_BITS"]


def live_capture(pdt, fs, secs, seed, f0, scale=1.0):
    iq = pdt.synth_capture(0, fs, secs, f0_hz=f0, seed=seed)
    return (iq.astype(np.float32) / np.float32(32768.0)) * np.float32(scale)


@pytest.mark.parametrize("fs,chunk,secs,seed,f0,scale", [(48000, 0, 6.0, 41, 900.0, 1.0), (48000, 2400, 20.0, 42, -2800.0, 0.02),
                                                         (48000, 1000, 5.0, 43, 300.0, 1.0), (50000, 10000, 8.0, 44, 1500.0, 3.0),
                                                         (48000, 2400, 60.0, 45, 2100.0, 0.3)])
def test_live_chain_matches_oracle(pdt, orc, fs, chunk, secs, seed, f0, scale):
    raw = live_capture(pdt, fs, secs, seed, f0, scale)
    o = orc.Oracle(orc.POES, fs, raw, chunk=chunk or 2400, chain=1)
    with pdt.Demodulator(pdt.MODE_POES, fs, chunk=chunk, chain=pdt.CHAIN_LIVE) as d:
        d.demod_raw(raw)
        for name in STAGES:
            got = d.stage(getattr(pdt, name))
            exp = o.stage(getattr(orc, name))
            assert got.tobytes() == exp.tobytes(), f"stage {name} differs"
        assert d.text() == o.text() and len(o.text()) > 0
        assert d.stats().lock_sample == o.lock_sample


def test_live_stream_in_blocks_of_2400(pdt, orc):
    """The twin's own shape: blocks of 2400 float32 frames pushed as they arrive; frames come out once, in order,
    while the stream is still running."""
    raw = live_capture(pdt, 48000, 12.0, 51, 700.0)
    o = orc.Oracle(orc.POES, 48000, raw, chunk=2400, chain=1, keep_stages=False)
    with pdt.Demodulator(pdt.MODE_POES, 48000, chain=pdt.CHAIN_LIVE) as d:
        d.stream_begin()
        parts = []
        for k in range(0, len(raw), 2400):
            parts.append(d.stream_push(raw[k:k + 2400]))
        early = sum(len(x) for x in parts)
        parts.append(d.stream_end())
        got = np.concatenate(parts)
        assert d.text() == o.text()
        assert got.tobytes() == d.frames_array().tobytes()
        assert early >= len(got) - 4 and len(got) > 100


def test_live_noise_only_is_squelched(pdt, orc):
    rng = np.random.default_rng(3)
    raw = (rng.standard_normal((96000, 2)) * 0.01).astype(np.float32)
    o = orc.Oracle(orc.POES, 48000, raw, chunk=2400, chain=1)
    with pdt.Demodulator(pdt.MODE_POES, 48000, chain=pdt.CHAIN_LIVE) as d:
        d.demod_raw(raw)
        assert d.stage(pdt.ST_PLL).tobytes() == o.stage(orc.ST_PLL).tobytes()
        assert d.stage(pdt.ST_AGC).tobytes() == o.stage(orc.ST_AGC).tobytes()
        assert d.text() == o.text()


@pytest.mark.parametrize("fs,chunk,secs,seed,f0,fmt", [(48000, 0, 9.0, 71, 130.0, "f32"), (48000, 2401, 8.0, 72, -75.0, "pcm"),
                                                       (48000, 2403, 8.0, 73, 40.0, "f32"), (32000, 1000, 7.0, 74, 160.0, "pcm"),
                                                       (48000, 2402, 30.0, 75, 99.0, "f32")])
def test_argos_twin_matches_oracle(pdt, orc, fs, chunk, secs, seed, f0, fmt):
    """The ARGOS sound-card twin (ARGOSdemodPortAudio/main.c:266-329; round 4): mode ARGOS + chain LIVE = the FLOAT build of the
    ARGOS chain -- float32 blocks of 2400 frames at 48 kHz, the twin's float time stamps, its synchroniser with the inverse sync
    word enabled, and behind a float block the malloc slack (0 - 3 floats by chunk size), the next chunk's size field as two
    floats and the lock signal (Q16).  Every stage equals the oracle, which tests/test_oracle_ref.py pins on the reference's
    objects built with the twin's config.h."""
    iq = pdt.synth_capture(1, fs, secs, f0_hz=f0, seed=seed)
    src = iq if fmt == "pcm" else iq.astype(np.float32) / np.float32(32768.0)
    o = orc.Oracle(orc.ARGOS, fs, src, chunk=chunk or 2400, chain=1)
    with pdt.Demodulator(pdt.MODE_ARGOS, fs, chunk=chunk, chain=pdt.CHAIN_LIVE) as d:
        assert d.dtype == np.float32
        if fmt == "pcm":
            d.demod(iq)
        else:
            d.demod_raw(src)
        for name in STAGES:
            got = d.stage(getattr(pdt, name))
            exp = o.stage(getattr(orc, name))
            assert got.dtype == np.float32 or name == "ST_BITS"
            assert got.tobytes() == exp.tobytes(), f"stage {name} differs"
        assert d.text() == o.text() and len(o.text()) > 0
        assert d.stats().lock_sample == o.lock_sample


def test_argos_twin_against_the_reference_objects(pdt, tmp_path):
    """... and the reference's own float objects (oracle/_ref/ref_demodARGOSf), run on the GPU box on a fresh capture."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_demodARGOSf")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_demodARGOSf not shipped")
    iq = pdt.synth_capture(1, 48000, 40.0, f0_hz=-120.0, seed=76)
    wav, out = tmp_path / "a.wav", tmp_path / "ref.txt"
    pdt.write_wav(str(wav), 48000, iq)
    subprocess.run([exe, str(wav), str(out)], check=True, capture_output=True)
    with pdt.Demodulator(pdt.MODE_ARGOS, 48000, chain=pdt.CHAIN_LIVE) as d:
        d.demod(iq)
        assert d.text() == out.read_bytes() and d.stats().frames >= 10


def test_argos_twin_finds_inverted_packets(pdt, orc):
    """ARGOSdemodPortAudio/ByteSync.c:112: the twin looks for the inverse sync word as well and re-inverts the packet (the file
    program does not); its stamp carries no "i" (:128).  A capture with its Q sign flipped (a spectrum-inverted receiver) has
    every packet inverted."""
    iq = pdt.synth_capture(1, 48000, 12.0, f0_hz=80.0, seed=77)
    iq = iq.copy()
    iq[:, 1] = -iq[:, 1]
    o = orc.Oracle(orc.ARGOS, 48000, iq, chunk=2400, chain=1)
    with pdt.Demodulator(pdt.MODE_ARGOS, 48000, chain=pdt.CHAIN_LIVE) as d:
        d.demod(iq)
        assert d.text() == o.text()
        assert d.stage(pdt.ST_BITS).tobytes() == o.stage(orc.ST_BITS).tobytes()


def test_cli_live_loop_from_a_pipe_and_from_a_file(pdt, orc, tmp_path):
    """bin/demodPOES -l: with "-" the twin's loop itself (standard input as the sound card), with a .raw file the
    same chain in one go; both write the oracle's file."""
    import os
    import subprocess
    from conftest import ROOT
    raw = live_capture(pdt, 48000, 9.0, 61, 1200.0)
    want = orc.Oracle(orc.POES, 48000, raw, chunk=2400, chain=1, keep_stages=False).text()
    exe = os.path.join(ROOT, "bin", "demodPOES")
    out = tmp_path / "piped.txt"
    r = subprocess.run([exe, "-l", "-s", "48", "-o", str(out), "-"], input=raw.tobytes(), capture_output=True)
    assert r.returncode == 0, r.stdout.decode()
    assert out.read_bytes() == want and b"PLL locked at" in r.stdout
    path = tmp_path / "cap.raw"
    raw.tofile(path)
    out2 = tmp_path / "file.txt"
    r = subprocess.run([exe, "-l", "-s", "48", "-o", str(out2), str(path)], capture_output=True)
    assert r.returncode == 0, r.stdout.decode()
    assert out2.read_bytes() == want


def test_live_chain_on_the_real_clip(pdt, orc, clip):
    """The reference's bundled NOAA-15 clip (real signal, 50 ksps) through the twin's chain: every stage equals the oracle,
    and the frames are good TIP frames (the chain differs from the file program's, the satellite's data do not)."""
    rate, iq = clip
    raw = iq.astype(np.float32) / np.float32(32768.0)
    o = orc.Oracle(orc.POES, rate, raw, chunk=2400, chain=1)
    with pdt.Demodulator(pdt.MODE_POES, rate, chain=pdt.CHAIN_LIVE) as d:
        d.demod_raw(raw)
        for name in STAGES:
            assert d.stage(getattr(pdt, name)).tobytes() == o.stage(getattr(orc, name)).tobytes(), f"stage {name} differs"
        assert d.text() == o.text()
        summary, _ = d.tip_check()
        assert summary["frames_checked"] >= 40 and summary["good_frames"] >= summary["frames_checked"] - 2


def test_argos_twin_stream_and_cli(pdt, orc, tmp_path):
    """The twin's own shape for ARGOS: blocks of 2400 float32 frames pushed as they arrive (pdt_stream_push_f32), and
    bin/demodARGOS -l with standard input as the sound card or with a WAV file; all of them write the oracle's text."""
    import os
    import subprocess
    from conftest import ROOT
    iq = pdt.synth_capture(1, 48000, 14.0, f0_hz=-60.0, seed=78)
    raw = iq.astype(np.float32) / np.float32(32768.0)
    o = orc.Oracle(orc.ARGOS, 48000, raw, chunk=2400, chain=1, keep_stages=False)
    assert len(o.text()) > 0
    with pdt.Demodulator(pdt.MODE_ARGOS, 48000, chain=pdt.CHAIN_LIVE) as d:
        d.stream_begin()
        parts = [d.stream_push(raw[k:k + 2400]) for k in range(0, len(raw), 2400)]
        parts.append(d.stream_end())
        assert d.text() == o.text()
        assert np.concatenate(parts).tobytes() == d.frames_array().tobytes()
    exe = os.path.join(ROOT, "bin", "demodARGOS")
    out = tmp_path / "piped.txt"
    r = subprocess.run([exe, "-l", "-o", str(out), "-"], input=raw.tobytes(), capture_output=True)
    assert r.returncode == 0, r.stdout.decode()
    assert out.read_bytes() == o.text()
    wav, out2 = tmp_path / "cap.wav", tmp_path / "file.txt"
    pdt.write_wav(str(wav), 48000, iq)
    r = subprocess.run([exe, "-l", "-P", "-o", str(out2), str(wav)], capture_output=True)
    assert r.returncode == 0, r.stdout.decode()
    assert out2.read_bytes() == orc.Oracle(orc.ARGOS, 48000, iq, chunk=2400, chain=1, keep_stages=False).text()
