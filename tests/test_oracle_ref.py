"""Oracle restatement vs the reference's own compiled DSP objects (oracle/_ref), stage by stage.
Skipped where oracle/_ref was not built (no /root/reference and no prebuilt files)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

REF_POES = os.path.join(ROOT, "oracle/_ref/ref_demodPOES")
REF_ARGOS = os.path.join(ROOT, "oracle/_ref/ref_demodARGOS")
pytestmark = pytest.mark.skipif(not (os.path.exists(REF_POES) and os.path.exists(REF_ARGOS)),
                                reason="oracle/_ref not built (make -C oracle ref)")

STAGES = {"pll": 2, "lock": 3, "fir": 4, "agc": 5, "sym": 6, "symt": 7, "bits": 8, "bitt": 9, "taps": 11, "iq": 0,
          "time": 1, "agcraw": 13, "avg": 14}          # agcraw: AGC output before Squelch (ARGOS), what -r writes to output.raw; avg: CarrierTrackPLL's return value per pass


def run_ref(binary, wav, tmp_path, extra=()):
    out = tmp_path / "ref.txt"
    dump = tmp_path / "refdump"
    subprocess.run([binary, *extra, "-d", str(dump), str(wav), str(out)], check=True, capture_output=True)
    text = out.read_bytes() if out.exists() else b""
    return text, dump


def compare_all(o, dump):
    for name, sid in STAGES.items():
        path = f"{dump}.{name}"
        if not os.path.exists(path):
            continue
        ref = open(path, "rb").read()
        if name == "lock" and o.mode == 0:
            continue
        assert o.stage(sid).tobytes() == ref, f"stage {name}: restatement differs from the reference objects"


@pytest.mark.parametrize("chunk", [10000, 3333, 777])
def test_clip_all_stages(orc, clip, tmp_path, chunk):
    rate, iq = clip
    text, dump = run_ref(REF_POES, os.path.join(GOLDEN, "5sec_clip.wav"), tmp_path, ["-c", str(chunk)])
    o = orc.Oracle(orc.POES, rate, iq, chunk=chunk)
    compare_all(o, dump)
    assert o.text() == text


@pytest.mark.parametrize("fs,seed,f0", [(50000, 5, -2300.0), (250000, 6, 3100.0), (32000, 7, 400.0)])
def test_synthetic_poes_all_stages(orc, pdt, tmp_path, fs, seed, f0):
    iq = pdt.synth_capture(0, fs, 4.0, f0_hz=f0, seed=seed)
    wav = tmp_path / "s.wav"
    pdt.write_wav(str(wav), fs, iq)
    text, dump = run_ref(REF_POES, wav, tmp_path)
    o = orc.Oracle(orc.POES, fs, iq)
    compare_all(o, dump)
    assert o.text() == text and len(text) > 0


def test_exact_multiple_of_chunk(orc, pdt, tmp_path):
    """Q7: data length an exact multiple of the chunk -> one extra empty iteration in the reference."""
    fs = 50000
    iq = pdt.synth_capture(0, fs, 2.0, seed=3)          # 100000 = 10 chunks exactly
    wav = tmp_path / "s.wav"
    pdt.write_wav(str(wav), fs, iq)
    text, dump = run_ref(REF_POES, wav, tmp_path)
    o = orc.Oracle(orc.POES, fs, iq)
    compare_all(o, dump)
    assert o.text() == text


@pytest.mark.parametrize("chunk", [2400, 1000, 2401])
def test_synthetic_argos_all_stages(orc, pdt, tmp_path, chunk):
    iq = pdt.synth_capture(1, 32000, 7.0, f0_hz=160.0, seed=11)
    wav = tmp_path / "a.wav"
    pdt.write_wav(str(wav), 32000, iq)
    text, dump = run_ref(REF_ARGOS, wav, tmp_path, ["-c", str(chunk)])
    o = orc.Oracle(orc.ARGOS, 32000, iq, chunk=chunk)
    compare_all(o, dump)
    assert o.text() == text and len(text) > 0


def test_noise_only_no_frames(orc, tmp_path, pdt):
    rng = np.random.default_rng(1)
    iq = rng.integers(-300, 300, size=(60000, 2)).astype(np.int16)
    wav = tmp_path / "n.wav"
    pdt.write_wav(str(wav), 50000, iq)
    text, dump = run_ref(REF_POES, wav, tmp_path)
    o = orc.Oracle(orc.POES, 50000, iq)
    compare_all(o, dump)
    assert o.text() == text


@pytest.mark.parametrize("seed,f0,secs,chunk", [(99, 120.0, 13.0, 2400), (12, -90.0, 20.0, 1000), (13, 60.0, 24.0, 2401),
                                                (14, 199.0, 16.0, 4800)])
def test_argos_portable_math_keeps_the_reference_output(orc, pdt, tmp_path, seed, f0, secs, chunk):
    """The GPU evaluates the double sine/cosine with plain IEEE operations instead of glibc's
    table-driven routine.  The oracle's matching "portable" mode must still reproduce the
    reference's symbol picks, bits and packet file exactly (the float streams differ in last bits)."""
    iq = pdt.synth_capture(1, 32000, secs, f0_hz=f0, seed=seed)
    wav = tmp_path / "a.wav"
    pdt.write_wav(str(wav), 32000, iq)
    text, dump = run_ref(REF_ARGOS, wav, tmp_path, ["-c", str(chunk)])
    o = orc.Oracle(orc.ARGOS, 32000, iq, chunk=chunk, math_mode=orc.MATH_PORTABLE)
    assert o.text() == text and len(text) > 0
    assert o.stage(orc.ST_BITS).tobytes() == open(f"{dump}.bits", "rb").read()
    ref_sym = np.fromfile(f"{dump}.sym", dtype=np.float64)
    sym = o.stage(orc.ST_SYM)
    assert len(sym) == len(ref_sym)
    assert np.allclose(sym, ref_sym, rtol=1e-9, atol=1e-12)          # same picks, last-bit differences only
    # and the portable sine/cosine itself is a <1 ulp evaluation
    import ctypes as C, math
    s, c = C.c_double(), C.c_double()
    rng = np.random.default_rng(0)
    for x in rng.uniform(-2 * np.pi, 2 * np.pi, 20000):
        orc.lib().orc_sincos_portable(float(x), C.byref(s), C.byref(c))
        assert abs(s.value - math.sin(x)) <= 1.2e-16 * max(abs(math.sin(x)), 1e-300) * 2 or abs(s.value - math.sin(x)) < 2.3e-16
        assert abs(c.value - math.cos(x)) < 2.3e-16


@pytest.mark.parametrize("scale", [1.0, 37.5, 0.004])
def test_raw_float32_input(orc, pdt, tmp_path, scale):
    """RAW path (GetComplexRawChunk, wave.c:413-540; POESTIPdemod/main.c:313-339): interleaved float32
    I,Q used without normalisation, rate from -s in kHz."""
    iq = pdt.synth_capture(0, 50000, 4.0, seed=5)
    raw = (iq.astype(np.float32) / np.float32(32768.0)) * np.float32(scale)
    path = tmp_path / "cap.raw"
    raw.tofile(path)
    text, dump = run_ref(REF_POES, path, tmp_path, ["-s", "50"])
    o = orc.Oracle(orc.POES, 50000, raw)
    compare_all(o, dump)
    assert o.text() == text and len(text) > 0
    if scale == 1.0:
        # the same samples as a WAV give the same file
        assert text == orc.Oracle(orc.POES, 50000, iq, keep_stages=False).text()


@pytest.mark.parametrize("mode,chunk,rng_kp", [("poes", 10000, (3.0, 0.15)), ("poes", 3333, (9.0, 0.05)), ("argos", 2400, (3.0, 0.15)),
                                               ("argos", 1000, (1.0, 0.3))])
def test_mm_clock_recovery_all_stages(orc, pdt, clip, tmp_path, mode, chunk, rng_kp):
    """MMClockRecovery (common/MMClockRecovery.c:5-83) at the sampler's call site -- the one-line
    switch the reference keeps commented out (ARGOSdemod/main.c:277).  Restatement vs the reference's own object,
    every stage, for the defaults of that call (stepRange 3, kp 0.15) and another pair."""
    rg, kp = rng_kp
    extra = ["-c", str(chunk), "-M", "-R", str(rg), "-K", str(kp)]
    if mode == "poes":
        rate, iq = clip
        text, dump = run_ref(REF_POES, os.path.join(GOLDEN, "5sec_clip.wav"), tmp_path, extra)
        o = orc.Oracle(orc.POES, rate, iq, chunk=chunk, sampler=1, mm_range=rg, mm_kp=kp)
    else:
        iq = pdt.synth_capture(1, 32000, 9.0, f0_hz=140.0, seed=21)
        wav = tmp_path / "a.wav"
        pdt.write_wav(str(wav), 32000, iq)
        text, dump = run_ref(REF_ARGOS, wav, tmp_path, extra)
        o = orc.Oracle(orc.ARGOS, 32000, iq, chunk=chunk, sampler=1, mm_range=rg, mm_kp=kp)
    compare_all(o, dump)
    assert o.text() == text
    assert len(o.stage(orc.ST_SYM)) > 1000


@pytest.mark.parametrize("fs,chunk,seed,f0,scale", [(48000, 2400, 31, 900.0, 1.0), (48000, 2400, 32, -2800.0, 0.02), (48000, 1000, 33, 300.0, 1.0),
                                                    (50000, 10000, 34, 1500.0, 3.0)])
def test_live_chain_all_stages(orc, pdt, tmp_path, fs, chunk, seed, f0, scale):
    """the sound-card twin's chain (POESTIPdemodPortAudio/main.c:324-393) -- float32 blocks of
    2400 frames at 48 kHz, acquisition gain 198.9437, lock threshold 0.10, Squelch(0.05) between PLL and FIR,
    Manchester threshold 0.75.  The twin cannot be built here (PortAudio); its stage functions are the common
    objects, which oracle/ref_driver.c -L calls in the twin's order with the twin's constants."""
    iq = pdt.synth_capture(0, fs, 5.0, f0_hz=f0, seed=seed)
    raw = (iq.astype(np.float32) / np.float32(32768.0)) * np.float32(scale)
    path = tmp_path / "live.raw"
    raw.tofile(path)
    text, dump = run_ref(REF_POES, path, tmp_path, ["-L", "-s", str(fs / 1000.0), "-c", str(chunk)])
    o = orc.Oracle(orc.POES, fs, raw, chunk=chunk, chain=1)
    for name, sid in STAGES.items():
        ref = open(f"{dump}.{name}", "rb").read()
        assert o.stage(sid).tobytes() == ref, f"stage {name}: restatement differs from the reference objects"
    assert o.text() == text and len(text) > 0
    # the squelch really acts before the lock: the PLL output starts as zeros
    pll = o.stage(orc.ST_PLL)
    assert o.lock_sample > 0 and not pll[: min(o.lock_sample, 200)].any() and pll[o.lock_sample + 5000:].any()
    # and it is not the file chain
    assert o.stage(orc.ST_PLL).tobytes() != orc.Oracle(orc.POES, fs, raw, chunk=chunk).stage(orc.ST_PLL).tobytes()


@pytest.mark.parametrize("name,args", [("clip.c10000", ["-c", "10000"]), ("clip.c1000", ["-c", "1000"])])
def test_progress_golden_is_what_the_reference_objects_print(tmp_path, name, args):
    """tests/golden/*.progress / *.avg.* (the chunk loop's progress line and CarrierTrackPLL's return values, used by
    tests/test_gpu_quality.py) are what the reference's objects produce here"""
    text, dump = run_ref(REF_POES, os.path.join(GOLDEN, "5sec_clip.wav"), tmp_path, args)
    assert open(f"{dump}.progress", "rb").read() == open(os.path.join(GOLDEN, name + ".progress"), "rb").read()
    assert open(f"{dump}.avg", "rb").read() == open(os.path.join(GOLDEN, name + ".avg.f32"), "rb").read()
    assert text == open(os.path.join(GOLDEN, name + ".txt"), "rb").read()


def test_progress_golden_synthetic(pdt, tmp_path):
    iq = pdt.synth_capture(0, 50000, 3.0, seed=1234)                  # 15 chunks exactly: the loop's extra pass
    wav = tmp_path / "p.wav"
    pdt.write_wav(str(wav), 50000, iq)
    _, dump = run_ref(REF_POES, wav, tmp_path)
    assert open(f"{dump}.progress", "rb").read() == open(os.path.join(GOLDEN, "poes_50000.progress"), "rb").read()
    avg = np.fromfile(f"{dump}.avg", dtype="<f4")
    assert len(avg) == 16 and avg[15].tobytes() == avg[14].tobytes()
    a = pdt.synth_capture(1, 32000, 13.0, seed=99)
    wav = tmp_path / "a.wav"
    pdt.write_wav(str(wav), 32000, a)
    _, dump = run_ref(REF_ARGOS, wav, tmp_path)
    assert open(f"{dump}.progress", "rb").read() == open(os.path.join(GOLDEN, "argos_32000.progress"), "rb").read()
    assert open(f"{dump}.avg", "rb").read() == open(os.path.join(GOLDEN, "argos_32000.avg.f64"), "rb").read()


REF_ARGOSF = os.path.join(ROOT, "oracle/_ref/ref_demodARGOSf")


@pytest.mark.skipif(not os.path.exists(REF_ARGOSF), reason="oracle/_ref/ref_demodARGOSf not built")
@pytest.mark.parametrize("fs,chunk,seed,f0,secs", [(48000, 2400, 21, 130.0, 9.0), (48000, 2401, 22, -75.0, 8.0), (48000, 2403, 24, 40.0, 8.0),
                                                   (32000, 1000, 23, 160.0, 7.0), (48000, 2402, 25, 99.0, 8.0)])
def test_argos_sound_card_twin_all_stages(orc, pdt, tmp_path, fs, chunk, seed, f0, secs):
    """The ARGOS sound-card twin (ARGOSdemodPortAudio/main.c:266-329): the FLOAT build of the ARGOS chain's stage functions
    (its config.h), its own ByteSync.c (inverse sync word enabled, no "i" on the stamp), its own time stamps and buffer order.
    The restatement (chain = 1) equals those objects at every stage, the reads past the block included: chunk sizes whose
    4-byte buffers leave 0, 1, 2 and 3 floats of malloc slack in front of the next chunk's size field."""
    iq = pdt.synth_capture(1, fs, secs, f0_hz=f0, seed=seed)
    wav = tmp_path / "a.wav"
    pdt.write_wav(str(wav), fs, iq)
    text, dump = run_ref(REF_ARGOSF, wav, tmp_path, ["-c", str(chunk)])
    o = orc.Oracle(orc.ARGOS, fs, iq, chunk=chunk, chain=1)
    assert o.dtype == np.float32
    compare_all(o, dump)
    assert o.text() == text and len(text) > 0
