"""The second form of the drop-in boundary: the reference's own stage prototypes (common/*.h) over the HIP kernels.

oracle/_ref/compat_demod{POES,ARGOS} is oracle/ref_driver.c -- the reference's chunk loop with its buffers, time arrays and
call-site constants -- linked against libpdt_compat_{poes,argos}.so (project-desert-tortoise_amd/host/pdt_compat.c) in place of
the reference's DSP objects (only wave.c's file reader still comes from the reference).  Its output file must be the reference
program's, byte for byte.  The binaries are built where /root/reference exists and travel to the GPU box.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden_text

pytestmark = pytest.mark.gpu

REF = os.path.join(ROOT, "oracle", "_ref")


def _need(name):
    path = os.path.join(REF, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not built (needs the reference tree at build time)")
    return path


def _run(exe, args, out):
    r = subprocess.run([exe] + args + [out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r


@pytest.mark.parametrize("chunk,golden", [(10000, "clip.c10000.txt"), (1000, "clip.c1000.txt"), (3333, "")])
def test_reference_loop_over_the_shim_reproduces_the_clip(tmp_path, chunk, golden):
    exe = _need("compat_demodPOES")
    out = str(tmp_path / "mf.txt")
    r = _run(exe, ["-c", str(chunk), os.path.join(GOLDEN, "5sec_clip.wav")], out)
    want = golden_text(golden) if golden else None
    ref_out = str(tmp_path / "ref.txt")
    _run(_need("ref_demodPOES"), ["-c", str(chunk), os.path.join(GOLDEN, "5sec_clip.wav")], ref_out)
    got = open(out, "rb").read()
    assert got == open(ref_out, "rb").read()
    if want is not None:
        assert got == want
    assert "Normalization Factor" in r.stdout and " : PLL locked at -3466.19Hz" in r.stdout          # the console surface


@pytest.mark.parametrize("fs,secs,chunk", [(250000, 6.0, 10000), (50000, 12.0, 777), (100000, 5.0, 10000)])
def test_shim_on_synthetic_poes_captures(pdt, tmp_path, fs, secs, chunk):
    exe = _need("compat_demodPOES")
    iq = pdt.synth_capture(0, fs, secs, seed=900 + chunk)
    wav = str(tmp_path / "cap.wav")
    pdt.write_wav(wav, fs, iq)
    out, ref_out = str(tmp_path / "mf.txt"), str(tmp_path / "ref.txt")
    _run(exe, ["-c", str(chunk), wav], out)
    _run(_need("ref_demodPOES"), ["-c", str(chunk), wav], ref_out)
    got = open(out, "rb").read()
    assert got == open(ref_out, "rb").read() and got.count(b"\n") > 30


def test_shim_with_the_mm_sampler(pdt, tmp_path):
    exe = _need("compat_demodPOES")
    wav = os.path.join(GOLDEN, "5sec_clip.wav")
    out, ref_out = str(tmp_path / "mf.txt"), str(tmp_path / "ref.txt")
    _run(exe, ["-M", wav], out)
    _run(_need("ref_demodPOES"), ["-M", wav], ref_out)
    assert open(out, "rb").read() == open(ref_out, "rb").read()


@pytest.mark.parametrize("chunk", [2400, 2401])
def test_shim_argos(pdt, tmp_path, chunk):
    """double build; chunk 2400 / 2401 = both alignments of the heap's size field behind the caller's buffers (Q16): the shim's
    sampler reads the caller's memory past the chunk exactly as the reference's does"""
    exe = _need("compat_demodARGOS")
    iq = pdt.synth_capture(1, 32000, 25.0, f0_hz=120.0, seed=17)
    wav = str(tmp_path / "argos.wav")
    pdt.write_wav(wav, 32000, iq)
    out, ref_out = str(tmp_path / "pk.txt"), str(tmp_path / "ref.txt")
    _run(exe, ["-c", str(chunk), wav], out)
    _run(_need("ref_demodARGOS"), ["-c", str(chunk), wav], ref_out)
    got = open(out, "rb").read()
    assert got == open(ref_out, "rb").read() and got.count(b"\n") >= 5
