"""The capture readers of the link-compatible shim (wave.h:27-29 -- ReadWavHeader, GetComplexWaveChunk, GetComplexRawChunk in
libpdt_compat_{poes,argos}.so, host only) against the reference's own objects (oracle/_ref/libref_{poes,argos}.so): headers,
samples and the running-sum time axis (Q1) bit for bit, chunk after chunk and file after file (the time axis lives in function
statics and is never reset), in both DECIMAL_TYPE builds; the truncations of Q5 (32-bit and 8-bit PCM) and the RAW float reader
included.  No GPU: the readers never touch one.  Skipped where oracle/_ref was not built."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

CSRC = os.path.join(ROOT, "project-desert-tortoise_amd", "csrc")
PAIRS = [("libpdt_compat_poes.so", "libref_poes.so", np.float32, np.complex64),
         ("libpdt_compat_argos.so", "libref_argos.so", np.float64, np.complex128)]
pytestmark = pytest.mark.skipif(not all(os.path.exists(os.path.join(ROOT, "oracle", "_ref", p[1])) and os.path.exists(os.path.join(CSRC, p[0]))
                                        for p in PAIRS), reason="oracle/_ref or the compat libraries not built")


class HEADER(C.Structure):
    _fields_ = [("riff", C.c_ubyte * 4), ("overall_size", C.c_uint), ("wave", C.c_ubyte * 4), ("fmt_chunk_marker", C.c_ubyte * 4),
                ("length_of_fmt", C.c_uint), ("format_type", C.c_uint), ("channels", C.c_uint), ("sample_rate", C.c_uint),
                ("byterate", C.c_uint), ("block_align", C.c_uint), ("bits_per_sample", C.c_uint), ("data_chunk_header", C.c_ubyte * 4),
                ("data_size", C.c_uint), ("type", C.c_ubyte)]


libc = C.CDLL(None)
libc.fopen.restype = C.c_void_p
libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
libc.fclose.argtypes = [C.c_void_p]
libc.ftell.argtypes = [C.c_void_p]
libc.ftell.restype = C.c_long


def load(path):
    L = C.CDLL(path)
    L.ReadWavHeader.restype = HEADER
    L.ReadWavHeader.argtypes = [C.c_void_p]
    for f in (L.GetComplexWaveChunk, L.GetComplexRawChunk):
        f.restype = C.c_ulong
        f.argtypes = [C.c_void_p, HEADER, C.c_void_p, C.c_void_p, C.c_ulong]
    return L


def pcm_wav(path, rate, bits, payload):
    blk = 2 * bits // 8
    hdr = b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 2, rate, rate * blk, blk, bits)
    hdr += b"data" + struct.pack("<I", len(payload))
    open(path, "wb").write(hdr + payload)


def read_all(L, path, real, cplx, chunk, raw_rate=0):
    """the file chunk by chunk, as the mains do: [(n, samples, times)], the header's bytes, the position behind the header"""
    f = libc.fopen(path.encode(), b"rb")
    if raw_rate:
        h = HEADER()
        h.channels, h.bits_per_sample, h.sample_rate, h.format_type, h.type = 2, 32, raw_rate, 1, 1
        pos = 0
    else:
        h = L.ReadWavHeader(f)
        pos = libc.ftell(f)
    out = []
    while True:
        data = np.full(chunk, -7, dtype=cplx)
        t = np.full(chunk, -7, dtype=real)
        n = (L.GetComplexRawChunk if raw_rate else L.GetComplexWaveChunk)(f, h, data.ctypes.data, t.ctypes.data, chunk)
        out.append((int(n), data.tobytes(), t.tobytes()))
        if n < chunk:
            break
    libc.fclose(f)
    return out, bytes(h), pos


@pytest.mark.parametrize("pair", PAIRS, ids=["float", "double"])
def test_readers_equal_the_reference_objects(pair, tmp_path):
    mine, ref = load(os.path.join(CSRC, pair[0])), load(os.path.join(ROOT, "oracle", "_ref", pair[1]))
    real, cplx = pair[2], pair[3]
    rng = np.random.default_rng(5)
    files = [(os.path.join(GOLDEN, "5sec_clip.wav"), 10000, 0), (os.path.join(GOLDEN, "5sec_clip.wav"), 3333, 0)]
    p16 = str(tmp_path / "exact.wav")                      # a multiple of the chunk: the extra pass with zero samples (Q7)
    pcm_wav(p16, 250000, 16, rng.integers(-32768, 32767, 4 * 5000, dtype=np.int16).tobytes())
    files.append((p16, 2500, 0))
    p32 = str(tmp_path / "wide.wav")                       # Q5: 32-bit PCM goes through an int16_t
    pcm_wav(p32, 48000, 32, rng.integers(-2**31, 2**31 - 1, 2 * 777, dtype=np.int64).astype("<i4").tobytes())
    files.append((p32, 500, 0))
    p8 = str(tmp_path / "narrow.wav")                      # Q5: 8-bit PCM reads the pair's first byte for both channels
    pcm_wav(p8, 32000, 8, rng.integers(0, 255, 2 * 901, dtype=np.uint8).tobytes())
    files.append((p8, 400, 0))
    praw = str(tmp_path / "cap.raw")                       # RAW: 32-bit floats as they are, a file that ends inside a pair
    open(praw, "wb").write(rng.standard_normal(2 * 1234).astype("<f4").tobytes() + b"\x01\x02\x03")
    files.append((praw, 500, 50000))
    files.append((os.path.join(GOLDEN, "5sec_clip.wav"), 10000, 0))          # once more: the time axis goes on from where it stood
    for path, chunk, raw_rate in files:
        a, ha, pa = read_all(mine, path, real, cplx, chunk, raw_rate)
        b, hb, pb = read_all(ref, path, real, cplx, chunk, raw_rate)
        assert ha == hb and pa == pb, path
        assert [x[0] for x in a] == [x[0] for x in b], path
        for (n, da, ta), (_, db, tb) in zip(a, b):
            assert da == db and ta == tb, (path, chunk)


def test_a_program_linked_against_the_shim_needs_no_reference_object():
    """oracle/_ref/compat_demod* (ref_driver.c over the shim): every symbol it imports comes from libpdt_compat / libpdt / libc /
    libm -- wave.o of the reference is no longer linked (VERDICT r4 missing #5)."""
    import subprocess
    for exe, lib in (("compat_demodPOES", "libpdt_compat_poes.so"), ("compat_demodARGOS", "libpdt_compat_argos.so")):
        path = os.path.join(ROOT, "oracle", "_ref", exe)
        if not os.path.exists(path):
            pytest.skip("compat_demod* not built")
        und = subprocess.run(["nm", "--undefined-only", path], capture_output=True, text=True).stdout
        defd = subprocess.run(["nm", "--defined-only", path], capture_output=True, text=True).stdout
        for sym in ("ReadWavHeader", "GetComplexWaveChunk", "CarrierTrackPLL", "GardenerClockRecovery") + (("GetComplexRawChunk",) if "POES" in exe else ()):
            assert f" U {sym}" in und and f" T {sym}" not in defd, (exe, sym)
        exported = subprocess.run(["nm", "-D", "--defined-only", os.path.join(CSRC, lib)], capture_output=True, text=True).stdout
        for sym in ("ReadWavHeader", "GetComplexWaveChunk", "GetComplexRawChunk"):
            assert f" T {sym}" in exported
