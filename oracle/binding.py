"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")

POES, ARGOS = 0, 1
(ST_IQ, ST_TIME, ST_PLL, ST_LOCK, ST_FIR, ST_AGC, ST_SYM, ST_SYMT, ST_BITS, ST_BITT, ST_COUNTS, ST_TAPS,
 ST_SYMIDX, ST_AGC_RAW, ST_AVG) = range(15)


class OrcFrame(C.Structure):
    _fields_ = [("time", C.c_double), ("bit_index", C.c_int64), ("inverted", C.c_uint8), ("nbytes", C.c_uint8),
                ("complete", C.c_uint8), ("pad", C.c_uint8), ("bytes", C.c_uint8 * 104)]


class OrcTipFrame(C.Structure):
    _fields_ = [("minor_id", C.c_uint16), ("spacecraft", C.c_uint8), ("parity", C.c_uint8), ("checked", C.c_uint8),
                ("has_time", C.c_uint8), ("day", C.c_uint16), ("day_ms", C.c_int32)]


class OrcTipSummary(C.Structure):
    _fields_ = [("frames_checked", C.c_uint64), ("good_frames", C.c_uint64), ("good_chunks", C.c_uint64),
                ("bad_chunks", C.c_uint64), ("spacecraft", C.c_int32), ("day", C.c_int32), ("t0_ms", C.c_int64),
                ("time_frames", C.c_uint64)]


TIP_DTYPE = np.dtype([("minor_id", "<u2"), ("spacecraft", "u1"), ("parity", "u1"), ("checked", "u1"), ("has_time", "u1"),
                      ("day", "<u2"), ("day_ms", "<i4")])

_lib = None


def build():
    subprocess.run(["make", "-C", HERE, "liboracle.so", "oracle_demod"], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.orc_open.restype = C.c_void_p
        L.orc_open.argtypes = [C.c_int, C.c_uint, C.c_ulong, C.c_double, C.c_int]
        L.orc_close.argtypes = [C.c_void_p]
        L.orc_run_pcm16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_run_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_run_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
        L.orc_text.restype = C.c_void_p
        L.orc_text.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.orc_num_frames.restype = C.c_size_t
        L.orc_num_frames.argtypes = [C.c_void_p]
        L.orc_frames.restype = C.POINTER(OrcFrame)
        L.orc_frames.argtypes = [C.c_void_p]
        L.orc_stage.restype = C.c_size_t
        L.orc_stage.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.orc_norm_factor.restype = C.c_double
        L.orc_norm_factor.argtypes = [C.c_void_p]
        L.orc_lock_sample.restype = C.c_long
        L.orc_lock_sample.argtypes = [C.c_void_p]
        L.orc_lock_freq_hz.restype = C.c_double
        L.orc_lock_freq_hz.argtypes = [C.c_void_p]
        L.orc_interp.argtypes = [C.c_void_p]
        L.orc_ntaps.argtypes = [C.c_void_p]
        L.orc_totals.restype = C.c_size_t
        L.orc_totals.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 3
        L.orc_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_sincosf.restype = None
        L.orc_hypotf.argtypes = [C.c_float, C.c_float]
        L.orc_hypotf.restype = C.c_float
        L.orc_q_rsqrt.argtypes = [C.c_float]
        L.orc_q_rsqrt.restype = C.c_float
        L.orc_arctan2_f32.argtypes = [C.c_float, C.c_float]
        L.orc_arctan2_f32.restype = C.c_float
        L.orc_make_lpf_f32.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int]
        L.orc_make_lpf_f64.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int]
        L.orc_set_math_mode.argtypes = [C.c_int]
        L.orc_set_math_mode.restype = None
        L.orc_sincos_portable.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_sincos_portable.restype = None
        _lib = L
    return _lib


MATH_LIBM, MATH_PORTABLE = 0, 1


def bytesync(mode: int, bits: np.ndarray, piece: int = 4160):
    """Run only the byte synchroniser of the restatement on a uint8 array of '0'/'1' characters.
    Returns (text, frames); the time stamp of bit k is k."""
    L = lib()
    h = L.orc_open(mode, 50000, 0, 0.0, 0)
    b = np.ascontiguousarray(bits, dtype=np.uint8)
    L.orc_run_bits(h, b.ctypes.data, b.size, piece)
    n = C.c_size_t()
    p = L.orc_text(h, C.byref(n))
    text = C.string_at(p, n.value)
    nf = L.orc_num_frames(h)
    fp = L.orc_frames(h)
    frames = [(fp[i].time, fp[i].bit_index, fp[i].inverted, fp[i].nbytes, fp[i].complete, bytes(fp[i].bytes)) for i in range(nf)]
    L.orc_close(h)
    return text, frames


class Oracle:
    """Run the CPU restatement over a whole capture (int16[n,2]).

    math_mode (ARGOS/double only): MATH_LIBM = glibc sincos/hypot (bit-identical to the reference
    objects), MATH_PORTABLE = the plain-double evaluation the HIP kernels use."""

    def __init__(self, mode: int, sample_rate: int, iq: np.ndarray, chunk: int = 0, norm_override: float = 0.0,
                 keep_stages: bool = True, math_mode: int = MATH_LIBM, sampler: int = 0, mm_range: float = 3.0,
                 mm_kp: float = 0.15, chain: int = 0):
        L = lib()
        L.orc_set_math_mode(math_mode)
        self._L = L
        self.mode = mode
        self.dtype = np.float64 if (mode == ARGOS and not chain) else np.float32     # (the ARGOS twin is the float build)
        self._h = L.orc_open(mode, sample_rate, chunk, norm_override, int(keep_stages))
        L.orc_set_sampler.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        L.orc_set_sampler.restype = None
        L.orc_set_sampler(self._h, sampler, mm_range, mm_kp)      # 1 = MMClockRecovery instead of Gardner
        if chain:                                                 # 1 = the sound-card twin's constants / stage order
            L.orc_set_chain.argtypes = [C.c_void_p, C.c_int]
            L.orc_set_chain.restype = C.c_int
            if L.orc_set_chain(self._h, chain) != 0:
                raise ValueError("orc_set_chain failed")
        if np.asarray(iq).dtype.kind == "f":                     # RAW float32 capture
            a = np.ascontiguousarray(iq, dtype="<f4").reshape(-1)
            L.orc_run_f32(self._h, a.ctypes.data, a.size // 2)
        else:
            a = np.ascontiguousarray(iq, dtype="<i2").reshape(-1)
            L.orc_run_pcm16(self._h, a.ctypes.data, a.size // 2)
        L.orc_set_math_mode(MATH_LIBM)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_close(self._h)
            self._h = None

    def text(self) -> bytes:
        n = C.c_size_t()
        p = self._L.orc_text(self._h, C.byref(n))
        return C.string_at(p, n.value)

    def frames(self):
        n = self._L.orc_num_frames(self._h)
        p = self._L.orc_frames(self._h)
        return [p[i] for i in range(n)]

    def stage(self, st: int) -> np.ndarray:
        n = self._L.orc_stage(self._h, st, None, 0)
        dt = {ST_BITS: np.uint8, ST_COUNTS: np.uint64, ST_SYMIDX: np.int64}.get(st, self.dtype)
        out = np.zeros(n // np.dtype(dt).itemsize, dtype=dt)
        if n:
            self._L.orc_stage(self._h, st, out.ctypes.data, n)
        return out

    @property
    def norm_factor(self):
        return self._L.orc_norm_factor(self._h)

    @property
    def lock_sample(self):
        return self._L.orc_lock_sample(self._h)

    @property
    def lock_freq_hz(self):
        return self._L.orc_lock_freq_hz(self._h)

    @property
    def interp(self):
        return self._L.orc_interp(self._h)

    def totals(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        f = self._L.orc_totals(self._h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value, f


def tip_check(frames):
    """frames: list of (time, bytes-like of 104, complete) or an array of OrcFrame -> (summary dict, records array).
    Restatement of the reference's MATLAB checkParity.m / daytimeDecode.m (oracle_tip.c)."""
    L = lib()
    L.orc_tip_check.restype = None
    L.orc_tip_check.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    n = len(frames)
    arr = (OrcFrame * max(n, 1))()
    for i, fr in enumerate(frames):
        if isinstance(fr, OrcFrame):
            arr[i] = fr
        else:
            t, b, complete = fr
            arr[i].time = float(t)
            arr[i].nbytes = len(b)
            arr[i].complete = 1 if complete else 0
            for k, v in enumerate(bytes(b)[:104]):
                arr[i].bytes[k] = v
    out = np.zeros(max(n, 1), dtype=TIP_DTYPE)
    sm = OrcTipSummary()
    L.orc_tip_check(C.byref(arr), n, out.ctypes.data, C.byref(sm))
    return {k: int(getattr(sm, k)) for k, _ in OrcTipSummary._fields_}, out[:n]
