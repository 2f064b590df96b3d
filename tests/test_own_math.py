"""The library's own restatements of the C-library functions the reference calls (csrc/pdt_device_math.h; the kernels run
the same code on the device) against the C library of this machine, bit for bit: glibc 2.35's double sincos / sin / cos
(table-driven IBM routines), sincosf, hypot (cabs) and hypotf (cabsf), over the argument ranges the chain produces and
well beyond.  Host only (pdt_host_math): runs without a GPU."""
import ctypes as C
import ctypes.util

import numpy as np
import pytest

libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
libm.sin.restype = libm.cos.restype = libm.hypot.restype = C.c_double
libm.sin.argtypes = libm.cos.argtypes = [C.c_double]
libm.hypot.argtypes = [C.c_double, C.c_double]
libm.sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
libm.sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
libm.hypotf.restype = C.c_float
libm.hypotf.argtypes = [C.c_float, C.c_float]


def args(lo, hi, n, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(lo, hi, n)
    x[::2] *= -1
    return x


RANGES = [(0.0, 2.0 ** -26), (1e-9, 0.13), (0.12, 0.86), (0.85, 2.43), (2.42, 6.3), (6.28, 60.0), (50.0, 1e5), (1e5, 1e8)]


@pytest.mark.parametrize("lo,hi", RANGES)
def test_double_sincos_sin_cos_equal_glibc(pdt, lo, hi):
    x = args(lo, hi, 60000, 11)
    s0, c0 = pdt.host_math(0, x)
    s1, _ = pdt.host_math(1, x)
    c2, _ = pdt.host_math(2, x)
    ws, wc, wsin, wcos = np.zeros_like(x), np.zeros_like(x), np.zeros_like(x), np.zeros_like(x)
    a, b = C.c_double(), C.c_double()
    for i, v in enumerate(x):
        libm.sincos(v, C.byref(a), C.byref(b))
        ws[i], wc[i] = a.value, b.value
        wsin[i], wcos[i] = libm.sin(v), libm.cos(v)
    for got, want in ((s0, ws), (c0, wc), (s1, wsin), (c2, wcos)):
        assert got.tobytes() == want.tobytes()


def test_special_points_of_the_double_routines(pdt):
    k = np.arange(0, 900)
    x = np.concatenate([k / 128.0, k / 128.0 + 2.0 ** -8, np.nextafter(k / 128.0 + 2.0 ** -8, 0), [0.126, 0.855469, 2.426265, 0.0, -0.0],
                        np.arange(1, 400) * (np.pi / 2), np.arange(1, 400) * np.float64(np.float32(np.pi))])
    s0, c0 = pdt.host_math(0, x)
    a, b = C.c_double(), C.c_double()
    for i, v in enumerate(x):
        libm.sincos(v, C.byref(a), C.byref(b))
        assert (np.float64(a.value).tobytes(), np.float64(b.value).tobytes()) == (s0[i].tobytes(), c0[i].tobytes()), v


def test_sincosf_equals_glibc(pdt):
    x = np.concatenate([args(0, 7.0, 150000, 5), args(0, 120.0, 50000, 6), args(0, 1e-4, 5000, 7)]).astype(np.float32).astype(np.float64)
    s, c = pdt.host_math(3, x)
    a, b = C.c_float(), C.c_float()
    for i, v in enumerate(x):
        libm.sincosf(C.c_float(v), C.byref(a), C.byref(b))
        assert a.value == s[i] and b.value == c[i], v


def test_hypot_equals_glibc(pdt):
    rng = np.random.default_rng(9)
    pcm = rng.integers(-32768, 32768, size=(120000, 2)) / 32768.0            # the values cabs sees: int16 / 32768
    free = rng.uniform(-1, 1, size=(60000, 2))
    edge = np.array([[0, 0], [0, 0.5], [0.25, 0], [1, 1], [-1, 1e-300], [3e-5, 3e-5], [1.0, 2.0 ** -53]])
    xy = np.concatenate([pcm, free, edge])
    h, _ = pdt.host_math(4, xy.reshape(-1))
    hf, _ = pdt.host_math(5, xy.astype(np.float32).astype(np.float64).reshape(-1))
    for i, (a, b) in enumerate(xy):
        assert libm.hypot(a, b) == h[i], (a, b)
        assert libm.hypotf(C.c_float(a), C.c_float(b)) == hf[i], (a, b)


def test_branch_free_sincosf_equals_the_library_form(pdt):
    """sincosf_flat (what k_mix_fir evaluates: no early-outs, the negated table as a sign flip) == sincosf_glibc == the C
    library: a dense sample of the PLL's phase range, every float below 2^-11 in steps, the quadrant boundaries and their
    neighbours, both zeros."""
    rng = np.random.default_rng(11)
    dense = rng.uniform(-7.0, 7.0, 400000).astype(np.float32)
    wide = rng.uniform(-119.9, 119.9, 100000).astype(np.float32)
    tiny = (np.arange(1, 0x39800000 + 0x400000, 9973, dtype=np.uint32)).view(np.float32)          # 0 < y < 2^-11
    k = np.arange(-80, 81)
    q = (k * (np.pi / 4)).astype(np.float32)
    edges = np.concatenate([q, np.nextafter(q, np.float32(-1000.0)), np.nextafter(q, np.float32(1000.0))])
    edges = np.concatenate([edges, np.array([0.0, -0.0, 0.5, -0.5, 2.0 ** -12, -(2.0 ** -12)], dtype=np.float32)])
    x = np.concatenate([dense, wide, tiny, -tiny, edges]).astype(np.float64)
    s3, c3 = pdt.host_math(3, x)
    s6, c6 = pdt.host_math(6, x)
    assert s3.astype(np.float32).tobytes() == s6.astype(np.float32).tobytes()
    assert c3.astype(np.float32).tobytes() == c6.astype(np.float32).tobytes()
    a, b = C.c_float(), C.c_float()
    for i in range(0, len(x), 97):
        libm.sincosf(C.c_float(x[i]), C.byref(a), C.byref(b))
        assert np.float32(a.value).tobytes() == np.float32(s6[i]).tobytes() and np.float32(b.value).tobytes() == np.float32(c6[i]).tobytes(), x[i]


def _every_float(lo, hi):
    a = np.arange(np.float32(lo).view(np.uint32), np.float32(hi).view(np.uint32) + 1, dtype=np.uint32).view(np.float32)
    return np.concatenate([a, -a])


def test_pll_wraps_in_fused_form_equal_the_reference_expressions(pdt):
    """One float PLL step's two wraps as the walkers evaluate them (pll_wrap_error_f32: one sign transfer + two fused
    multiply-adds; pll_wrap_phase_f32: k = trunc(p / 2pi), two fused multiply-adds, no select) against the reference's
    expressions -- compare promoted to double, correct by -+2 M_PI in double, narrow (CarrierTrackingPLL.c:168-188) -- over EVERY
    float of the ranges the corrections can fire in, plus a sample of the range they leave alone."""
    rng = np.random.default_rng(5)
    small = np.concatenate([rng.uniform(-3.2, 3.2, 200000).astype(np.float32), np.array([0.0, 1e-30, -1e-30, 3.1415925, -3.1415925], np.float32)])
    x = np.concatenate([_every_float(3.0, 9.5), small])
    xd = x.astype(np.float64)
    ref = np.where(xd > np.pi, (xd - 2 * np.pi).astype(np.float32), np.where(xd < -np.pi, (xd + 2 * np.pi).astype(np.float32), x))
    got, _ = pdt.host_math(7, xd)
    assert got.astype(np.float32).tobytes() == ref.astype(np.float32).tobytes()
    small = np.concatenate([rng.uniform(-6.3, 6.3, 200000).astype(np.float32), np.array([0.0, 1e-30, -1e-30, 6.283185, -6.283185], np.float32)])
    x = np.concatenate([_every_float(6.0, 12.5), small])         # (the one-correction variant is selected for |p| < 4pi - 0.05)
    xd = x.astype(np.float64)
    ref = np.where(xd > 2 * np.pi, (xd - 2 * np.pi).astype(np.float32), np.where(xd < -2 * np.pi, (xd + 2 * np.pi).astype(np.float32), x))
    got, _ = pdt.host_math(8, xd)
    assert got.astype(np.float32).tobytes() == ref.astype(np.float32).tobytes()
