// pdt_device_math.h -- scalar math of the PLL / AGC stages on gfx950, and (the same functions, host side) of the tap generator.
//
// Every routine reproduces, operation for operation, what the reference's CPU
// build evaluates (x86-64, gcc -O2, no contraction, glibc 2.35 libm), so that a
// lane on the GPU walks exactly the same float trajectory as the CPU:
//   * the file is compiled with -ffp-contract=off; the only fused operations are
//     the explicit __builtin_fma calls that mirror the FMA build of glibc's
//     sincosf (sysdeps/ieee754/flt-32/s_sincosf.c, x86-64 ifunc variant);
//   * float division and double sqrt are the correctly rounded HIP defaults;
//   * float denormals are preserved (HIP default on gfx9).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pdt {

#define PDT_PI 3.14159265358979323846

__host__ __device__ __forceinline__ uint32_t bits_of(float v) { uint32_t u; __builtin_memcpy(&u, &v, 4); return u; }
__host__ __device__ __forceinline__ uint64_t bits_of(double v) { uint64_t u; __builtin_memcpy(&u, &v, 8); return u; }
__host__ __device__ __forceinline__ float float_of(uint32_t u) { float v; __builtin_memcpy(&v, &u, 4); return v; }

// ---- sincosf: glibc 2.35 algorithm (reference call site CarrierTrackingPLL.c:106-107)
struct SinCosPoly { double c0, c1, c2, c3, c4, s1, s2, s3; };

__host__ __device__ __forceinline__ void sincosf_eval(double x, double x2, bool neg, int n, float &sinv, float &cosv)
{
    // coefficients of the +cos table; the -cos table negates the c's
    const double c0 = neg ? -0x1p0 : 0x1p0;
    const double c1 = neg ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
    const double c2 = neg ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5;
    const double c3 = neg ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
    const double c4 = neg ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double x3 = x * x2;
    const double x4 = x2 * x2;
    const double s1v = __builtin_fma(x2, s3, s2);
    const double c2v = __builtin_fma(x2, c4, c3);
    const double c1v = __builtin_fma(x2, c1, c0);
    const double x5 = x2 * x3;
    const double x6 = x2 * x4;
    const double s = __builtin_fma(x3, s1, x);
    const double c = __builtin_fma(x4, c2, c1v);
    const float sv = (float)__builtin_fma(s1v, x5, s);
    const float cv = (float)__builtin_fma(c2v, x6, c);
    sinv = (n & 1) ? cv : sv;
    cosv = (n & 1) ? sv : cv;
}

// valid for |y| < 120 (the PLL phase lives in (-2pi, 2pi])
__host__ __device__ __forceinline__ void sincosf_glibc(float y, float &sinv, float &cosv)
{
    const double x = (double)y;
    const uint32_t top = (bits_of(y) >> 20) & 0x7ffu;
    if (top < 0x3f4u) {
        if (top < 0x398u) {
            sinv = y;
            cosv = 1.0f;
            return;
        }
        sincosf_eval(x, x * x, false, 0, sinv, cosv);
    } else {
        const double r = x * 0x1.45f306dc9c883p+23;
        const int n = ((int32_t)r + 0x800000) >> 24;
        const double xr = __builtin_fma(-(double)n, 0x1.921fb54442d18p+0, x);
        const double sg = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        sincosf_eval(xr * sg, xr * xr, (n & 2) != 0, n, sinv, cosv);
    }
}

// sincosf_glibc without its branches, for the kernels that evaluate it for every sample of a capture: every argument goes
// through the quadrant reduction.  Same bits for every |y| < 120: below pi/4 the quadrant is 0 and the reduction hands back x
// itself (fma(-0, pi/2, x) = x), which is all the routine's own shortcut for |y| < 0.5 saves; below 2^-12, where the routine
// returns (y, 1) without evaluating anything, the polynomials round to exactly that (|x^3/6| < ulp(x)/4, x^2/2 < 2^-25).
// The table with the negated cosine coefficients (quadrants 2, 3) becomes a negation of the result -- every
// intermediate of that polynomial changes sign, rounding is symmetric --, the sign[] factor of the sine argument a flip of
// its sign bit.  tests/test_own_math.py compares the two forms (and the C library) over the whole range.
__host__ __device__ __forceinline__ void sincosf_flat(float y, float &sinv, float &cosv)
{
    const double x = (double)y;
    const double r = x * 0x1.45f306dc9c883p+23;
    const int n = ((int32_t)r + 0x800000) >> 24;
    const double xr = __builtin_fma(-(double)n, 0x1.921fb54442d18p+0, x);
    const double x2 = xr * xr;
    uint64_t xb = bits_of(xr);
    xb ^= (uint64_t)(((uint32_t)(n + 1) << 30) & 0x80000000u) << 32;         // quadrants 1, 2: the sine polynomial takes -x
    double xs;
    __builtin_memcpy(&xs, &xb, 8);
    const double c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double x3 = xs * x2;
    const double x4 = x2 * x2;
    const double s1v = __builtin_fma(x2, s3, s2);
    const double c2v = __builtin_fma(x2, c4, c3);
    const double c1v = __builtin_fma(x2, c1, 0x1p0);
    const double x5 = x2 * x3;
    const double x6 = x2 * x4;
    const double s = __builtin_fma(x3, s1, xs);
    const double c = __builtin_fma(x4, c2, c1v);
    float sv = (float)__builtin_fma(s1v, x5, s);
    sv = (y == 0.0f) ? y : sv;                                                    // (sin(-0) = -0: the sums above give +0)
    const float cp = (float)__builtin_fma(c2v, x6, c);
    const float cv = float_of(bits_of(cp) ^ (((uint32_t)n << 30) & 0x80000000u));   // quadrants 2, 3: the negated table
    sinv = (n & 1) ? cv : sv;
    cosv = (n & 1) ? sv : cv;
}

// ---- arctan2 approximation of CarrierTrackingPLL.c:15-40
// The two wraps of one float PLL step (CarrierTrackingPLL.c:168-188) in the form the walkers evaluate.  hi = (float)(2pi) lies
// above 2pi, d = hi - 2pi; x -+ hi is exact for pi <= |x| <= 4pi (Sterbenz) and adding -+d to it rounds once, to the float the
// reference's double expression narrows to (every float of the range: tests/test_oracle_math.py::test_unwrap_f32_exhaustive).
//   pll_wrap_error_f32: "if (d > M_PI) d -= 2 M_PI; else if (d < -M_PI) d += 2 M_PI" -- the promoted comparison is a float
//       comparison with (float)pi (PiCmp); the sign is transferred once (s = +-1) and both corrections are fused multiply-adds
//       with an exact product, i.e. the same two roundings as the subtraction and the addition they stand for.
//   pll_wrap_phase_f32: "while (p > 2 M_PI) p -= 2 M_PI; while (p < -2 M_PI) p += 2 M_PI" for |p| < 4pi - 0.05 (one correction
//       at most; the host selects the looping variant otherwise).  k = trunc(p * (float)(1 / 2pi)) is 0 below (float)(2pi) and
//       +-1 from there on: (float)(2pi) times the constant rounds to exactly 1, its predecessor to 1 - 2^-24, and the rounded
//       product is monotonic; fma(k, d, fma(k, -hi, p)) then IS the selected value: p itself for k = 0.  (For p = -0 it would
//       return +0; a state of the loop is never -0: round-to-nearest sums give -0 only from (-0) + (-0), and the loop starts
//       at +0.)
// tests/test_own_math.py compares both with the plain forms over every float of their ranges.
__host__ __device__ __forceinline__ float pll_wrap_error_f32(float x)
{
    const float hi = 6.2831854820251465f, d = 1.7484555314695172e-07f;
#if defined(__HIP_DEVICE_COMPILE__)
    // The same five operations in a fixed order: the compare first, the select last.  Left to the scheduler the compare lands
    // right in front of the select (the step is one dependent chain, nothing else wants the slot) and the two wait states a
    // vector compare needs before its mask is read cost an s_nop -- one issue slot of a lone wavefront's 17 per step.
    float wrapped, s, r;
    asm("v_cmp_ge_f32_e64 vcc, |%3|, %4\n\t"
        "v_bfi_b32 %1, %5, 1.0, %3\n\t"
        "v_fma_f32 %0, %1, %6, %3\n\t"
        "v_fma_f32 %0, %1, %7, %0\n\t"
        "v_cndmask_b32_e32 %2, %3, %0, vcc"
        : "=&v"(wrapped), "=&v"(s), "=&v"(r)
        : "v"(x), "s"(3.14159274101257324f), "s"(0x7fffffffu), "s"(-hi), "s"(d)
        : "vcc");
    return r;
#else
    const float s = __builtin_copysignf(1.0f, x);
    const float wrapped = __builtin_fmaf(s, d, __builtin_fmaf(s, -hi, x));
    return (__builtin_fabsf(x) >= 3.14159274101257324f) ? wrapped : x;
#endif
}
__host__ __device__ __forceinline__ float pll_wrap_phase_f32(float p)
{
    const float hi = 6.2831854820251465f, d = 1.7484555314695172e-07f;
    const float k = __builtin_truncf(p * 0.15915494309189535f);
    return __builtin_fmaf(k, d, __builtin_fmaf(k, -hi, p));
}

__host__ __device__ __forceinline__ float arctan2_ref(float y, float x)
{
    const float abs_y = (float)((double)__builtin_fabsf(y) + 1e-10);
    float r;
    double base;
    if (x >= 0) {
        r = (x - abs_y) / (x + abs_y);
        base = 0.78539816339744825;
    } else {
        r = (x + abs_y) / (abs_y - x);
        base = 2.35619449019234475;
    }
    const float angle = (float)(base - 0.78539816339744825 * (double)r);
    return (y < 0) ? -angle : angle;
}

__host__ __device__ __forceinline__ double arctan2_ref(double y, double x)
{
    const double abs_y = __builtin_fabs(y) + 1e-10;
    double r, base;
    if (x >= 0) {
        r = (x - abs_y) / (x + abs_y);
        base = 0.78539816339744825;
    } else {
        r = (x + abs_y) / (abs_y - x);
        base = 2.35619449019234475;
    }
    const double angle = base - 0.78539816339744825 * r;
    return (y < 0) ? -angle : angle;
}

// ---- Q_rsqrt of CarrierTrackingPLL.c:43-52 (always float, also in the double build)
__host__ __device__ __forceinline__ float q_rsqrt(float x)
{
    const float xhalf = 0.5f * x;
    int32_t i = (int32_t)bits_of(x);
    i = 0x5f3759df - (i >> 1);
    x = float_of((uint32_t)i);
    x = x * (1.5f - xhalf * x * x);
    x = x * (1.5f - xhalf * x * x);
    return x;
}

// ---- cabsf as glibc 2.35 computes it for finite inputs (AGC.c:57,65)
__host__ __device__ __forceinline__ float hypotf_glibc(float x, float y)
{
    return (float)__builtin_sqrt((double)x * (double)x + (double)y * (double)y);
}

// ---- double-precision sin / cos / sincos as glibc 2.35 computes them (reference call sites CarrierTrackingPLL.c:134-135,
// LowPassFilter.c:148,163).  glibc's dbl-64 routines are the IBM Accurate Mathematical Library's: |x| is split at a multiple
// of 1/128 (adding and subtracting 1.5 * 2^45), sin / cos of that multiple come from a double-double table, the remainder
// goes through short Taylor polynomials, and the pieces are combined with the angle-addition formulas; arguments beyond
// 2.426 are first reduced by pi/2 against a 4-piece constant (136 bits).  Restated from the published algorithm (the
// source is not in the reference tree); plain IEEE double operations in the order written there, no fused operation --
// bit-identical to the C library on every argument tried (tests/test_own_math.py: sincos, sin, cos over all ranges).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ const double kSinCosTab[440] = {
#include "pdt_sincostab.h"
};
#else
static const double kSinCosTab[440] = {
#include "pdt_sincostab.h"
};
#endif

namespace ibm {
constexpr double sn3 = -1.66666666666664880952546298448555E-01, sn5 = 8.33333214285722277379541354343671E-03,
                 cs2 = 4.99999999999999999999950396842453E-01, cs4 = -4.16666666666664434524222570944589E-02,
                 cs6 = 1.38888874007937613028114285595617E-03;
constexpr double s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ECEp-7, s3 = -0x1.A01A019DB08B8p-13, s4 = 0x1.71DE27B9A7ED9p-19,
                 s5 = -0x1.ADDFFC2FCDF59p-26;
constexpr double big = 0x1.8p45, hp0 = 0x1.921FB54442D18p0, hp1 = 0x1.1A62633145C07p-54;
constexpr double mp1 = 0x1.921FB58000000p0, mp2 = -0x1.DDE973C000000p-27, pp3 = -0x1.CB3B398000000p-55, pp4 = -0x1.d747f23e32ed7p-83;
constexpr double hpinv = 0x1.45F306DC9C883p-1, toint = 0x1.8p52;

// F: the build glibc's x86-64 dispatcher picks on an FMA-capable CPU.  Measured against the C library here (tests/
// test_own_math.py): sincos() is the plain build -- no fused operation -- while the stand-alone sin() and cos() are the FMA
// build, in which the compiler has contracted every a * b + c of the source into one fused operation (they differ from
// sincos() in ~0.07 % of arguments).  F = true reproduces that contraction.
template <bool F> __host__ __device__ __forceinline__ double ma(double a, double b, double c) { return F ? __builtin_fma(a, b, c) : a * b + c; }
template <bool F> __host__ __device__ __forceinline__ double ms(double a, double b, double c) { return F ? __builtin_fma(-a, b, c) : c - a * b; }   // c - a*b

template <bool F> __host__ __device__ __forceinline__ double do_sin(double x, double dx)
{
    const double xold = x;
    if (__builtin_fabs(x) < 0.126) {
        const double xx = x * x;
        const double poly = ma<F>(ma<F>(ma<F>(ma<F>(s5, xx, s4), xx, s3), xx, s2), xx, s1);
        const double d = F ? __builtin_fma(poly, x, -(0.5 * dx)) : poly * x - 0.5 * dx;
        const double t = ma<F>(d, xx, dx);
        return x + t;
    }
    if (x <= 0) dx = -dx;
    const double ux = big + __builtin_fabs(x);
    x = __builtin_fabs(x) - (ux - big);
    const int k = (int)(uint32_t)bits_of(ux) * 4;
    const double xx = x * x;
    const double s = x + ma<F>(x * xx, ma<F>(xx, sn5, sn3), dx);
    const double c2 = ma<F>(xx, ma<F>(xx, cs6, cs4), cs2);
    const double c = F ? __builtin_fma(x, dx, xx * c2) : x * dx + xx * c2;
    const double sn = kSinCosTab[k], ssn = kSinCosTab[k + 1], cs = kSinCosTab[k + 2], ccs = kSinCosTab[k + 3];
    const double cor = ma<F>(cs, s, ms<F>(sn, c, ma<F>(s, ccs, ssn)));
    return __builtin_copysign(sn + cor, xold);
}
template <bool F> __host__ __device__ __forceinline__ double do_cos(double x, double dx)
{
    if (x < 0) dx = -dx;
    const double ux = big + __builtin_fabs(x);
    x = __builtin_fabs(x) - (ux - big) + dx;
    const int k = (int)(uint32_t)bits_of(ux) * 4;
    const double xx = x * x;
    const double s = ma<F>(x * xx, ma<F>(xx, sn5, sn3), x);
    const double c = xx * ma<F>(xx, ma<F>(xx, cs6, cs4), cs2);
    const double sn = kSinCosTab[k], ssn = kSinCosTab[k + 1], cs = kSinCosTab[k + 2], ccs = kSinCosTab[k + 3];
    const double cor = ms<F>(sn, s, ms<F>(cs, c, ms<F>(s, ssn, ccs)));
    return cs + cor;
}
// |x| < 105414350: quadrant, and the reduced argument as a + da
template <bool F> __host__ __device__ __forceinline__ int reduce(double x, double &a, double &da)
{
    const double t = ma<F>(x, hpinv, toint);
    const double xn = t - toint;
    const double y = ms<F>(xn, mp2, ms<F>(xn, mp1, x));
    const int n = (int)(bits_of(t) & 3u);
    const double t2 = ms<F>(xn, pp3, y);
    double db = ms<F>(xn, pp3, y - t2);
    const double b = ms<F>(xn, pp4, t2);
    db += ms<F>(xn, pp4, t2 - b);
    a = b;
    da = db;
    return n;
}
template <bool F> __host__ __device__ __forceinline__ double do_sincos(double a, double da, int n)
{
    const double r = (n & 1) ? do_cos<F>(a, da) : do_sin<F>(a, da);
    return (n & 2) ? -r : r;
}
}  // namespace ibm

// sincos(x): valid for |x| < 105414350 (the PLL phase lives in (-2 pi, 2 pi])
__host__ __device__ __forceinline__ void sincos_glibc(double x, double &sv, double &cv)
{
    const uint32_t k = (uint32_t)(bits_of(x) >> 32) & 0x7fffffffu;
    if (k < 0x400368fdu) {
        if (k < 0x3e400000u) { sv = x; cv = 1.0; return; }
        if (k < 0x3feb6000u) { sv = ibm::do_sin<false>(x, 0); cv = ibm::do_cos<false>(x, 0); return; }
        const double y = ibm::hp0 - __builtin_fabs(x), a = y + ibm::hp1, da = (y - a) + ibm::hp1;
        sv = __builtin_copysign(ibm::do_cos<false>(a, da), x);
        cv = ibm::do_sin<false>(a, da);
        return;
    }
    double a, da;
    const int n = ibm::reduce<false>(x, a, da);
    sv = ibm::do_sincos<false>(a, da, n);
    cv = ibm::do_sincos<false>(a, da, n + 1);
}
// sin(x) / cos(x) as separate calls (the tap generator): the same pieces, dispatched as s_sin.c does, FMA build
__host__ __device__ __forceinline__ double sin_glibc(double x)
{
    const uint32_t k = (uint32_t)(bits_of(x) >> 32) & 0x7fffffffu;
    if (k < 0x3e500000u) return x;
    if (k < 0x3feb6000u) return ibm::do_sin<true>(x, 0);
    if (k < 0x400368fdu) return __builtin_copysign(ibm::do_cos<true>(ibm::hp0 - __builtin_fabs(x), ibm::hp1), x);
    double a, da;
    const int n = ibm::reduce<true>(x, a, da);
    return ibm::do_sincos<true>(a, da, n);
}
__host__ __device__ __forceinline__ double cos_glibc(double x)
{
    const uint32_t k = (uint32_t)(bits_of(x) >> 32) & 0x7fffffffu;
    if (k < 0x3e400000u) return 1.0;
    if (k < 0x3feb6000u) return ibm::do_cos<true>(x, 0);
    if (k < 0x400368fdu) {
        const double y = ibm::hp0 - __builtin_fabs(x), a = y + ibm::hp1, da = (y - a) + ibm::hp1;
        return ibm::do_sin<true>(a, da);
    }
    double a, da;
    const int n = ibm::reduce<true>(x, a, da);
    return ibm::do_sincos<true>(a, da, n + 1);
}

// ---- cabs (double) as glibc 2.35 computes it for the magnitudes that occur (|x|, |y| <= 1): e_hypot.c, the variant
// without a fused multiply-add (AGC.c:57,65 in the double build)
__host__ __device__ __forceinline__ double hypot_glibc(double x, double y)
{
    double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    if (ax < ay) { const double t = ax; ax = ay; ay = t; }
    if (ay <= ax * 0x1p-54) return ax + ay;
    double h = __builtin_sqrt(ax * ax + ay * ay), t1, t2;
    if (h <= 2.0 * ay) {
        const double delta = h - ay;
        t1 = ax * (2.0 * delta - ax);
        t2 = (delta - 2.0 * (ax - ay)) * delta;
    } else {
        const double delta = h - ax;
        t1 = 2.0 * delta * (ax - 2.0 * ay);
        t2 = (4.0 * delta - ay) * ay + delta * delta;
    }
    h -= (t1 + t2) / (2.0 * h);
    return h;
}

template <typename T> struct Real;
template <> struct Real<double> {
    static __host__ __device__ __forceinline__ double abs(double v) { return __builtin_fabs(v); }
    static __host__ __device__ __forceinline__ double max(double a, double b) { return __builtin_fmax(a, b); }
    static __host__ __device__ __forceinline__ double rint(double v) { return __builtin_rint(v); }
    static __host__ __device__ __forceinline__ void sincos(double p, double &s, double &c) { sincos_glibc(p, s, c); }
    static __host__ __device__ __forceinline__ double hypot(double x, double y) { return hypot_glibc(x, y); }
};
template <> struct Real<float> {
    static __host__ __device__ __forceinline__ float abs(float v) { return __builtin_fabsf(v); }
    static __host__ __device__ __forceinline__ float max(float a, float b) { return __builtin_fmaxf(a, b); }
    static __host__ __device__ __forceinline__ float rint(float v) { return __builtin_rintf(v); }
    static __host__ __device__ __forceinline__ void sincos(float p, float &s, float &c) { sincosf_glibc(p, s, c); }
    static __host__ __device__ __forceinline__ float hypot(float x, float y) { return hypotf_glibc(x, y); }
};

}  // namespace pdt
