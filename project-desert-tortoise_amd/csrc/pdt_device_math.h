// pdt_device_math.h -- gfx950 device-side scalar math for the PLL / AGC stages.
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

// ---- sincosf: glibc 2.35 algorithm (reference call site CarrierTrackingPLL.c:106-107)
struct SinCosPoly { double c0, c1, c2, c3, c4, s1, s2, s3; };

__device__ __forceinline__ void sincosf_eval(double x, double x2, bool neg, int n, float &sinv, float &cosv)
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
__device__ __forceinline__ void sincosf_glibc(float y, float &sinv, float &cosv)
{
    const double x = (double)y;
    const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;
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

// ---- arctan2 approximation of CarrierTrackingPLL.c:15-40
__device__ __forceinline__ float arctan2_ref(float y, float x)
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

__device__ __forceinline__ double arctan2_ref(double y, double x)
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
__device__ __forceinline__ float q_rsqrt(float x)
{
    const float xhalf = 0.5f * x;
    int32_t i = __float_as_int(x);
    i = 0x5f3759df - (i >> 1);
    x = __int_as_float(i);
    x = x * (1.5f - xhalf * x * x);
    x = x * (1.5f - xhalf * x * x);
    return x;
}

// ---- cabsf as glibc 2.35 computes it for finite inputs (AGC.c:57,65)
__device__ __forceinline__ float hypotf_glibc(float x, float y)
{
    return (float)__builtin_sqrt((double)x * (double)x + (double)y * (double)y);
}

// ---- double-precision sine/cosine for the ARGOS chain: plain IEEE double operations only
// (Cody-Waite reduction by pi/2 in two steps, fdlibm/musl minimax kernels).  The reference calls
// glibc's table-driven sincos() here (CarrierTrackingPLL.c:134-135); this evaluation differs from
// it in the last bit of ~3% of arguments, which the contracting stages absorb -- the oracle has a
// matching "portable" mode that is checked against the reference's bit/packet output.
__device__ __forceinline__ double ksin_d(double x, double y)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x, w = z * z;
    const double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
    const double v = z * x;
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
__device__ __forceinline__ double kcos_d(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x, w0 = z * z;
    const double r = z * (C1 + z * (C2 + z * C3)) + (w0 * w0) * (C4 + z * (C5 + z * C6));
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}
__device__ __forceinline__ void sincos_portable(double x, double &sv, double &cv)
{
    const double fn = __builtin_rint(x * 6.36619772367581382433e-01);
    const int n = (int)fn;
    const double t2 = x - fn * 1.57079632673412561417e+00;
    double w = fn * 6.07710050630396597660e-11;
    const double t = t2 - w;
    w = fn * 2.02226624879595063154e-21 - ((t2 - t) - w);
    const double y0 = t - w;
    const double y1 = (t - y0) - w;
    const double s = ksin_d(y0, y1), c = kcos_d(y0, y1);
    const int q = n & 3;
    sv = (q == 0) ? s : ((q == 1) ? c : ((q == 2) ? -s : -c));
    cv = (q == 0) ? c : ((q == 1) ? -s : ((q == 2) ? -c : s));
}

template <typename T> struct Real;
template <> struct Real<double> {
    static __device__ __forceinline__ double abs(double v) { return __builtin_fabs(v); }
    static __device__ __forceinline__ double max(double a, double b) { return __builtin_fmax(a, b); }
    static __device__ __forceinline__ double rint(double v) { return __builtin_rint(v); }
    static __device__ __forceinline__ void sincos(double p, double &s, double &c) { sincos_portable(p, s, c); }
    static __device__ __forceinline__ double hypot(double x, double y) { return __builtin_sqrt(x * x + y * y); }
};
template <> struct Real<float> {
    static __device__ __forceinline__ float abs(float v) { return __builtin_fabsf(v); }
    static __device__ __forceinline__ float max(float a, float b) { return __builtin_fmaxf(a, b); }
    static __device__ __forceinline__ float rint(float v) { return __builtin_rintf(v); }
    static __device__ __forceinline__ void sincos(float p, float &s, float &c) { sincosf_glibc(p, s, c); }
    static __device__ __forceinline__ float hypot(float x, float y) { return hypotf_glibc(x, y); }
};

}  // namespace pdt
