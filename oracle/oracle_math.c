/*
 * oracle/oracle_math.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Restatements of the third-party arithmetic the reference pulls in from
 * glibc 2.35-0ubuntu3.11 libm (not vendored under /root/reference) and of the
 * two small helpers in common/CarrierTrackingPLL.c, written so that the
 * results do not depend on the libm of the machine the oracle runs on.
 *
 *  orc_sincosf : glibc 2.35 sysdeps/ieee754/flt-32/s_sincosf.c as built for
 *                x86-64 with FMA (the ifunc variant selected on every FMA+AVX2
 *                CPU: sysdeps/x86_64/fpu/multiarch/s_sincosf-fma.c).  The
 *                published algorithm: y -> double, quadrant n = round(y*2/pi)
 *                through the 2^24-scaled "hpi_inv" constant, r = y - n*pi/2,
 *                then degree-7/8 minimax polynomials in r^2 evaluated in
 *                double; every a + b*c in the evaluation is one fused
 *                multiply-add in that build, which is what is written below
 *                with explicit fma() calls.  tests/test_oracle_math.py checks
 *                bit-equality with the container's sincosf over 2^24-spaced
 *                sweeps and random arguments in (-2pi, 2pi].
 *                Used at: common/CarrierTrackingPLL.c:106-107 (gcc -O2 merges
 *                the sinf/cosf pair into one sincosf call).
 *  orc_hypotf  : glibc 2.35 sysdeps/ieee754/flt-32/e_hypotf.c:
 *                (float) sqrt((double)x*x + (double)y*y) for finite inputs.
 *                Used at: common/AGC.c:57,65 (cabsf).
 *  orc_q_rsqrt : common/CarrierTrackingPLL.c:43-52.
 *  orc_arctan2 : common/CarrierTrackingPLL.c:15-40.
 */
#define _GNU_SOURCE
#include <math.h>
#include <string.h>
#include "oracle.h"

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* polynomial tables: [0] = +cos form, [1] = -cos form (negated c's) */
typedef struct { double c0, c1, c2, c3, c4, s1, s2, s3; } sc_poly;
static const sc_poly SC[2] = {
    { 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16,
      -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13 },
    { -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16,
      -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13 },
};
static const double SC_SIGN[4] = { 1.0, -1.0, -1.0, 1.0 };
#define SC_HPI_INV 0x1.45f306dc9c883p+23 /* 2/pi * 2^24 */
#define SC_HPI     0x1.921fb54442d18p+0  /* pi/2 */

static inline void sc_eval(double x, double x2, const sc_poly *p, int n, float *sinp, float *cosp)
{
    double x3 = x * x2;
    double x4 = x2 * x2;
    double s1v = fma(x2, p->s3, p->s2);
    double c2v = fma(x2, p->c4, p->c3);
    double c1v = fma(x2, p->c1, p->c0);
    double x5 = x2 * x3;
    double x6 = x2 * x4;
    double s = fma(x3, p->s1, x);
    double c = fma(x4, p->c2, c1v);
    float sv = (float)fma(s1v, x5, s);
    float cv = (float)fma(c2v, x6, c);
    if (n & 1) { *sinp = cv; *cosp = sv; }
    else       { *sinp = sv; *cosp = cv; }
}

void orc_sincosf(float y, float *sinp, float *cosp)
{
    double x = (double)y;
    uint32_t top = (f2u(y) >> 20) & 0x7ff;
    if (top < 0x3f4) {                      /* |y| < pi/4 (by exponent/top mantissa bits) */
        if (top < 0x398) {                  /* |y| < 2^-12 */
            *sinp = y;
            *cosp = 1.0f;
            return;
        }
        sc_eval(x, x * x, &SC[0], 0, sinp, cosp);
    } else if (top < 0x42f) {               /* |y| < 120 */
        double r = x * SC_HPI_INV;
        int n = ((int32_t)r + 0x800000) >> 24;
        double xr = fma(-(double)n, SC_HPI, x);
        double sg = SC_SIGN[n & 3];
        sc_eval(xr * sg, xr * xr, &SC[(n >> 1) & 1], n, sinp, cosp);
    } else {
        /* outside the PLL's phase range (-2pi, 2pi]; defer to libm */
        sincosf(y, sinp, cosp);
    }
}

float orc_hypotf(float x, float y)
{
    if (!isfinite(x) || !isfinite(y))
        return hypotf(x, y);
    return (float)sqrt((double)x * (double)x + (double)y * (double)y);
}

/* ---- double precision (ARGOS) -----------------------------------------------------------
 * Two modes, selected with orc_set_math_mode():
 *   0 (default)  glibc's own sincos()/hypot(): the restatement is then bit-identical to the
 *                reference objects at every stage (tests/test_oracle_ref.py), which pins the
 *                restated logic;
 *   1 "portable" the evaluation the HIP kernels use for the ARGOS chain: a plain-double
 *                sine/cosine (Cody-Waite reduction by pi/2 in two steps and the classic
 *                fdlibm/musl minimax kernels; every operation a single IEEE double operation,
 *                no libm), and hypot = sqrt(x*x + y*y).  glibc's routines are not
 *                restated for the device (table-driven, 440-entry table); the two modes differ
 *                in the last bit of a small fraction of results, which the contracting stages
 *                absorb: tests check that the portable mode reproduces the reference's bits,
 *                symbol and packet output on the ARGOS fixtures, and the GPU is compared
 *                bit-for-bit with the portable mode.
 */
static int g_math_mode = 0;
void orc_set_math_mode(int mode) { g_math_mode = mode; }
int orc_get_math_mode(void) { return g_math_mode; }

static const double
    PIO2_1 = 1.57079632673412561417e+00,  /* first 33 bits of pi/2 */
    PIO2_1T = 6.07710050650619224932e-11, /* pi/2 - PIO2_1 */
    PIO2_2 = 6.07710050630396597660e-11,  /* second 33 bits of pi/2 */
    PIO2_2T = 2.02226624879595063154e-21, /* pi/2 - (PIO2_1 + PIO2_2) */
    INVPIO2 = 6.36619772367581382433e-01,
    KS1 = -1.66666666666666324348e-01, KS2 = 8.33333333332248946124e-03, KS3 = -1.98412698298579493134e-04,
    KS4 = 2.75573137070700676789e-06, KS5 = -2.50507602534068634195e-08, KS6 = 1.58969099521155010221e-10,
    KC1 = 4.16666666666666019037e-02, KC2 = -1.38888888888741095749e-03, KC3 = 2.48015872894767294178e-05,
    KC4 = -2.75573143513906633035e-07, KC5 = 2.08757232129817482790e-09, KC6 = -1.13596475577881948265e-11;

static inline double ksin(double x, double y)
{
    const double z = x * x, w = z * z;
    const double r = KS2 + z * (KS3 + z * KS4) + z * w * (KS5 + z * KS6);
    const double v = z * x;
    return x - ((z * (0.5 * y - v * r) - y) - v * KS1);
}
static inline double kcos(double x, double y)
{
    const double z = x * x, w0 = z * z;
    const double r = z * (KC1 + z * (KC2 + z * KC3)) + (w0 * w0) * (KC4 + z * (KC5 + z * KC6));
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}

void orc_sincos_portable(double x, double *sp, double *cp)
{
    /* valid for |x| < ~1e5; the PLL phase lives in (-2pi, 2pi] */
    const double fn = rint(x * INVPIO2);
    const int n = (int)fn;
    double t = x - fn * PIO2_1;
    double w = fn * PIO2_1T;
    /* second Cody-Waite step, always taken (good to ~118 bits) */
    const double t2 = t;
    w = fn * PIO2_2;
    t = t2 - w;
    w = fn * PIO2_2T - ((t2 - t) - w);
    (void)PIO2_1T;
    const double y0 = t - w;
    const double y1 = (t - y0) - w;
    const double s = ksin(y0, y1), c = kcos(y0, y1);
    switch (n & 3) {
    case 0: *sp = s; *cp = c; break;
    case 1: *sp = c; *cp = -s; break;
    case 2: *sp = -s; *cp = -c; break;
    default: *sp = -c; *cp = s; break;
    }
}

double orc_hypot(double x, double y) { return g_math_mode ? sqrt(x * x + y * y) : hypot(x, y); }
void orc_sincos(double x, double *s, double *c)
{
    if (g_math_mode) orc_sincos_portable(x, s, c);
    else sincos(x, s, c);
}

float orc_q_rsqrt(float x)
{
    float xhalf = 0.5f * x;
    int32_t i = (int32_t)f2u(x);
    i = 0x5f3759df - (i >> 1);
    x = u2f((uint32_t)i);
    x = x * (1.5f - xhalf * x * x);
    x = x * (1.5f - xhalf * x * x);
    return x;
}

#define ATAN_C1 (0.78539816339744825)
#define ATAN_C2 (2.35619449019234475)

float orc_arctan2_f32(float y, float x)
{
    float r, angle;
    float abs_y = (float)((double)fabsf(y) + 1e-10);
    if (x >= 0) {
        r = (x - abs_y) / (x + abs_y);
        angle = (float)(ATAN_C1 - ATAN_C1 * (double)r);
    } else {
        r = (x + abs_y) / (abs_y - x);
        angle = (float)(ATAN_C2 - ATAN_C1 * (double)r);
    }
    return (y < 0) ? -angle : angle;
}

double orc_arctan2_f64(double y, double x)
{
    double r, angle;
    double abs_y = fabs(y) + 1e-10;
    if (x >= 0) {
        r = (x - abs_y) / (x + abs_y);
        angle = ATAN_C1 - ATAN_C1 * r;
    } else {
        r = (x + abs_y) / (abs_y - x);
        angle = ATAN_C2 - ATAN_C1 * r;
    }
    return (y < 0) ? -angle : angle;
}
