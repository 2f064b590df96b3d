// pdt_kernels_front.h -- front half of the chain on gfx950: PCM decode + carrier PLL,
// low-pass / interpolating FIR, StaticGain, AGC (+Squelch).
//
// Parallelisation scheme for the serial float recurrences (PLL, AGC): the stream
// is cut into blocks; lane l of a wavefront owns block l and walks it
// sequentially, after first replaying `warm` samples before its block from a
// guessed state.  Both recurrences contract: a wrong state becomes *bit-identical*
// to the true trajectory after a bounded number of samples, so
// after the warm-up the lane is on the true trajectory.  This is not assumed:
// each lane records its state at its official block start, and a fix-up kernel
// compares it bitwise with the end state of the previous block; on mismatch the
// block is re-run from the true state.  By induction over blocks the output is
// exactly the sequential result.
#pragma once
#include <type_traits>
#include <utility>
#include "pdt_device_math.h"

namespace pdt {

// ------------------------------------------------------------------------------------------
// Carrier tracking PLL (reference: common/CarrierTrackingPLL.c:54-278)
// ------------------------------------------------------------------------------------------
template <typename T> struct PllParams {
    T Fs;
    T lock_thr;      // d_lock_threshold
    T lock_alpha;    // lockSigAlpha
    T alpha_acq, beta_acq, alpha_trk, beta_trk;
    T alpha_wide, beta_wide;   // warm-up only: 8x the acquisition bandwidth (pulls in a kHz-off frequency guess)
    T max_freq, min_freq;
    T sweep0, avg0, phase0;
    T freq0, locksig0;   // the rest of the acquisition's starting state, and the sample it starts at: the reference's first-call
    long long i0;        // values (0, 0, sample 0) for a capture, the carried state when a stream continues before the lock
    T cond_lo, cond_hi;   // |pi/2 - averagePhase| < 0.05 (evaluated the reference's way) <=> cond_lo <= averagePhase <= cond_hi
    int want_lock;   // 1 = lockSignalStreamOut != NULL (ARGOS)
};

template <typename T> struct PllState {
    T phase, freq, avg_phase, locksig, sweep;
};

// what the acquisition kernel leaves behind for the tracking kernels and the host
template <typename T> struct PllLockInfo {
    long long lock_sample;   // global index of the sample at which lock was declared, -1 = none
    PllState<T> st;          // state after that sample
    T freq_at_lock;          // d_freq when "PLL locked at" is printed
    T avg_at_lock;
};

// Sample source: the WAV payload as it is (interleaved int16 I,Q -> value/32768, wave.c:127-172) or
// RAW interleaved float32 I,Q taken as they are (wave.c:413-540).  The format is uniform per launch.
struct IqSrc {
    const void *p;
    int fmt;          // 0 = PCM16 pairs, 1 = float32 pairs
};

template <typename T> struct IqSample {
    static __device__ __forceinline__ void get(IqSrc s, long long i, T &a, T &b)
    {
        if (s.fmt == 0) {
            const int v = reinterpret_cast<const int *>(s.p)[i];     // I | Q<<16, little endian
            a = (T)(short)(v & 0xffff) / (T)32768;
            b = (T)(short)(v >> 16) / (T)32768;
        } else {
            const float2 v = reinterpret_cast<const float2 *>(s.p)[i];
            a = (T)v.x;
            b = (T)v.y;
        }
    }
};

// The same in two halves, so that a load can be issued long before its conversion is wanted (the acquisition's pipeline): the raw
// words of sample i, and their conversion -- exactly IqSample<T>::get's.
__device__ __forceinline__ uint2 iq_raw(IqSrc s, long long i)
{
    uint2 r;
    if (s.fmt == 0) {
        r.x = (unsigned)reinterpret_cast<const int *>(s.p)[i];
        r.y = 0u;
    } else {
        const float2 v = reinterpret_cast<const float2 *>(s.p)[i];
        r.x = __float_as_uint(v.x);
        r.y = __float_as_uint(v.y);
    }
    return r;
}
template <typename T> __device__ __forceinline__ void iq_conv(int fmt, uint2 r, T &a, T &b)
{
    if (fmt == 0) {
        const int v = (int)r.x;
        a = (T)(short)(v & 0xffff) / (T)32768;
        b = (T)(short)(v >> 16) / (T)32768;
    } else {
        a = (T)__uint_as_float(r.x);
        b = (T)__uint_as_float(r.y);
    }
}

// NS consecutive IQ samples with 16-byte loads (4 PCM16 pairs / 2 float pairs per load; the address need only be
// 4- resp. 8-byte aligned).  Same conversion as IqSample<T>::get.  A lane-per-block access touches one cache line per lane
// whatever its width, so the wide form is a quarter (half) of the load instructions; the elementwise kernels use it to
// take four samples per thread.
template <typename T, int FMT, int NS>
__device__ __forceinline__ void iq_block(const void *p, long long i, T (&a)[NS], T (&b)[NS])
{
    if (FMT == 0) {
        struct __attribute__((packed, aligned(4))) Q { int v[4]; };
#pragma unroll
        for (int q = 0; q < NS / 4; q++) {
            const Q w = *reinterpret_cast<const Q *>(reinterpret_cast<const int *>(p) + i + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                a[4 * q + e] = (T)(short)(w.v[e] & 0xffff) / (T)32768;
                b[4 * q + e] = (T)(short)(w.v[e] >> 16) / (T)32768;
            }
        }
    } else {
        struct __attribute__((packed, aligned(8))) Q { float v[4]; };
#pragma unroll
        for (int q = 0; q < NS / 2; q++) {
            const Q w = *reinterpret_cast<const Q *>(reinterpret_cast<const float2 *>(p) + i + 2 * q);
            a[2 * q] = (T)w.v[0]; b[2 * q] = (T)w.v[1];
            a[2 * q + 1] = (T)w.v[2]; b[2 * q + 1] = (T)w.v[3];
        }
    }
}

// shared part of one PLL iteration: mix, error, loop update, wrap, clamp (:106-188)
template <typename T>
__device__ __forceinline__ void pll_core(T a, T b, T &phase, T &freq, T alpha, T beta, T maxf, T minf, T &o_re, T &o_im,
                                         T &t_real, T &t_imag)
{
    Real<T>::sincos(phase, t_imag, t_real);
    const T c = t_real, d = -t_imag;
    o_re = a * c - b * d;
    o_im = a * d + b * c;
    const T sample_phase = arctan2_ref(b, a);
    const T diff = sample_phase - phase;
    T err;
    if ((double)diff > PDT_PI)
        err = (T)((double)diff - 2 * PDT_PI);
    else if ((double)diff < -PDT_PI)
        err = (T)((double)diff + 2 * PDT_PI);
    else
        err = diff;
    freq = freq + beta * err;
    phase = phase + freq + alpha * err;
    while ((double)phase > 2 * PDT_PI) phase = (T)((double)phase - 2.0 * PDT_PI);
    while ((double)phase < -2 * PDT_PI) phase = (T)((double)phase + 2.0 * PDT_PI);
    if (freq > maxf)
        freq = maxf;
    else if (freq < minf)
        freq = minf;
}

// lock-detector EMA (:194-220)
template <typename T> __device__ __forceinline__ T pll_locksig(T a, T b, T t_real, T t_imag, T locksig, T lock_alpha)
{
    const T mag2 = a * a + b * b;
    const T inv = (T)q_rsqrt((float)mag2);
    const T re = a * inv, im = b * inv;
    return (T)((double)locksig * (1.0 - (double)lock_alpha) + (double)(lock_alpha * (re * t_real + im * t_imag)));
}

// ---- tracking phase, split in three so that only the true recurrence is serial -------------
//   k_pll_theta : theta_i = arctan2(Im x_i, Re x_i)                (elementwise, state-free)
//   k_pll_phase : (phase, freq) recurrence over theta, one lane per block, stores the phase
//                 *used for* sample i (the value before the update)  (serial, ~12 ops/sample)
//   k_pll_mix   : realDataOut_i = Im(x_i e^{-j phase_i}) via the glibc sincosf evaluation
//                 (+ the lock-detector input term for ARGOS)          (elementwise)
// The float operations and their order are exactly those of one loop iteration of the
// reference; only the order in which *independent* iterations' pieces run is changed.

// Lane-tiled ("LT") layout of the theta and phase streams.  The (phase, freq) recurrence is walked one lane per block of B
// samples, 64 consecutive blocks per wavefront, all lanes at the same offset inside their blocks.  In natural order a
// wavefront's 16-byte loads then touch 64 different cache lines -- 64 clocks of the CU's texture-address path per load, which
// is what bounds the walkers as soon as a few of them share a CU (a batch of captures, an hour-long capture).  So the
// streams the walkers touch are kept transposed: tile w = blocks [64 w, 64 w + 64); its vector row q holds, lane after lane,
// the q-th 16-byte vector of each of the 64 blocks.  A wavefront in lock step reads / writes 1 KiB of consecutive bytes per
// instruction.  The elementwise kernels on either side (theta, mix) transpose through LDS, coalesced on both faces.
template <typename T> struct Lt {
    static constexpr int VN = 16 / sizeof(T);       // elements per 16-byte vector
    static constexpr int ROW = 64 * VN;             // elements per row
    static constexpr int RG = 16;                   // rows a transposing workgroup handles (B is a multiple of RG * VN)
    // element index of natural sample i
    static __host__ __device__ __forceinline__ long long index(long long i, long long B)
    {
        const long long j = i / B, p = i - j * B;
        return (j >> 6) * (64 * B) + (p / VN) * ROW + (j & 63) * VN + (p % VN);
    }
};

template <typename T> __device__ __forceinline__ void iq_vec(IqSrc pcm, long long i, T (&a)[Lt<T>::VN], T (&b)[Lt<T>::VN])
{
    constexpr int VN = Lt<T>::VN;
    if constexpr (VN == 4) {
        if (pcm.fmt == 0) iq_block<T, 0, 4>(pcm.p, i, a, b);
        else iq_block<T, 1, 4>(pcm.p, i, a, b);
    } else {
#pragma unroll
        for (int e = 0; e < VN; e++) IqSample<T>::get(pcm, i + e, a[e], b[e]);
    }
}

// theta_i = arctan2(Im x_i, Re x_i) (:128), written in the LT layout.  One workgroup = RG rows of one tile: the I/Q of
// 64 lane-rows x RG vectors is read in runs of RG * 16 sample bytes, transposed through LDS, and leaves as RG KiB of
// consecutive bytes.
template <typename T>
__device__ __forceinline__ void k_pll_theta(IqSrc pcm, long long n, long long B, T *__restrict__ theta_lt)
{
    constexpr int VN = Lt<T>::VN, ROW = Lt<T>::ROW, RG = Lt<T>::RG;
    __shared__ Vec16<T> s_v[64 * (RG + 1)];
    const long long groups = B / (VN * RG);
    const long long w = (long long)blockIdx.x / groups, g = (long long)blockIdx.x - w * groups;
    if ((w << 6) * B >= n) return;
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
        const int l = pass * 16 + ((int)threadIdx.x >> 4), qd = (int)threadIdx.x & 15;
        const long long i = ((w << 6) + l) * B + (g * RG + qd) * VN;
        Vec16<T> o;
        if (i + VN <= n) {
            T a[VN], b[VN];
            iq_vec<T>(pcm, i, a, b);
#pragma unroll
            for (int e = 0; e < VN; e++) o.v[e] = arctan2_ref(b[e], a[e]);
        } else {
#pragma unroll
            for (int e = 0; e < VN; e++) {
                o.v[e] = 0;
                if (i + e < n) {
                    T a, b;
                    IqSample<T>::get(pcm, i + e, a, b);
                    o.v[e] = arctan2_ref(b, a);
                }
            }
        }
        s_v[l * (RG + 1) + qd] = o;
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
        const int qd = pass * 4 + ((int)threadIdx.x >> 6), l = (int)threadIdx.x & 63;
        *reinterpret_cast<Vec16<T> *>(theta_lt + (w << 6) * B + (g * RG + qd) * ROW + l * VN) = s_v[l * (RG + 1) + qd];
    }
}

// Comparisons of a DT value with the double constants pi / 2pi, as the reference writes them
// ("(sample_phase-d_phase) > M_PI", "d_phase > 2*M_PI": the float operand is promoted).  For
// float the promoted comparison is equivalent to a float comparison with the neighbouring
// float: (float)pi and (float)(2pi) both lie ABOVE the double constants, so
//   (double)x >  pi_d   <=>  x >= (float)pi,      (double)x < -pi_d   <=>  x <= -(float)pi.
template <typename T> struct PiCmp;
template <> struct PiCmp<float> {
    static __device__ __forceinline__ bool gt_pi(float x) { return x >= 3.14159274101257324f; }
    static __device__ __forceinline__ bool lt_mpi(float x) { return x <= -3.14159274101257324f; }
    static __device__ __forceinline__ bool gt_2pi(float x) { return x >= 6.28318548202514648f; }
    static __device__ __forceinline__ bool lt_m2pi(float x) { return x <= -6.28318548202514648f; }
};
template <> struct PiCmp<double> {
    static __device__ __forceinline__ bool gt_pi(double x) { return x > PDT_PI; }
    static __device__ __forceinline__ bool lt_mpi(double x) { return x < -PDT_PI; }
    static __device__ __forceinline__ bool gt_2pi(double x) { return x > 2 * PDT_PI; }
    static __device__ __forceinline__ bool lt_m2pi(double x) { return x < -2 * PDT_PI; }
};

// |x| against pi / 2pi with the reference's promoted comparison semantics (see PiCmp)
template <typename T> struct PiAbs;
template <> struct PiAbs<float> {
    static __device__ __forceinline__ bool ge_pi(float x) { return __builtin_fabsf(x) >= 3.14159274101257324f; }
    static __device__ __forceinline__ bool ge_2pi(float x) { return __builtin_fabsf(x) >= 6.28318548202514648f; }
    static __device__ __forceinline__ float clamp(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
};
template <> struct PiAbs<double> {
    static __device__ __forceinline__ bool ge_pi(double x) { return __builtin_fabs(x) > PDT_PI; }
    static __device__ __forceinline__ bool ge_2pi(double x) { return __builtin_fabs(x) > 2 * PDT_PI; }
    static __device__ __forceinline__ double clamp(double v, double lo, double hi) { return (v > hi) ? hi : ((v < lo) ? lo : v); }
};

// x -/+ 2pi evaluated in double and narrowed, exactly as the reference writes it
// ("error = (..) - 2*M_PI" / "+ 2*M_PI", "d_phase - 2.0*M_PI" / "+ 2.0*M_PI"): the sign of the
// correction is the opposite of the sign of x, so one f64 add serves both branches.
template <typename T> __device__ __forceinline__ T unwrap_2pi(T x)
{
    const double off = __builtin_copysign(2.0 * PDT_PI, -(double)x);
    T r = (T)((double)x + off);
    asm volatile("" : "+v"(r));          // keep it a value (select below), not a re-branched computation
    return r;
}
// float: (float)((double)x -+ 2pi) without leaving f32.  With hi = (float)(2pi) and
// d = (float)(hi - 2pi) [hi lies above 2pi], x - hi is exact for pi <= |x| <= 4pi (Sterbenz) and
// (x - hi) + d rounds to the same float as the double expression for EVERY float in that range
// (all 17.2 M of them compared in tests/test_oracle_math.py::test_unwrap_f32_exhaustive); the
// result is only used when |x| >= pi, and |x| < 13 always holds (|theta| <= pi, |phase| < 2pi + 2).
template <> __device__ __forceinline__ float unwrap_2pi<float>(float x)
{
    const float hi = __builtin_copysignf(6.2831854820251465f, x);
    const float d = __builtin_copysignf(1.7484555314695172e-07f, x);
    float r = (x - hi) + d;
    asm volatile("" : "+v"(r));
    return r;
}

// one step of the loop filter given theta (:165-188), branch-free.
// SLOW_WRAP keeps the reference's "while" wrap loops for loop gains so large that a single
// +-2pi correction might not suffice (|freq| + alpha*pi + beta*pi >= 2pi); the host selects it.
template <typename T, bool SLOW_WRAP = false>
__device__ __forceinline__ void pll_phase_step(T th, T &phase, T &freq, T alpha, T beta, T maxf, T minf)
{
    if constexpr (std::is_same<T, float>::value && !SLOW_WRAP) {
        // float: 16 vector operations per step instead of 19 (the walkers are bound by instruction issue -- one wavefront per
        // SIMD, 4 clocks per operation -- so the count is the time): both wraps as fused multiply-adds with exact products,
        // pdt_device_math.h
        const float err = pll_wrap_error_f32(th - phase);
        const float f1 = freq + beta * err;
        phase = pll_wrap_phase_f32(phase + f1 + alpha * err);
        freq = PiAbs<float>::clamp(f1, minf, maxf);
        return;
    }
    const T diff = th - phase;
    const T wrapped = unwrap_2pi(diff);
    const T err = PiAbs<T>::ge_pi(diff) ? wrapped : diff;
    const T f1 = freq + beta * err;
    T ph = phase + f1 + alpha * err;
    const T phw = unwrap_2pi(ph);
    ph = PiAbs<T>::ge_2pi(ph) ? phw : ph;
    if (SLOW_WRAP) {
        while (PiCmp<T>::gt_2pi(ph)) ph = (T)((double)ph - 2.0 * PDT_PI);
        while (PiCmp<T>::lt_m2pi(ph)) ph = (T)((double)ph + 2.0 * PDT_PI);
    }
    phase = ph;
    freq = PiAbs<T>::clamp(f1, minf, maxf);
}

// theta of one sample, as k_pll_theta computes it (the acquisition reads a few thousand samples in natural order; the theta
// stream itself is kept in the block-parallel kernel's LT layout)
template <typename T> __device__ __forceinline__ T theta_of(IqSrc pcm, long long i)
{
    T a, b;
    IqSample<T>::get(pcm, i, a, b);
    return arctan2_ref(b, a);
}
#define PDT_ACQ_NB 32
__device__ __forceinline__ float lane_get(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }
__device__ __forceinline__ double lane_get(double v, int k)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, k);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), k);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <typename T> __device__ __forceinline__ void pll_sweep_step(T &fr, T &sw, T maxf, T minf)
{
    fr = fr + sw;                                    // :239-252
    if (fr >= maxf) sw = -sw;
    else if (fr <= minf) sw = -sw;
    else if (fr >= 0) sw = Real<T>::abs(sw);
    else sw = -Real<T>::abs(sw);
}
// the same step as selects (no exec-mask branches), applied only when `on`
template <typename T> __device__ __forceinline__ void pll_sweep_sel(T &fr, T &sw, T maxf, T minf, bool on)
{
    const T f2 = fr + sw;
    const T mag = Real<T>::abs(sw);
    const T by_sign = (f2 >= 0) ? mag : -mag;
    const T s2 = (f2 >= maxf || f2 <= minf) ? -sw : by_sign;
    fr = on ? f2 : fr;
    sw = on ? s2 : sw;
}

// Four samples of the acquisition's loop filter (float, plain wrap) as ONE block of machine code: pll_phase_step's 16 operations
// (the block of pll_vec4_asm below, same order, same results) and, with the sweep gate open, pll_sweep_sel's step in four:
//   f2 = fr + sw;  t = |sw| with the sign of f2 (v_bfi);  rail = |f2| >= max;  sw = rail ? -sw : t
// -- the reference's four-way if (CarrierTrackingPLL.c:239-252) with min = -max (make_pll_params: the two limits are one product
// and its negation) and f2 never -0 (a sum is -0 only if both terms are, and the sweep step is not zero).  Round 6: the loop is the
// pace of everything in front of the lock -- a lone wavefront, every operation an issue slot of ~2.2 ns -- and took 25 slots a
// sample: theta by v_readlane, the phase kept by compare + select, seven operations of sweep logic.  Here theta arrives four
// samples at a time by one broadcast LDS read, the four phases leave by one LDS write, and the sweep is four operations:
// 20.5 slots.  pb[k] = the phase sample k was mixed with (what the detectors need); phase / freq / sweep = the state after the
// fourth sample.
template <bool OPEN>
__device__ __forceinline__ void acq_vec4_asm(const Vec16<float> &th, float &phase, float &freq, float &sweep, float (&pb)[4], float alpha,
                                             float beta, float minf, float maxf_v)
{
    const float hi = 6.2831854820251465f, d = 1.7484555314695172e-07f;
    float t1, t2, e, f1, p0, p1, p2, p3;
    // (a vector compare's mask wants two other instructions before it is read: inside the block nobody inserts wait states, so
    // the select that ends a sweep step stands behind the first operation of the next sample's step)
#define PDT_ACQ_HEAD(TH, P) "v_sub_f32 %[e], " TH ", " P "\n\t"
#define PDT_ACQ_REST(P, PN)                                             \
    "v_cmp_ge_f32_e64 vcc, |%[e]|, %[pi]\n\t"                           \
    "v_bfi_b32 %[t1], %[mask], 1.0, %[e]\n\t"                           \
    "v_fma_f32 %[t2], %[t1], %[nhi], %[e]\n\t"                          \
    "v_fma_f32 %[t2], %[t1], %[d], %[t2]\n\t"                           \
    "v_cndmask_b32_e32 %[e], %[e], %[t2], vcc\n\t"                      \
    "v_mul_f32 %[t1], %[beta], %[e]\n\t"                                \
    "v_add_f32 %[f1], %[fr], %[t1]\n\t"                                 \
    "v_add_f32 %[t2], " P ", %[f1]\n\t"                                  \
    "v_mul_f32 %[t1], %[alpha], %[e]\n\t"                               \
    "v_add_f32 %[t2], %[t2], %[t1]\n\t"                                 \
    "v_mul_f32 %[t1], 0x3e22f983, %[t2]\n\t"                            \
    "v_trunc_f32 %[t1], %[t1]\n\t"                                      \
    "v_fma_f32 %[t2], %[t1], %[nhi], %[t2]\n\t"                         \
    "v_fma_f32 " PN ", %[t1], %[d], %[t2]\n\t"                           \
    "v_med3_f32 %[fr], %[f1], %[minf], %[maxf]\n\t"
#define PDT_ACQ_SWEEP_A                                                 \
    "v_add_f32 %[fr], %[fr], %[sw]\n\t"                                 \
    "v_cmp_ge_f32_e64 vcc, |%[fr]|, %[maxf]\n\t"                        \
    "v_bfi_b32 %[t1], %[mask], %[sw], %[fr]\n\t"
#define PDT_ACQ_SWEEP_B "v_cndmask_b32_e64 %[sw], %[t1], -%[sw], vcc\n\t"
    pb[0] = phase;
    if constexpr (OPEN)
        asm volatile(PDT_ACQ_HEAD("%[th0]", "%[ph]") PDT_ACQ_REST("%[ph]", "%[p0]") PDT_ACQ_SWEEP_A
                     PDT_ACQ_HEAD("%[th1]", "%[p0]") PDT_ACQ_SWEEP_B PDT_ACQ_REST("%[p0]", "%[p1]") PDT_ACQ_SWEEP_A
                     PDT_ACQ_HEAD("%[th2]", "%[p1]") PDT_ACQ_SWEEP_B PDT_ACQ_REST("%[p1]", "%[p2]") PDT_ACQ_SWEEP_A
                     PDT_ACQ_HEAD("%[th3]", "%[p2]") PDT_ACQ_SWEEP_B PDT_ACQ_REST("%[p2]", "%[p3]") PDT_ACQ_SWEEP_A
                     "s_nop 0\n\t" PDT_ACQ_SWEEP_B
                     : [fr] "+v"(freq), [sw] "+v"(sweep), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3), [t1] "=&v"(t1),
                       [t2] "=&v"(t2), [e] "=&v"(e), [f1] "=&v"(f1)
                     : [th0] "v"(th.v[0]), [th1] "v"(th.v[1]), [th2] "v"(th.v[2]), [th3] "v"(th.v[3]), [ph] "v"(phase),
                       [pi] "s"(3.14159274101257324f), [mask] "s"(0x7fffffffu), [nhi] "s"(-hi), [d] "s"(d), [alpha] "s"(alpha), [beta] "s"(beta),
                       [minf] "s"(minf), [maxf] "v"(maxf_v)
                     : "vcc");
    else
        asm volatile(PDT_ACQ_HEAD("%[th0]", "%[ph]") PDT_ACQ_REST("%[ph]", "%[p0]")
                     PDT_ACQ_HEAD("%[th1]", "%[p0]") PDT_ACQ_REST("%[p0]", "%[p1]")
                     PDT_ACQ_HEAD("%[th2]", "%[p1]") PDT_ACQ_REST("%[p1]", "%[p2]")
                     PDT_ACQ_HEAD("%[th3]", "%[p2]") PDT_ACQ_REST("%[p2]", "%[p3]")
                     : [fr] "+v"(freq), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3), [t1] "=&v"(t1),
                       [t2] "=&v"(t2), [e] "=&v"(e), [f1] "=&v"(f1)
                     : [th0] "v"(th.v[0]), [th1] "v"(th.v[1]), [th2] "v"(th.v[2]), [th3] "v"(th.v[3]), [ph] "v"(phase),
                       [pi] "s"(3.14159274101257324f), [mask] "s"(0x7fffffffu), [nhi] "s"(-hi), [d] "s"(d), [alpha] "s"(alpha), [beta] "s"(beta),
                       [minf] "s"(minf), [maxf] "v"(maxf_v)
                     : "vcc");
#undef PDT_ACQ_HEAD
#undef PDT_ACQ_REST
#undef PDT_ACQ_SWEEP_A
#undef PDT_ACQ_SWEEP_B
    pb[1] = p0; pb[2] = p1; pb[3] = p2;
    phase = p3;
}

// Acquisition, pipelined over two wavefronts.  The stream is taken in batches, sample k of a batch in lane k.  The (phase, freq)
// loop filter of a batch -- serial, run under the hypothesis that the sweep condition |pi/2 - averagePhase| < 0.05 keeps the value
// it had (it changes a handful of times per capture) -- depends on the detector passes (lane-parallel sincos / mix / arctan2,
// then the two serial EMAs with the sweep condition and the lock test of every sample) only through the two rare events "sweep gate
// flipped" and "locked"; so wavefront 0 runs the loop filter of batch t while wavefront 1 evaluates
// the detectors of batch t-1 from the per-sample states wavefront 0 left in LDS.  When wavefront 1
// reports an event inside batch t-1, the speculative batch t is dropped and wavefront 0 resumes from
// the corrected state after the event sample.  Same arithmetic per sample, half the time per batch.
// Round 6: the float build with the plain wrap (POES) takes batches of 128 samples, two per lane ("halves" h = 0, 1: sample
// 64 h + lane): the per-batch costs -- two barriers, the hand-over through LDS, the prefetch of the next batch -- were a fifth
// of the acquisition's time at 64.  The double build and the slow-wrap variants keep 64.
template <typename T, int NB> struct alignas(16) AcqSlot {
    T phi[NB];                        // phase used for sample k (the value before its update)
    long long i0;
    int nb, valid, hyp;
    T ph_beg, fr_beg, sw_beg;         // loop-filter state in front of the batch (an event replays the filter from here)
    T ph_end, fr_end, sw_end;         // ... and after the whole batch under the hypothesis
};
template <typename T> struct AcqVerdict {
    int event;                         // 0 = the batch stands, 1 = it ended early at sample k
    int k, hyp, locked;
    T phase, freq, sweep;              // loop-filter state after sample k (sweep step taken with the true gate)
};

template <typename T, bool SLOW, bool EXCL = false>
__device__ __forceinline__ void k_pll_acquire_pipe(IqSrc pcm, long long n, PllParams<T> P,
                                                          T *__restrict__ out, T *__restrict__ lock_out,
                                                          PllLockInfo<T> *__restrict__ info, T *__restrict__ avg_out = nullptr)
{
    // FAST (float, plain wrap): theta of the batch being filtered read back four samples at a time (one broadcast LDS read in
    // place of four v_readlane, acq_vec4_asm); the detectors' input terms as doubles and the two EMAs' values per sample likewise
    constexpr bool FAST = std::is_same<T, float>::value && !SLOW;
    constexpr int H = FAST ? 2 : 1, NB = 64 * H;
    __shared__ AcqSlot<T, NB> slot[2];
    __shared__ AcqVerdict<T> verdict;
    __shared__ __attribute__((aligned(16))) float s_theta[FAST ? NB : 4];
    __shared__ __attribute__((aligned(16))) double s_tu[FAST ? 2 * NB : 2];
    __shared__ __attribute__((aligned(16))) float s_ema[FAST ? 2 * NB : 2];
    // EXCL: claim a whole SIMD's register file per wavefront (256 + 256 registers), so that the dispatcher can only put
    // these serial wavefronts on SIMDs that hold no wavefront of the concurrent block-parallel kernel -- sharing issue
    // slots with one costs them up to 15 %.  Only requested while that kernel leaves SIMDs free (the host decides).
    if (EXCL) asm volatile("" ::: "v255", "a255");
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const T avg_alpha = (T)0.00005;
    const double k_avg = 1.0 - (double)avg_alpha, k_lock = 1.0 - (double)P.lock_alpha;
    // wavefront 0: loop-filter state; wavefront 1: detector state
    T phase = P.phase0, freq = P.freq0, sweep = P.sweep0;
    T avg = P.avg0, locksig = P.locksig0;
    bool hyp = avg >= P.cond_lo && avg <= P.cond_hi;
    long long i_prod = P.i0;           // next sample the loop filter will take
    long long i_pre = -1;              // start of the batch th_pre was loaded for
    T th_pre[H];
    long long iq_pre = -1;             // (wavefront 1) start of the batch a_pre / b_pre were loaded for
    T a_pre[H], b_pre[H];
#pragma unroll
    for (int h = 0; h < H; h++) { th_pre[h] = 0; a_pre[h] = 0; b_pre[h] = 0; }
    long long lock_at = -1;
    T freq_at_lock = 0, avg_at_lock = P.avg0;
    T fin_phase = P.phase0, fin_freq = P.freq0, fin_sweep = P.sweep0;      // loop-filter state at the end (kept by wavefront 1)
    if (threadIdx.x < 2) slot[threadIdx.x].valid = 0;
    if (threadIdx.x == 0) verdict.event = 0;
    __syncthreads();
    for (int t = 0;; t++) {
        AcqSlot<T, NB> &mine = slot[t & 1];
        AcqSlot<T, NB> &theirs = slot[(t & 1) ^ 1];
        const bool have_prev = theirs.valid != 0;
        const bool produce = i_prod < n;
        if (!have_prev && !produce) break;
        if (wave == 0) {
            // ---- loop filter of the batch starting at i_prod
            if (produce) {
                const int nb = (int)((n - i_prod < NB) ? (n - i_prod) : NB);
                // theta of this batch was requested one batch ago (unless an event moved the start)
                T th_l[H];
#pragma unroll
                for (int h = 0; h < H; h++) {
                    th_l[h] = th_pre[h];
                    if (i_pre != i_prod) th_l[h] = (64 * h + lane < nb) ? theta_of<T>(pcm, i_prod + 64 * h + lane) : (T)0;
                }
                i_pre = i_prod + nb;
#pragma unroll
                for (int h = 0; h < H; h++)
                    if (i_pre + 64 * h + lane < n) th_pre[h] = theta_of<T>(pcm, i_pre + 64 * h + lane);
                T ph = phase, fr = freq, sw = sweep;
                T phi_l[H];
#pragma unroll
                for (int h = 0; h < H; h++) phi_l[h] = 0;
                bool filtered = false;
                if constexpr (FAST) {
                    if (nb == NB) {
                        // the whole batch in blocks of four samples (acq_vec4_asm), no branch: theta of block g + 1 is requested
                        // from LDS before block g runs (the LDS answers a wavefront's requests in order: the writes below are
                        // seen by the reads behind them)
#pragma unroll
                        for (int h = 0; h < H; h++) s_theta[64 * h + lane] = th_l[h];
                        float maxf_v = P.max_freq;
                        asm volatile("" : "+v"(maxf_v));
                        const Vec16<float> *tq = reinterpret_cast<const Vec16<float> *>(s_theta);
                        Vec16<float> *pq = reinterpret_cast<Vec16<float> *>(mine.phi);
                        Vec16<float> cur = tq[0];
                        if (hyp) {
#pragma unroll
                            for (int g = 0; g < NB / 4; g++) {
                                const Vec16<float> nxt = tq[(g + 1) & (NB / 4 - 1)];
                                Vec16<float> pv;
                                acq_vec4_asm<true>(cur, ph, fr, sw, pv.v, P.alpha_acq, P.beta_acq, P.min_freq, maxf_v);
                                pq[g] = pv;
                                cur = nxt;
                            }
                        } else {
#pragma unroll
                            for (int g = 0; g < NB / 4; g++) {
                                const Vec16<float> nxt = tq[(g + 1) & (NB / 4 - 1)];
                                Vec16<float> pv;
                                acq_vec4_asm<false>(cur, ph, fr, sw, pv.v, P.alpha_acq, P.beta_acq, P.min_freq, maxf_v);
                                pq[g] = pv;
                                cur = nxt;
                            }
                        }
                        filtered = true;
                    }
                }
                if (!filtered) {
                    // (four samples per trip: a taken branch costs a lone wavefront as much as nine instructions.  This loop keeps
                    // only what the detectors need of every sample, the phase it was mixed with: the states around an event sample
                    // are replayed from the front of the batch by the wavefront that finds the event (a handful per capture), and
                    // the sweep step is compiled in or out with the gate's hypothesis instead of selected per sample.)
#pragma unroll
                    for (int h = 0; h < H; h++) {
                        const int nh = (nb - 64 * h < 64) ? nb - 64 * h : 64;         // samples of this half (may be <= 0)
                        const T thh = th_l[h];
                        T phl = 0;
                        auto filt_open = [&](int k) {
                            const T th = lane_get(thh, k);
                            phl = (lane == k) ? ph : phl;
                            pll_phase_step<T, SLOW>(th, ph, fr, P.alpha_acq, P.beta_acq, P.max_freq, P.min_freq);
                            pll_sweep_sel(fr, sw, P.max_freq, P.min_freq, true);       // (the select form: no branches in the chain)
                        };
                        auto filt_closed = [&](int k) {
                            const T th = lane_get(thh, k);
                            phl = (lane == k) ? ph : phl;
                            pll_phase_step<T, SLOW>(th, ph, fr, P.alpha_acq, P.beta_acq, P.max_freq, P.min_freq);
                        };
                        int k = 0;
                        if (hyp) {
                            for (; k + 4 <= nh; k += 4) { filt_open(k); filt_open(k + 1); filt_open(k + 2); filt_open(k + 3); }
                            for (; k < nh; k++) filt_open(k);
                        } else {
                            for (; k + 4 <= nh; k += 4) { filt_closed(k); filt_closed(k + 1); filt_closed(k + 2); filt_closed(k + 3); }
                            for (; k < nh; k++) filt_closed(k);
                        }
                        phi_l[h] = phl;
                    }
#pragma unroll
                    for (int h = 0; h < H; h++)
                        if (64 * h + lane < nb) mine.phi[64 * h + lane] = phi_l[h];
                }
                if (lane == 0) {
                    mine.i0 = i_prod; mine.nb = nb; mine.hyp = hyp ? 1 : 0; mine.valid = 1;
                    mine.ph_beg = phase; mine.fr_beg = freq; mine.sw_beg = sweep;
                    mine.ph_end = ph; mine.fr_end = fr; mine.sw_end = sw;
                }
            } else if (lane == 0) {
                mine.valid = 0;
            }
        } else {
            // ---- detectors of the previous batch
            if (have_prev) {
                const long long i0 = theirs.i0;
                const int nb = theirs.nb;
                const bool h_open = theirs.hyp != 0;
                T t_l[H], u_l[H], o_l[H], a_l[H], b_l[H];
                // the IQ samples of this batch were requested one batch ago (unless an event moved the start); those of
                // the batch wavefront 0 is filtering now are requested here
#pragma unroll
                for (int h = 0; h < H; h++) {
                    t_l[h] = 0; u_l[h] = 0; o_l[h] = 0;
                    a_l[h] = a_pre[h]; b_l[h] = b_pre[h];
                    if (iq_pre != i0 && 64 * h + lane < nb) IqSample<T>::get(pcm, i0 + 64 * h + lane, a_l[h], b_l[h]);
                }
                iq_pre = -1;
                if (produce) {
                    iq_pre = i_prod;
#pragma unroll
                    for (int h = 0; h < H; h++)
                        if (i_prod + 64 * h + lane < n) IqSample<T>::get(pcm, i_prod + 64 * h + lane, a_pre[h], b_pre[h]);
                }
#pragma unroll
                for (int h = 0; h < H; h++)
                    if (64 * h + lane < nb) {
                        T t_real, t_imag;
                        Real<T>::sincos(theirs.phi[64 * h + lane], t_imag, t_real);
                        const T c = t_real, d = -t_imag;
                        const T o_re = a_l[h] * c - b_l[h] * d;
                        const T o_im = a_l[h] * d + b_l[h] * c;
                        o_l[h] = o_im;
                        const T ph = arctan2_ref(o_im, o_re);
                        t_l[h] = avg_alpha * Real<T>::abs(ph);
                        const T mag2 = a_l[h] * a_l[h] + b_l[h] * b_l[h];
                        const T inv = (T)q_rsqrt((float)mag2);
                        const T re = a_l[h] * inv, im = b_l[h] * inv;
                        u_l[h] = P.lock_alpha * (re * t_real + im * t_imag);
                    }
                // the serial part is the two EMAs only (a wavefront's pace is its instruction count); the sweep gate and
                // the lock test of sample k are evaluated afterwards by the lane that kept the EMA values of that sample
                T av = avg, ls = locksig, av_l[H], ls_l[H];
#pragma unroll
                for (int h = 0; h < H; h++) { av_l[h] = 0; ls_l[h] = 0; }
                bool ema_done = false;
                if constexpr (FAST) {
                    // the input terms of both EMAs as doubles side by side (one broadcast LDS read per sample in place of two
                    // v_readlane and two conversions), the two values after every sample back through LDS (one write in place of
                    // a compare and two selects): 14 issue slots a sample instead of 20 -- this wavefront must stay ahead of the
                    // loop filter's
                    if (nb == NB) {
#pragma unroll
                        for (int h = 0; h < H; h++) {
                            double2 tu;
                            tu.x = (double)t_l[h];
                            tu.y = (double)u_l[h];
                            reinterpret_cast<double2 *>(s_tu)[64 * h + lane] = tu;
                        }
                        const double2 *tq = reinterpret_cast<const double2 *>(s_tu);
                        float2 *eq = reinterpret_cast<float2 *>(s_ema);
#pragma unroll 16
                        for (int k = 0; k < NB; k++) {
                            const double2 in = tq[k];
                            av = (T)((double)av * k_avg + in.x);
                            ls = (T)((double)ls * k_lock + in.y);
                            float2 o;
                            o.x = (float)av;
                            o.y = (float)ls;
                            eq[k] = o;
                        }
#pragma unroll
                        for (int h = 0; h < H; h++) {
                            const float2 mine_e = eq[64 * h + lane];
                            av_l[h] = (T)mine_e.x;
                            ls_l[h] = (T)mine_e.y;
                        }
                        ema_done = true;
                    }
                }
                if (!ema_done) {
#pragma unroll
                    for (int h = 0; h < H; h++) {
                        const int nh = (nb - 64 * h < 64) ? nb - 64 * h : 64;
                        const T tl = t_l[h], ul = u_l[h];
                        T avl = 0, lsl = 0;
                        auto ema = [&](int k) {
                            av = (T)((double)av * k_avg + (double)lane_get(tl, k));
                            ls = (T)((double)ls * k_lock + (double)lane_get(ul, k));
                            const bool me = lane == k;
                            lsl = me ? ls : lsl;
                            avl = me ? av : avl;
                        };
                        int k = 0;
                        for (; k + 4 <= nh; k += 4) { ema(k); ema(k + 1); ema(k + 2); ema(k + 3); }
                        for (; k < nh; k++) ema(k);
                        av_l[h] = avl;
                        ls_l[h] = lsl;
                    }
                }
                // the first event of the batch: a flip of the sweep gate or the lock, whichever sample comes first
                int ev_k = -1;
                bool ev_is_flip = false, ev_is_lock = false;
#pragma unroll
                for (int h = 0; h < H; h++) {
                    const bool in_batch = 64 * h + lane < nb;
                    const bool cond_l = av_l[h] >= P.cond_lo && av_l[h] <= P.cond_hi;
                    const unsigned long long ev_flip = __ballot(in_batch && cond_l != h_open);
                    const unsigned long long ev_lock = __ballot(in_batch && ls_l[h] > P.lock_thr);
                    const unsigned long long ev = ev_flip | ev_lock;
                    if (ev_k < 0 && ev) {
                        const int kk = __builtin_ctzll(ev);
                        ev_k = 64 * h + kk;
                        ev_is_flip = ((ev_flip >> kk) & 1ull) != 0;
                        ev_is_lock = ((ev_lock >> kk) & 1ull) != 0;
                    }
                }
                int done = nb;
                if (ev_k >= 0) {
                    const int k = ev_k;
                    const bool cond = ev_is_flip ? !h_open : h_open;
                    // the loop filter again from the front of the batch up to the event sample: the gate kept its hypothesis
                    // for the samples before it, the event sample takes the true one
                    T rp = theirs.ph_beg, fr = theirs.fr_beg, sw = theirs.sw_beg;
#pragma unroll
                    for (int h = 0; h < H; h++) {
                        const T thr_l = (64 * h + lane < nb) ? arctan2_ref(b_l[h], a_l[h]) : (T)0;        // theta_of of this lane's sample
                        const int q_hi = (k - 64 * h < 63) ? k - 64 * h : 63;                                 // last sample of this half to replay
                        for (int q = 0; q <= q_hi; q++) {
                            pll_phase_step<T, SLOW>(lane_get(thr_l, q), rp, fr, P.alpha_acq, P.beta_acq, P.max_freq, P.min_freq);
                            if (64 * h + q < k && h_open) pll_sweep_step(fr, sw, P.max_freq, P.min_freq);
                        }
                    }
                    if (cond) pll_sweep_step(fr, sw, P.max_freq, P.min_freq);
                    done = k + 1;
#pragma unroll
                    for (int h = 0; h < H; h++)
                        if ((k >> 6) == h) {
                            avg = lane_get(av_l[h], k & 63);
                            locksig = lane_get(ls_l[h], k & 63);
                        }
                    fin_phase = rp; fin_freq = fr; fin_sweep = sw;
                    if (ev_is_lock) {
                        lock_at = i0 + k;
                        freq_at_lock = fr;
                        avg_at_lock = avg;
                    }
                    if (lane == 0) {
                        verdict.event = 1; verdict.k = k; verdict.hyp = cond ? 1 : 0; verdict.locked = ev_is_lock ? 1 : 0;
                        verdict.phase = fin_phase; verdict.freq = fr; verdict.sweep = sw;
                    }
                } else {
                    avg = av;
                    locksig = ls;
                    fin_phase = theirs.ph_end; fin_freq = theirs.fr_end; fin_sweep = theirs.sw_end;
                    if (lane == 0) verdict.event = 0;
                }
#pragma unroll
                for (int h = 0; h < H; h++)
                    if (64 * h + lane < done) {
                        out[i0 + 64 * h + lane] = o_l[h];
                        if (lock_out) lock_out[i0 + 64 * h + lane] = ls_l[h];
                        if (avg_out) avg_out[i0 + 64 * h + lane] = av_l[h];    // averagePhase after this sample (:124,152; the value :277 returns)
                    }
            } else if (lane == 0) {
                verdict.event = 0;
            }
        }
        __syncthreads();
        // ---- hand-over
        const AcqVerdict<T> v = verdict;
        const bool ended = have_prev && v.event && v.locked;
        if (have_prev && v.event) {
            // the previous batch ended early at sample k: drop the speculative batch and resume after it
            i_prod = theirs.i0 + v.k + 1;
            phase = v.phase; freq = v.freq; sweep = v.sweep;
            hyp = v.hyp != 0;
            if (threadIdx.x == 0) mine.valid = 0;
        } else if (produce) {
            i_prod = mine.i0 + mine.nb;
            phase = mine.ph_end; freq = mine.fr_end; sweep = mine.sw_end;
        }
        __syncthreads();
        if (ended) break;
    }
    if (wave == 1 && lane == 0) {
        PllState<T> st;
        st.phase = fin_phase; st.freq = fin_freq; st.avg_phase = avg; st.locksig = locksig; st.sweep = fin_sweep;
        info->lock_sample = lock_at;
        info->st = st;
        info->freq_at_lock = freq_at_lock;
        info->avg_at_lock = avg_at_lock;
    }
}

template <typename T> struct PllSeam {
    T phase0, freq0;   // state at the block's official start (after warm-up)
    T phase1, freq1;   // state after the block's last sample
};

#ifndef PDT_PF
#define PDT_PF 8   // look-ahead depth (vectors per lane) of the lane-per-block stream walkers
#endif

// Look-ahead ring of a lane-per-block stream walker, hand-issued.  Every lane streams its own block, so its loads
// cannot coalesce and must be issued far ahead; left to the compiler, the re-loads of an unrolled register ring are
// moved around freely (hoisted above the arithmetic, sunk across the back edge) and the loop header waits for all of
// them: each trip then exposes a memory latency.  Here the data in flight never live in compiler-visible registers:
// ring_issue sends 16 bytes per lane from global memory straight into LDS (slot base + 16 * lane), ring_wait<N> is
// an explicit `s_waitcnt vmcnt(N)` (the counter covers loads and stores, oldest first, so N = number of ring
// requests known to be younger is always enough), and the walker then reads its own 16 bytes of the slot with an
// ordinary LDS load.  Both carry a memory clobber: the compiler keeps LDS reads behind the wait that guards them.
#define PDT_RING_SLOT 1024        // bytes per slot: 64 lanes x 16
__device__ __forceinline__ void ring_issue(const void *g, unsigned lds_slot_addr)
{
    // (s_nop: wait state between writing M0 and the LDS-direct load that uses it.  M0 is reserved by the compiler, which
    // loads it immediately before each of its own uses and never keeps a value in it: tests/test_abi.py checks that the
    // kernels using this contain no other reference to m0.)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_slot_addr) : "memory");
}
template <int N> __device__ __forceinline__ void ring_wait()
{
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// run the recurrence over the natural samples [i0, i1) of the LT theta stream, optionally storing the pre-update phase of
// every sample -- into the LT phase stream (OUT_LT: out is its base) or into a linear buffer indexed by the natural sample
// index (out[i]; the single-lane walkers of the head and of the seam repairs).  Block by block: inside a block a lane's
// consecutive 16-byte vectors are one row (ROW elements) apart.  VOTE also counts, over the range, the samples whose
// detector error lies beyond +-pi/2 (the basin vote of the warm-up).
#ifndef PDT_PLL_PF
#define PDT_PLL_PF 48  // look-ahead of the block-parallel PLL walkers: every load is a fresh KiB from L2 / HBM (no line is touched twice),
                       // measured 1.29 / 1.11 / 1.05 / 1.02 ms at 4 / 8 / 16 / 24 vectors (bench capture); an hour at 250 ksps, where
                       // the warm-ups' re-reads keep HBM at 4 TB/s and a load takes longer: 6.3 / 6.06 / 6.2 ms at 24 / 48 / 64 (the
                       // ten-minute capture is indifferent: 0.85 / 0.84 / 0.88).  One wavefront per SIMD: the 192 registers are free.
#endif
// Round 4: the look-ahead of the block-parallel walkers and of the head as an LDS ring (`ring` != nullptr: RPF slots of 1 KiB
// for this wavefront).  With the ring in registers the compiler's loop header waits for vmcnt(0) -- every trip of PF vectors
// exposes one whole memory latency (the listing under profiles/r4: `s_waitcnt vmcnt(0)` at the top of the 48-vector loop; 27 %
// of the kernel's wave cycles were memory waits at an hour of 250 ksps, where HBM is busy with the warm-ups' re-reads and a
// load takes microseconds).  In the LT layout a wavefront's load is one KiB of consecutive bytes with lane l's 16 bytes at
// l * 16: exactly where `global_load_lds_dwordx4` puts them.  Vector k + 1 is read from LDS while vector k is walked, after
// `s_waitcnt vmcnt(RPF - 2)` (the loads younger than it; loads complete in order among themselves, the stores in the
// counter only make the wait conservative), and slot k is refilled for vector k + RPF as soon as k has been walked.
#ifndef PDT_PLL_RING_PF
#define PDT_PLL_RING_PF 32
#endif
// Four steps of the float loop filter (pll_phase_step, the 16-operation form) as ONE block of machine code, with the refill
// of the look-ahead slot the vector came from: `s_mov_b32 m0` first, the LDS-direct load last.  Why by hand: the walkers are
// bound by instruction issue (tools/probes/pll_mem_probe.hip: 34 ns a step with theta in registers = 81 clocks = 20 issue
// slots where the arithmetic is 16), and the compiler puts an `s_nop` behind every inline-assembly block it cannot look into
// -- one slot per step with the error wrap as its own block, one more per vector behind the write of M0.  (Measured and not
// used: the two products beta e and alpha e as one `v_pk_mul_f32` -- a packed f32 operation takes two issue slots, 8.5 clocks
// dependent, tools/probes/lat_probe.hip.)  Same operations in the same order as pll_wrap_error_f32 / pll_wrap_phase_f32 /
// pll_phase_step (pdt_device_math.h): every result is the one they give, the GPU suite compares the streams bit for bit.
// p[k] = the phase after step k (p[3] is the new state; the phases BEFORE the steps -- what the kernels store -- are phase,
// p[0], p[1], p[2]).
__device__ __forceinline__ void pll_vec4_asm(const Vec16<float> &th, float phase, float &freq, float (&p)[4], float alpha, float beta,
                                             float minf, float maxf_v, const void *refill, unsigned slot_addr)
{
    const float hi = 6.2831854820251465f, d = 1.7484555314695172e-07f;
    float t1, t2, e, f1, fa, fb, fc;
#define PDT_PLL_STEP(TH, P, F, PN, FN)                                  \
    "v_sub_f32 %[e], " TH ", " P "\n\t"                                  \
    "v_cmp_ge_f32_e64 vcc, |%[e]|, %[pi]\n\t"                           \
    "v_bfi_b32 %[t1], %[mask], 1.0, %[e]\n\t"                           \
    "v_fma_f32 %[t2], %[t1], %[nhi], %[e]\n\t"                          \
    "v_fma_f32 %[t2], %[t1], %[d], %[t2]\n\t"                           \
    "v_cndmask_b32_e32 %[e], %[e], %[t2], vcc\n\t"                      \
    "v_mul_f32 %[t1], %[beta], %[e]\n\t"                                \
    "v_add_f32 %[f1], " F ", %[t1]\n\t"                                  \
    "v_add_f32 %[t2], " P ", %[f1]\n\t"                                  \
    "v_mul_f32 %[t1], %[alpha], %[e]\n\t"                               \
    "v_add_f32 %[t2], %[t2], %[t1]\n\t"                                 \
    "v_mul_f32 %[t1], 0x3e22f983, %[t2]\n\t"                            \
    "v_trunc_f32 %[t1], %[t1]\n\t"                                      \
    "v_fma_f32 %[t2], %[t1], %[nhi], %[t2]\n\t"                         \
    "v_fma_f32 " PN ", %[t1], %[d], %[t2]\n\t"                           \
    "v_med3_f32 " FN ", %[f1], %[minf], %[maxf]\n\t"
    asm volatile("s_mov_b32 m0, %[slot]\n\t"
                 PDT_PLL_STEP("%[th0]", "%[ph]", "%[fr]", "%[p0]", "%[fa]")
                 PDT_PLL_STEP("%[th1]", "%[p0]", "%[fa]", "%[p1]", "%[fb]")
                 PDT_PLL_STEP("%[th2]", "%[p1]", "%[fb]", "%[p2]", "%[fc]")
                 PDT_PLL_STEP("%[th3]", "%[p2]", "%[fc]", "%[p3]", "%[fr]")
                 "global_load_lds_dwordx4 %[g], off"
                 : [fr] "+v"(freq), [p0] "=&v"(p[0]), [p1] "=&v"(p[1]), [p2] "=&v"(p[2]), [p3] "=&v"(p[3]), [t1] "=&v"(t1), [t2] "=&v"(t2),
                   [e] "=&v"(e), [f1] "=&v"(f1), [fa] "=&v"(fa), [fb] "=&v"(fb), [fc] "=&v"(fc)
                 : [th0] "v"(th.v[0]), [th1] "v"(th.v[1]), [th2] "v"(th.v[2]), [th3] "v"(th.v[3]), [ph] "v"(phase),
                   [pi] "s"(3.14159274101257324f), [mask] "s"(0x7fffffffu), [nhi] "s"(-hi), [d] "s"(d), [alpha] "s"(alpha), [beta] "s"(beta),
                   [minf] "s"(minf), [maxf] "v"(maxf_v), [g] "v"(refill), [slot] "s"(slot_addr)
                 : "vcc", "memory");
#undef PDT_PLL_STEP
}

// `ckl` (STORE, from a block's first sample, ring path only): this lane's column of the frequency checkpoints -- the loop
// frequency in front of sample c * PDT_PLL_CKPT of the block goes to ckl[64 c] for every c >= 1 that begins a trip of the ring
// loop (pll_ckpt_valid).  With the phase stream they let a seam repair see where its re-run has merged with the stored
// trajectory (k_pll_fix): the state (phase, freq) is the whole memory of the loop.
#define PDT_PLL_CKPT 1024
template <typename T> __host__ __device__ __forceinline__ bool pll_ckpt_valid(long long c, long long len)
{
    return c >= 1 && c * PDT_PLL_CKPT + (long long)PDT_PLL_RING_PF * (16 / (long long)sizeof(T)) <= len;
}
// checkpoints per block (the stride of a tile's columns)
__host__ __device__ __forceinline__ long long pll_ckpt_count(long long B) { return B / PDT_PLL_CKPT + 1; }

template <typename T, bool STORE, bool SLOW, bool OUT_LT, int PF = PDT_PLL_PF, bool VOTE = false>
__device__ __forceinline__ void pll_phase_range(const T *__restrict__ theta_lt, T *__restrict__ out, long long B, long long i0,
                                                long long i1, T &phase, T &freq, T alpha, T beta, T maxf, T minf,
                                                int *far = nullptr, int *seen = nullptr, unsigned char *ring = nullptr,
                                                T *ckl = nullptr)
{
    constexpr int RPF = PDT_PLL_RING_PF;
    constexpr int VN = Lt<T>::VN, ROW = Lt<T>::ROW;
    constexpr int OSTR = OUT_LT ? ROW : VN;
    auto step = [&](T th) {
        if (VOTE) {
            T d = th - phase;
            if (d > (T)PDT_PI) d -= (T)(2 * PDT_PI);
            if (d < (T)-PDT_PI) d += (T)(2 * PDT_PI);
            *far += (Real<T>::abs(d) > (T)(PDT_PI / 2)) ? 1 : 0;
            *seen += 1;
        }
        pll_phase_step<T, SLOW>(th, phase, freq, alpha, beta, maxf, minf);
    };
    long long i = i0;
    while (i < i1) {
        const long long j = i / B, p0 = i - j * B;
        const long long seg_end = ((j + 1) * B < i1) ? (j + 1) * B : i1;
        long long cnt = seg_end - i;                       // samples of this block still to walk
        int e = (int)(p0 % VN);
        const long long lt_off = (j >> 6) * (64 * B) + (p0 / VN) * ROW + (j & 63) * VN;    // vector that holds sample i
        const T *tp = theta_lt + lt_off;
        T *op = nullptr;
        if (STORE) op = OUT_LT ? out + lt_off : out + (i - e);
        if (e != 0) {                                      // leading partial vector
            for (; e < VN && cnt > 0; e++, cnt--) {
                if (STORE) op[e] = phase;
                step(tp[e]);
            }
            tp += ROW;
            if (STORE) op += OSTR;
        }
        long long nv = cnt / VN;                           // whole vectors
        cnt -= nv * VN;
        // Software pipeline: PF vectors per lane are always in flight; each register set is re-loaded right after it has
        // been consumed and is next needed PF-1 vectors later, so the recurrence never waits on memory.  (Look-ahead loads
        // run up to PF rows past the segment: every LT buffer is allocated with that much slack.  The opaque offset stops
        // the compiler from sinking the look-ahead load back to its use.)  On gfx9 stores share the vmcnt counter with
        // loads, so waiting for a look-ahead load also waits for every older store: the single-lane walkers use PF = 32.
        if (ring && nv >= RPF) {
            const unsigned ring0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)ring);   // (one ring per wavefront)
            const unsigned char *mine = ring + 16 * (threadIdx.x & 63);
#pragma unroll
            for (int u = 0; u < RPF; u++) ring_issue(tp + (long long)u * ROW, ring0 + u * PDT_RING_SLOT);
            ring_wait<RPF - 1>();
            Vec16<T> cur = *reinterpret_cast<const Vec16<T> *>(mine), nxt;
            long long v = 0;
            for (; v + RPF <= nv; v += RPF) {
                if (STORE && ckl && v != 0 && (v & (PDT_PLL_CKPT / VN - 1)) == 0) ckl[(v / (PDT_PLL_CKPT / VN)) * 64] = freq;
#pragma unroll
                for (int u = 0; u < RPF; u++) {
                    ring_wait<RPF - 2>();
                    nxt = *reinterpret_cast<const Vec16<T> *>(mine + ((u + 1) % RPF) * PDT_RING_SLOT);
                    if constexpr (std::is_same<T, float>::value && !SLOW && !VOTE) {
                        float pn[4];
                        pll_vec4_asm(cur, phase, freq, pn, alpha, beta, minf, maxf, tp + (v + RPF + u) * ROW, ring0 + u * PDT_RING_SLOT);
                        if (STORE) {
                            Vec16<T> pv;
                            pv.v[0] = phase; pv.v[1] = pn[0]; pv.v[2] = pn[1]; pv.v[3] = pn[2];
                            *reinterpret_cast<Vec16<T> *>(op + (v + u) * OSTR) = pv;
                        }
                        phase = pn[3];
                    } else {
                        Vec16<T> pv;
#pragma unroll
                        for (int w = 0; w < VN; w++) {
                            pv.v[w] = phase;
                            step(cur.v[w]);
                        }
                        if (STORE) *reinterpret_cast<Vec16<T> *>(op + (v + u) * OSTR) = pv;
                        ring_issue(tp + (v + RPF + u) * ROW, ring0 + u * PDT_RING_SLOT);
                    }
                    cur = nxt;
                }
            }
            // the ring now holds vectors v .. v + RPF - 1 (cur = vector v): the remaining nv - v (< RPF) are already on their way
            ring_wait<0>();
            const int rest = (int)(nv - v);
            for (int u = 0; u < rest; u++) {
                if (u > 0) cur = *reinterpret_cast<const Vec16<T> *>(mine + u * PDT_RING_SLOT);
                Vec16<T> pv;
#pragma unroll
                for (int w = 0; w < VN; w++) {
                    pv.v[w] = phase;
                    step(cur.v[w]);
                }
                if (STORE) *reinterpret_cast<Vec16<T> *>(op + (v + u) * OSTR) = pv;
            }
        } else if (nv >= PF) {
            Vec16<T> buf[PF];
#pragma unroll
            for (int u = 0; u < PF; u++) buf[u] = *reinterpret_cast<const Vec16<T> *>(tp + (long long)u * ROW);
            long long v = 0;
            for (; v + PF <= nv; v += PF) {
#pragma unroll
                for (int u = 0; u < PF; u++) {
                    Vec16<T> pv;
#pragma unroll
                    for (int w = 0; w < VN; w++) {
                        pv.v[w] = phase;
                        step(buf[u].v[w]);
                    }
                    if (STORE) *reinterpret_cast<Vec16<T> *>(op + (v + u) * OSTR) = pv;
                    // reload the slot only after its last use: the new value can then live in the same
                    // registers (a reload issued earlier is copied at the loop end, behind a full wait)
                    long long q = (v + PF + u) * ROW;
                    asm volatile("" : "+v"(q));
                    buf[u] = *reinterpret_cast<const Vec16<T> *>(tp + q);
                }
            }
            // the ring now holds vectors v .. v + PF - 1: the remaining nv - v (< PF) of them are already here
            const int rest = (int)(nv - v);
#pragma unroll
            for (int u = 0; u < PF; u++) {
                if (u < rest) {
                    Vec16<T> pv;
#pragma unroll
                    for (int w = 0; w < VN; w++) {
                        pv.v[w] = phase;
                        step(buf[u].v[w]);
                    }
                    if (STORE) *reinterpret_cast<Vec16<T> *>(op + (v + u) * OSTR) = pv;
                }
            }
        } else {
            for (long long v = 0; v < nv; v++) {
                const Vec16<T> tv = *reinterpret_cast<const Vec16<T> *>(tp + v * ROW);
                Vec16<T> pv;
#pragma unroll
                for (int w = 0; w < VN; w++) {
                    pv.v[w] = phase;
                    step(tv.v[w]);
                }
                if (STORE) *reinterpret_cast<Vec16<T> *>(op + v * OSTR) = pv;
            }
        }
        tp += nv * ROW;
        if (STORE) op += nv * OSTR;
        for (int t = 0; t < (int)cnt; t++) {               // trailing partial vector
            if (STORE) op[t] = phase;
            step(tp[t]);
        }
        i = seg_end;
    }
}

// Data-derived starting guess for a warm-up that begins at sample ws (any guess is legal --
// exactness comes from the seam check -- but a PM signal has a second stable lock point pi
// away from the carrier, and Doppler moves the carrier far from its value at lock, so the
// guess must land in the right basin): frequency from the lag-L autocorrelation angle,
// phase from the coherent sum of the de-rotated samples (the +-m modulation averages to
// cos m > 0 along the carrier).
template <typename T, int FMT>
__device__ __forceinline__ void pll_guess_fmt(IqSrc pcm, long long ws, long long n, int lag, T fallback_freq,
                                              T &phase, T &freq)
{
#ifndef PDT_GUESS_K
#define PDT_GUESS_K 512    // samples of the autocorrelation behind the frequency guess (1024 / 512 / 256: phase kernel 1.15 / 1.12 / 1.10 ms, same seam statistics)
#endif
    const int K = PDT_GUESS_K, KP = 96;
    float rr = 0, ri = 0;
    long long cnt = 0;
    if (ws + K + lag <= n) {
        // 32 samples and their lagged partners per trip, 16-byte loads, all of them in flight together: every trip exposes
        // one memory latency (the sums keep their sample order, so the guess does not depend on the batch length)
        constexpr int GB = 32;
        for (int k0 = 0; k0 < K; k0 += GB) {
            float a0[GB], b0[GB], a1[GB], b1[GB];
            iq_block<float, FMT, GB>(pcm.p, ws + k0, a0, b0);
            iq_block<float, FMT, GB>(pcm.p, ws + k0 + lag, a1, b1);
#pragma unroll
            for (int u = 0; u < GB; u++) {
                rr += a1[u] * a0[u] + b1[u] * b0[u];      // x1 * conj(x0)
                ri += b1[u] * a0[u] - a1[u] * b0[u];
            }
        }
        cnt = K;
    }
    float f = (float)fallback_freq;
    if (cnt >= 64 && (rr != 0 || ri != 0)) f = atan2f(ri, rr) / (float)lag;
    float sr = 0, si = 0;
    if (ws + KP <= n) {
        constexpr int PB = 32;
        for (int k0 = 0; k0 < KP; k0 += PB) {
            float a0[PB], b0[PB];
            iq_block<float, FMT, PB>(pcm.p, ws + k0, a0, b0);
#pragma unroll
            for (int u = 0; u < PB; u++) {
                float sn, cs;
                __sincosf(f * (float)(k0 + u), &sn, &cs);
                sr += a0[u] * cs + b0[u] * sn;            // x * e^{-j f k}
                si += b0[u] * cs - a0[u] * sn;
            }
        }
    }
    float ph = (sr != 0 || si != 0) ? atan2f(si, sr) : 0.0f;
    // representative used by the reference's wrap logic: (0, 2pi] for f >= 0, [-2pi, 0) otherwise
    if (f >= 0 && ph < 0) ph += 6.28318530718f;
    if (f < 0 && ph > 0) ph -= 6.28318530718f;
    phase = (T)ph;
    freq = (T)f;
}

template <typename T>
__device__ __forceinline__ void pll_guess(IqSrc pcm, long long ws, long long n, int lag, T fallback_freq, T &phase, T &freq)
{
    if (pcm.fmt == 0) pll_guess_fmt<T, 0>(pcm, ws, n, lag, fallback_freq, phase, freq);
    else pll_guess_fmt<T, 1>(pcm, ws, n, lag, fallback_freq, phase, freq);
}

// Blocks are aligned to absolute multiples of B: block j covers [j*B, min(n, (j+1)*B)).
// Warm-up = [wide-band stage, Wacq/4 samples] + [acquisition-gain stage, Wacq] + [tracking-gain
// stage, Wtrk].  The loop gains of the first two stages are free choices (they only steer the
// guess); the tracking stage runs the reference's own gains so that the state merges with the
// true trajectory.
//
// The kernel does not need to know where the one-time lock happened: every block starts from a
// data-derived guess, so it runs CONCURRENTLY with the sequential acquisition kernel (second
// stream).  k_pll_head then walks from the lock sample to the end of its block with the true
// state, and k_pll_fix validates every later seam against that chain; phases computed for
// samples before the lock are simply never used.
// What k_pll_phase tells k_pll_head (which runs beside it and walks a few blocks further while that costs nothing): HINTS only.
struct PllPhaseHint {
    unsigned done;              // += 1 per finished workgroup
    unsigned pad_;
    unsigned long long t0;      // the constant 100 MHz clock when its first workgroup began (0 = not yet)
};
__device__ __forceinline__ unsigned long long pdt_wall_clock() { return wall_clock64(); }

template <typename T, bool SLOW>
__device__ __forceinline__ void k_pll_phase(IqSrc pcm, const T *__restrict__ theta, long long n,
                                                   PllParams<T> P, long long B, long long Wacq, long long Wtrk, int lag,
                                                   T *__restrict__ phi, PllSeam<T> *__restrict__ seams,
                                                   PllPhaseHint *__restrict__ hint /* k_pll_head's */,
                                                   int short_group = -1 /* the workgroup that walks the blocks in front of Wtrk, or -1 */,
                                                   T *__restrict__ ckpt = nullptr /* frequency checkpoints (pll_phase_range), or none */,
                                                   T consensus = 0 /* > 0: a walker whose frequency behind the wide-band stage is further than this
                                                                      from its wavefront's median takes the median */)
{
    // A SIMD of its own for every walker wavefront (the whole register file claimed, as k_pll_acquire_pipe and k_pll_head do):
    // the acquisition's two wavefronts are placed first and take two SIMDs of a CU; a workgroup of this kernel that lands on
    // the same CU would seat its four wavefronts on the two SIMDs left -- two walkers per SIMD, half the pace for as long as the
    // acquisition runs, and the kernel ends with its slowest wavefront (round 4, kernel trace: 5.8 ms in the chain against
    // 5.2 ms for the same body alone, tools/probes/pll_mem_probe.hip).  With the claim such a workgroup needs four empty SIMDs.
    asm volatile("" ::: "v255", "a255");
    // (`Done` counts a workgroup when its thread 0 leaves, not when its last walker does: the counter is a HINT for k_pll_head,
    // which walks a few blocks further while it says this kernel is still at work.  How many blocks the head takes over -- and
    // with it pll_seam_fixes and the two kernels' times -- therefore depends on timing; the result never does: every seam
    // behind the head is validated by k_pll_fix whoever walked it.)
    struct Done {                                   // (counted on every way out of the kernel)
        unsigned *p;
        __device__ ~Done() { if (threadIdx.x == 0) atomicAdd(p, 1u); }
    } done_{&hint->done};
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&hint->t0, pdt_wall_clock(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // The blocks that begin less than Wtrk into the capture have a warm-up that starts at sample 0: their walkers cross the block
    // boundaries -- where pll_phase_range starts a new segment -- at other steps than everybody else's, and a wavefront that
    // holds both kinds runs every segment for the longer of the two: an hour at 250 ksps, Wtrk = 5 B + 2 192: wavefront 0 walked
    // 6 B + B steps where the others walk Wtrk + B, and the kernel ended 0.7 ms after everybody else with it (round 4,
    // tools/probes/pll_mem_probe.hip real).  Those few walkers get a wavefront of their own (workgroup `short_group`, one lane
    // each: all of them aligned at sample 0, the longest walks ceil(W / B) B <= W + B steps); their lanes of wavefront 0
    // stay idle.  (W = the three stages together, since it is the tracking stage that a walker near the beginning shortens: below.)
    const long long Wwide = (Wacq / 4 + 3) & ~3ll;
    long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (short_group >= 0) {
        const long long n_short = (Wwide + Wacq + Wtrk + B - 1) / B;   // (the launch makes sure that these fit into one wavefront)
        if ((int)blockIdx.x == short_group) {
            if ((long long)threadIdx.x >= n_short) return;
            j = threadIdx.x;
        } else if (j < n_short) {
            return;
        }
    }
    const long long start = j * B;
    if (start >= n) return;
    const long long end = (start + B < n) ? start + B : n;
    long long w_wide = Wwide, w_acq = Wacq, w_trk = Wtrk;
    if (start < w_wide + w_acq + w_trk) {
        // near the beginning of the capture the TRACKING stage is what gets shorter: a walker that goes from its guess straight
        // into the narrow loop (round 3 shortened the wide-band and acquisition stages first) can sit at the frequency limit
        // for good -- the seam then costs two whole block walks, and such a block lies right behind k_pll_head's stretch
        // (c3 with 17 472-sample blocks: block 6, dfreq 0.108)
        w_wide = (start < w_wide) ? (start & ~3ll) : w_wide;
        w_acq = (start - w_wide < w_acq) ? ((start - w_wide) & ~3ll) : w_acq;
        w_trk = start - w_wide - w_acq;
    }
    const long long ws = start - w_trk - w_acq - w_wide;
    T phase, freq;
    pll_guess(pcm, ws, n, lag, (T)0, phase, freq);
    if (freq > P.max_freq) freq = P.max_freq;
    if (freq < P.min_freq) freq = P.min_freq;
    // the look-ahead ring of this wavefront (pll_phase_range): the tracking warm-up and the block itself go through it
    __shared__ __attribute__((aligned(16))) unsigned char ring_all[4 * PDT_PLL_RING_PF * PDT_RING_SLOT];
    unsigned char *ring = ring_all + (threadIdx.x >> 6) * (PDT_PLL_RING_PF * PDT_RING_SLOT);
    pll_phase_range<T, false, SLOW, true>(theta, phi, B, ws, ws + w_wide, phase, freq, P.alpha_wide, P.beta_wide, P.max_freq, P.min_freq);
    // Consensus (round 4).  On a weak signal one walker in a few hundred leaves the wide-band stage so far from the carrier that
    // the acquisition-gain stage runs it into the frequency limit, where the narrow tracking loop never finds back: its block
    // and the next one are then walked again by k_pll_fix (10 minutes at 250 ksps with six times the noise: 13 such walkers of
    // 11 269 were 26 repairs and 2.6 of the step's 10.9 ms).  The 64 blocks of a wavefront are seconds apart -- the carrier moves
    // by tens of Hz per second at most -- so an outlier takes the wavefront's median frequency (any start state is legal: the
    // seams are validated all the same).  tools/probes/pll_mem_probe.hip lost: 22 lost walkers -> 0.
    if (consensus > (T)0) {
        const bool part = w_wide == Wwide && Wwide > 0;                    // ran the whole stage
        const unsigned long long pm = __ballot(part);
        const int np = __popcll(pm);
        if (np >= 8) {
            T med = freq;
            for (int k = 0; k < 64; k++) {
                if (!((pm >> k) & 1ull)) continue;                         // (uniform)
                const T c = __shfl(freq, k);
                const int lt = __popcll(__ballot(part && freq < c)), le = __popcll(__ballot(part && freq <= c));
                if (lt <= (np - 1) / 2 && le > (np - 1) / 2) med = c;
            }
            if (part && Real<T>::abs(freq - med) > consensus) freq = med;
        }
    }
    // acquisition-gain stage; its last 128 samples vote on which of the two stable lock points we
    // fell into: at the carrier the detector error sits at +-m (|err| < pi/2), at the false point
    // pi away it sits at +-(pi - m) (|err| > pi/2)
    const long long a0 = ws + w_wide, a1 = a0 + w_acq;
    const long long vote0 = (w_acq > 160) ? a1 - 128 : a1;
    pll_phase_range<T, false, SLOW, true>(theta, phi, B, a0, vote0, phase, freq, P.alpha_acq, P.beta_acq, P.max_freq, P.min_freq);
    int far = 0, seen = 0;
    pll_phase_range<T, false, SLOW, true, PDT_PLL_PF, true>(theta, phi, B, vote0, a1, phase, freq, P.alpha_acq, P.beta_acq, P.max_freq,
                                                        P.min_freq, &far, &seen);
    if (2 * far > seen) {
        phase = (phase > 0) ? phase - (T)PDT_PI : phase + (T)PDT_PI;     // stays inside (-2pi, 2pi)
        if (freq >= 0 && phase < 0) phase += (T)(2 * PDT_PI);
        if (freq < 0 && phase > 0) phase -= (T)(2 * PDT_PI);
    }
    pll_phase_range<T, false, SLOW, true>(theta, phi, B, a1, start, phase, freq, P.alpha_trk, P.beta_trk, P.max_freq, P.min_freq,
                                          nullptr, nullptr, ring);
    PllSeam<T> sm;
    sm.phase0 = phase;
    sm.freq0 = freq;
    pll_phase_range<T, true, SLOW, true>(theta, phi, B, start, end, phase, freq, P.alpha_trk, P.beta_trk, P.max_freq, P.min_freq,
                                         nullptr, nullptr, ring, ckpt ? ckpt + ((j >> 6) * pll_ckpt_count(B)) * 64 + (j & 63) : nullptr);
    sm.phase1 = phase;
    sm.freq1 = freq;
    seams[j] = sm;
#ifdef PDT_FIX_TRACE
    if ((threadIdx.x & 63) == 0 || j < 8)
        printf("k_pll_phase: group %d wavefront %d lane %d block %lld: stages %lld %lld %lld + %lld, %.3f ms\n", (int)blockIdx.x, (int)threadIdx.x >> 6,
               (int)threadIdx.x & 63, j, w_wide, w_acq, w_trk, end - start, (double)(pdt_wall_clock() - __hip_atomic_load(&hint->t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) * 1e-5);
#endif
}

// The first samples after the lock cannot be block-parallel: the true state starts from the
// acquisition's state and needs ~W samples to merge with anything a warm-up can reach.  One wavefront
// walks them -- from the sample after the lock to the end of the first block that begins W samples
// later -- with the TRUE state (one lane; a speculative straight-line variant that skipped the wrap
// logic was measured slower than the plain select-based step).  It runs beside k_pll_phase (it depends only on the acquisition),
// so it writes the phases and the per-block seam records to side buffers; k_pll_fix grafts them over
// the block-parallel results before it validates the later seams.
template <typename T> struct PllHeadInfo { long long s0, s1, j0, nblk; };

template <typename T, bool SLOW, bool EXCL = false>
__device__ __forceinline__ void k_pll_head(const T *__restrict__ theta, long long n, PllParams<T> P,
                                                  const PllLockInfo<T> *__restrict__ info, long long B, long long W,
                                                  T *__restrict__ phi_head, PllSeam<T> *__restrict__ seams_head,
                                                  PllHeadInfo<T> *__restrict__ hinfo, long long max_blocks,
                                                  long long W_max /* >= W: walk on up to there while ... */,
                                                  const PllPhaseHint *__restrict__ hint /* ... fewer than */,
                                                  unsigned phase_groups /* workgroups of k_pll_phase have finished, and */,
                                                  long long phase_steps /* its walkers' steps say that it will run long enough */)
{
    if (EXCL) asm volatile("" ::: "v255", "a255");       // a SIMD of its own, see k_pll_acquire_pipe
    __shared__ __attribute__((aligned(16))) unsigned char ring[PDT_PLL_RING_PF * PDT_RING_SLOT];
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const long long lock_at = info->lock_sample;
    PllHeadInfo<T> hi;
    hi.s0 = hi.s1 = 0; hi.j0 = 0; hi.nblk = 0;
    const long long S = lock_at + 1;
    if (lock_at >= 0 && S < n) {
        const long long j0 = S / B;
        long long j1 = (S + W + B - 1) / B;                 // first block that starts >= W after the lock
        if (j1 - j0 + 1 > max_blocks) j1 = j0 + max_blocks - 1;
        long long j2 = (S + W_max + B - 1) / B;             // ... and the last one worth walking while the kernel beside us runs
        if (j2 - j0 + 1 > max_blocks) j2 = j0 + max_blocks - 1;
        if (j2 < j1) j2 = j1;
        T phase = info->st.phase, freq = info->st.freq;
        long long pos = S, k = 0;
        const unsigned long long t_begin = pdt_wall_clock();
        for (long long j = j0; j <= j2 && pos < n; j++, k++) {
            const long long end = ((j + 1) * B < n) ? (j + 1) * B : n;
            // (beyond the mandatory stretch only while the block-parallel kernel is still at work: those blocks are free, and the
            // seams they take over are the ones most likely to be open -- and only if this block will be done before that kernel
            // is: its walkers step at this walker's pace (a lone wavefront on its SIMD, the same loop), so the clock at which
            // they began + phase_steps of the steps timed here is when they end; a block of an hour at 250 ksps is 1 ms, and
            // before this test the head ended up to a block after the kernel it was keeping company: 6.05 against 5.62 ms.)
            if (j > j1) {
                if (__hip_atomic_load(&hint->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= phase_groups) break;
                const unsigned long long t0 = __hip_atomic_load(&hint->t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), now = pdt_wall_clock();
                if (t0 != 0 && now > t0 && pos > S) {
                    const float per_step = (float)(now - t_begin) / (float)(pos - S);
                    // (the walkers beside us share HBM with 700 others: ~5 % slower than this one; and the seam right behind the
                    // mandatory stretch is the one that stays open now and then -- two block walks of repair -- so ending up to
                    // a third of a block after them is still the better bet)
                    if ((float)(now - t0) + per_step * (float)(end - pos) > per_step * (1.05f * (float)phase_steps + 0.3f * (float)B)) break;
                }
            }
            PllSeam<T> sm;
            sm.phase0 = phase;
            sm.freq0 = freq;
            // phi_head is indexed from the 16-byte aligned sample at or below S (vector stores stay aligned)
            pll_phase_range<T, true, SLOW, false, 32>(theta, phi_head - (S & ~3ll), B, pos, end, phase, freq, P.alpha_trk, P.beta_trk,
                                                      P.max_freq, P.min_freq, nullptr, nullptr, ring);
            sm.phase1 = phase;
            sm.freq1 = freq;
            seams_head[k] = sm;
            pos = end;
        }
        hi.s0 = S; hi.s1 = pos; hi.j0 = j0; hi.nblk = k;
    }
    *hinfo = hi;
}

template <typename T> __device__ __forceinline__ bool bits_equal(T x, T y);
template <> __device__ __forceinline__ bool bits_equal<float>(float x, float y)
{
    return __float_as_uint(x) == __float_as_uint(y);
}
template <> __device__ __forceinline__ bool bits_equal<double>(double x, double y)
{
    return __double_as_longlong(x) == __double_as_longlong(y);
}

// Seam validation + repair.  A seam is healthy when the previous block's end state equals, bit for bit,
// the state this block reached at its official start after its warm-up.  One workgroup of 16 wavefronts
// works in rounds: all seams not yet final are checked in parallel; the first (up to 16) unhealthy blocks
// are re-run at the same time, one wavefront each, from their predecessors' end states INTO A SCRATCH
// buffer; the re-runs are then committed in order for as long as each leaves its block's end state
// unchanged (the usual case: the re-run merges with the old trajectory inside the block), because only
// then is the next one's starting state known to be final.  The first re-run that changes its end
// state is committed too -- everything before it is final -- and the following seam is examined again
// in the next round.  So isolated unhealthy seams cost one block walk in total, a cascade costs one
// walk per block as a sequential pass would, and a healthy block is never overwritten from a state
// that is not known to be the true one.
#define PDT_FIX_THREADS 1024
#define PDT_FIX_LIST 64

// One lane walks block [start, end) again from (phase, freq), piece by piece of PDT_PLL_CKPT samples.  In front of every piece
// that begins at a checkpoint it compares its state with the stored trajectory's -- the phase stream holds the phase in front of
// that sample, ck_old the frequency -- and stops when the two are equal bit for bit: from there on the stored samples ARE what
// it would compute (round 4: an isolated open seam merges within a few loop time constants; the re-run used to walk the whole
// block, 1 ms for an hour at 250 ksps, and the seam behind k_pll_head's stretch stays open now and then).  The frequencies it
// passes go to ck_new (column stride ck_new_stride: the scratch list of a re-run, or ck_old itself when it writes in place).
// Returns the number of samples walked; `merged` says that the block's recorded end state still holds.
template <typename T, bool SLOW, bool OUT_LT>
__device__ __forceinline__ long long pll_rewalk(const T *__restrict__ theta, T *out, long long B, long long start, long long end, T &phase,
                                                T &freq, const PllParams<T> &P, const T *phi_old /* LT */, T *ck_old /* column, stride 64, or null */,
                                                T *ck_new, long long ck_new_stride, bool &merged, unsigned char *ring = nullptr)
{
    const long long len = end - start;
    merged = false;
    long long p = 0;
    while (p < len) {
        const long long c = p / PDT_PLL_CKPT;
        if (ck_old && pll_ckpt_valid<T>(c, len)) {
            const T f_old = ck_old[c * 64];
            const T p_old = phi_old[Lt<T>::index(start + p, B)];
            if (bits_equal(p_old, phase) && bits_equal(f_old, freq)) { merged = true; break; }
            ck_new[c * ck_new_stride] = freq;
        }
        const long long q = (p + PDT_PLL_CKPT < len) ? p + PDT_PLL_CKPT : len;
        // (`ring`: the hand-issued LDS look-ahead of the block walkers -- 36 - 39 ns a sample against 58 for the register ring the
        // compiler schedules; the cascade of k_pll_fix, which walks whole runs of blocks, passes one)
        pll_phase_range<T, true, SLOW, OUT_LT, 32>(theta, out, B, start + p, start + q, phase, freq, P.alpha_trk, P.beta_trk, P.max_freq, P.min_freq,
                                                   nullptr, nullptr, ring);
        p = q;
    }
    return p;
}
template <typename T, bool SLOW>
__device__ __forceinline__ void k_pll_fix(const T *__restrict__ theta, long long n, PllParams<T> P,
                                                 const PllLockInfo<T> *__restrict__ info, long long B, T *__restrict__ phi,
                                                 PllSeam<T> *__restrict__ seams, const T *__restrict__ phi_head,
                                                 const PllSeam<T> *__restrict__ seams_head,
                                                 const PllHeadInfo<T> *__restrict__ hinfo, T *__restrict__ scratch,
                                                 unsigned *__restrict__ counters /* [0]=blocks [1]=fixes */, int mode,
                                                 long long region_blocks, long long region_offset,
                                                 T *ckpt = nullptr /* k_pll_phase's frequency checkpoints, or none */)
{
    // mode 0: graft the head's phases and seam records over the block-parallel ones, nothing else (one workgroup);
    // mode 1: region pass -- workgroup g validates and repairs the seams of its own region of `region_blocks` blocks, taking the
    //         end state of the block before its region as true.  Optimistic: a repair in the region to the left that changes that
    //         state invalidates the assumption, which a later pass (other region boundaries, or the final pass) finds, because a
    //         block re-run from state X records X as its start state and the seam check compares exactly that.  On a healthy
    //         capture every region just finds its seams closed; on a weak signal, where seams fail in runs that must be walked
    //         one block after the other, the runs of all regions are walked side by side;
    // mode 2: final pass over all seams (one workgroup): whatever is still open is repaired in order, as before.
    constexpr int NW = PDT_FIX_THREADS / 64;
    __shared__ long long s_bad[PDT_FIX_LIST];
    __shared__ T s_end[NW][2], s_beg[NW][2];
    __shared__ long long s_len[NW];                                 // samples a re-run walked before it merged (or the block's)
    __shared__ unsigned s_nbad;
    __shared__ long long s_min;
    __shared__ __attribute__((aligned(16))) unsigned char cascade_ring[PDT_PLL_RING_PF * PDT_RING_SLOT];      // (thread 0's walks, below)
    const long long lock_at = info->lock_sample;
    if (lock_at < 0) {
        if (threadIdx.x == 0 && mode == 2) counters[0] = 0;
        return;
    }
    const long long S = lock_at + 1;
    const PllHeadInfo<T> hi = *hinfo;
    const long long j0 = hi.j0;                                     // block that contains the lock
    const long long nb_abs = (n + B - 1) / B;                       // absolute block count
    const long long NC = pll_ckpt_count(B);
    const long long BS = ((B + 63) & ~63ll) + ((NC + 63) & ~63ll);  // scratch stride per wavefront: the block, then its checkpoints
    auto ck_col = [&](long long blk) { return ckpt ? ckpt + ((blk >> 6) * NC) * 64 + (blk & 63) : (T *)nullptr; };
    if (mode == 0) {
        // (a few workgroups: the head of an hour at 250 ksps is 100 000 samples, 0.2 ms for a single one)
        const long long t0 = (long long)blockIdx.x * PDT_FIX_THREADS + threadIdx.x, tn = (long long)gridDim.x * PDT_FIX_THREADS;
        for (long long i = hi.s0 + t0; i < hi.s1; i += tn) phi[Lt<T>::index(i, B)] = phi_head[i - (hi.s0 & ~3ll)];
        for (long long k = t0; k < hi.nblk; k += tn) seams[j0 + k] = seams_head[k];
        return;
    }
    unsigned fixes = 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long from = (hi.nblk > 0) ? j0 + hi.nblk : nb_abs;         // seams before `from` are final
    long long r_hi = nb_abs;                                        // seams [from, r_hi) are this workgroup's
    if (mode == 1) {
        const long long r_lo = (long long)blockIdx.x * region_blocks - region_offset;
        r_hi = (r_lo + region_blocks < nb_abs) ? r_lo + region_blocks : nb_abs;
        if (from < r_lo) from = r_lo;
        if (from < 1) from = 1;
        scratch += (long long)blockIdx.x * NW * BS;
    }
    for (;;) {
        if (threadIdx.x == 0) { s_nbad = 0; s_min = r_hi; }
        __syncthreads();
        for (long long r = from + threadIdx.x; r < r_hi; r += PDT_FIX_THREADS) {
            const PllSeam<T> prev = seams[r - 1];
            const PllSeam<T> cur = seams[r];
            if (!(bits_equal(prev.phase1, cur.phase0) && bits_equal(prev.freq1, cur.freq0))) {
                const unsigned slot = atomicAdd(&s_nbad, 1u);
                if (slot < PDT_FIX_LIST) s_bad[slot] = r;
                atomicMin((unsigned long long *)&s_min, (unsigned long long)r);
            }
        }
        __syncthreads();
        unsigned nbad = s_nbad;
        if (nbad == 0) break;
        if (nbad > PDT_FIX_LIST) {
            // too many to order: take only the first one this round
            nbad = 1;
            __syncthreads();
            if (threadIdx.x == 0) s_bad[0] = s_min;
            __syncthreads();
        } else {
            // ascending order (tiny list: rank sort by the first threads)
            long long mine = 0;
            unsigned rank = 0;
            if (threadIdx.x < nbad) {
                mine = s_bad[threadIdx.x];
                for (unsigned q = 0; q < nbad; q++) rank += (s_bad[q] < mine) ? 1u : 0u;
            }
            __syncthreads();
            if (threadIdx.x < nbad) s_bad[rank] = mine;
            __syncthreads();
            if (nbad > (unsigned)NW) nbad = NW;
        }
        // re-run into the scratch buffer
        if ((unsigned)wave < nbad && lane == 0) {
            const long long rb = s_bad[wave];
            const PllSeam<T> prev = seams[rb - 1];
            T phase = prev.phase1, freq = prev.freq1;
            s_beg[wave][0] = phase;                 // the state this re-run really started from: that, and not a later reading
            s_beg[wave][1] = freq;                  // of the neighbour's record (another region may be rewriting it), is recorded
            const long long start = rb * B;
            const long long end = ((rb + 1) * B < n) ? (rb + 1) * B : n;
            // scratch index = sample index - (start rounded down to 4): vector stores stay aligned
            bool merged;
            T *sc = scratch + (long long)wave * BS;
            s_len[wave] = pll_rewalk<T, SLOW, false>(theta, sc - (start & ~3ll), B, start, end, phase, freq, P, phi, ck_col(rb),
                                                     sc + ((B + 63) & ~63ll), 1, merged);
#ifdef PDT_FIX_TRACE
            printf("pll fix mode %d: seam %lld of %lld (head blocks %lld..%lld), walked %lld of %lld, merged %d; dphase %g dfreq %g\n", mode, rb, nb_abs, j0,
                   j0 + hi.nblk - 1, s_len[wave], end - start, (int)merged, (double)(prev.phase1 - seams[rb].phase0), (double)(prev.freq1 - seams[rb].freq0));
#endif
            const PllSeam<T> old = seams[rb];
            s_end[wave][0] = merged ? old.phase1 : phase;       // merged: the rest of the block, its end state included, stands
            s_end[wave][1] = merged ? old.freq1 : freq;
        }
        __threadfence();
        __syncthreads();
        // commit in order while the end states stand
        unsigned ncommit = 0;
        for (unsigned q = 0; q < nbad; q++) {
            ncommit = q + 1;
            const PllSeam<T> old = seams[s_bad[q]];
            if (!(bits_equal(old.phase1, s_end[q][0]) && bits_equal(old.freq1, s_end[q][1]))) break;
        }
        __syncthreads();
        for (unsigned q = 0; q < ncommit; q++) {
            const long long rb = s_bad[q];
            const long long start = rb * B;
            const long long end = ((rb + 1) * B < n) ? (rb + 1) * B : n;
            const T *src = scratch + (long long)q * BS - (start & ~3ll);
            const long long walked = s_len[q];
            for (long long i = start + threadIdx.x; i < start + walked; i += PDT_FIX_THREADS) phi[Lt<T>::index(i, B)] = src[i];
            if (ckpt) {                                             // the frequencies the re-run passed (it compared before it wrote)
                const T *cks = scratch + (long long)q * BS + ((B + 63) & ~63ll);
                T *col = ck_col(rb);
                for (long long c = 1 + threadIdx.x; c * PDT_PLL_CKPT < walked; c += PDT_FIX_THREADS)
                    if (pll_ckpt_valid<T>(c, end - start)) col[c * 64] = cks[c];
            }
            if (threadIdx.x == 0) {
                PllSeam<T> upd;
                upd.phase0 = s_beg[q][0];
                upd.freq0 = s_beg[q][1];
                upd.phase1 = s_end[q][0];
                upd.freq1 = s_end[q][1];
                seams[rb] = upd;
            }
        }
        fixes += ncommit;
        from = s_bad[ncommit - 1] + 1;
        __threadfence();
        __syncthreads();
        // cascade: when the last committed re-run changed its block's end state, the following blocks will fail
        // one after the other until the true trajectory merges with the stored one (on weak signals that can be
        // most of the capture); walk them right here, in place -- everything before them is final --
        // instead of paying a scan + synchronisation round for each
        {
            const long long rl = s_bad[ncommit - 1];
            if (threadIdx.x == 0) {
                long long r = rl + 1;
                unsigned extra = 0;
                while (r < r_hi) {
                    const PllSeam<T> prev = seams[r - 1];
                    const PllSeam<T> cur = seams[r];
                    if (bits_equal(prev.phase1, cur.phase0) && bits_equal(prev.freq1, cur.freq0)) break;
                    T phase = prev.phase1, freq = prev.freq1;
                    const long long start = r * B;
                    const long long end = ((r + 1) * B < n) ? (r + 1) * B : n;
                    bool merged;
                    T *col = ck_col(r);
                    (void)pll_rewalk<T, SLOW, true>(theta, phi, B, start, end, phase, freq, P, phi, col, col, 64, merged, cascade_ring);
                    PllSeam<T> upd;
                    upd.phase0 = prev.phase1;
                    upd.freq0 = prev.freq1;
                    upd.phase1 = merged ? cur.phase1 : phase;
                    upd.freq1 = merged ? cur.freq1 : freq;
                    seams[r] = upd;
                    extra++;
                    r++;
                }
                s_min = r;
                s_nbad = extra;
            }
            __threadfence();
            __syncthreads();
            from = s_min;
            fixes += s_nbad;
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        if (mode == 2) counters[0] = (unsigned)((S < n) ? nb_abs - j0 : 0);
        if (fixes) atomicAdd(&counters[1], fixes);
    }
}

// Tail pass (round 6): the stretches of a capture where the loop has nothing to track -- a receiver that stays on after the
// satellite has set, a fade behind a building.  On noise the tracking loop does not contract: no warm-up merges with
// the truth there, every seam of the stretch is open, and the true trajectory has to be WALKED through it, one lane, ~40 ns a
// sample -- a minute at 250 ksps is 0.6 s.  Left to k_pll_fix that walk starts when the acquisition and the head are through; but
// it needs neither: the blocks in front of a stretch merged with the truth as blocks do wherever there is a signal, so the end
// state of the last block of a run of `run` closed seams is -- almost certainly -- the true state there.  This kernel runs on the
// side stream right behind k_pll_phase, BESIDE the acquisition (which spends its own second on the noise in front of the pass):
// every open seam that follows `run` closed ones starts a stretch; workgroup g takes the g-th stretch and walks it from the end
// state of the block in front of it until it has passed `run` closed seams in a row (the signal is back:
// the stored trajectory is the walker's from there on) or the end of the capture -- in place, every block's seam record rewritten
// with the state the walk really started from and ended in, as k_pll_fix's cascade does; blocks whose seam is closed stand as
// they are.  Walks never meet: a walk ends behind `run` closed seams, which is where the next stretch begins.  Nothing is taken on trust: k_pll_fix validates every seam afterwards
// as ever, and a start state that a later repair changes fails its seam check and is walked again.  (The noise in FRONT of the
// pass, and a capture that is noise throughout, have no closed run in front of them: nothing happens there.)
// ... the stretches are listed first, by a launch of its own (one workgroup: k_pll_tail_scan), and walked by the next (a workgroup
// per stretch): a walker rewrites seam records, and a search running beside it could take a half-written one for a closed seam.
#define PDT_TAIL_MAX 1024   // (a weak stretch -- the first and last minute of a pass -- has isolated open seams by the hundred: each its own walker, side by side)
#define PDT_TAIL_LDS_CLAIM 147456   // of a CU's 163 840 bytes
struct PllTailList {
    unsigned n;                       // stretches found (more than PDT_TAIL_MAX: the rest is k_pll_fix's)
    unsigned pad_;
    long long start[PDT_TAIL_MAX];    // first block of each
};
template <typename T>
__device__ __forceinline__ void k_pll_tail_scan(long long n, long long B, const PllSeam<T> *__restrict__ seams, PllTailList *__restrict__ list,
                                                int run)
{
    __shared__ unsigned s_n;
    const long long nb = (n + B - 1) / B;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    auto closed = [&](long long r) {
        const PllSeam<T> prev = seams[r - 1];
        const PllSeam<T> cur = seams[r];
        return bits_equal(prev.phase1, cur.phase0) && bits_equal(prev.freq1, cur.freq0);
    };
    for (long long r = run + 1 + (long long)threadIdx.x; r < nb; r += blockDim.x) {
        if (closed(r)) continue;
        bool all = true;
        for (int q = 1; q <= run && all; q++) all = closed(r - q);
        if (all) {
            const unsigned slot = atomicAdd(&s_n, 1u);
            if (slot < PDT_TAIL_MAX) list->start[slot] = r;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) list->n = s_n < (unsigned)PDT_TAIL_MAX ? s_n : (unsigned)PDT_TAIL_MAX;
}
template <typename T, bool SLOW>
__device__ __forceinline__ void k_pll_tail(const T *__restrict__ theta, long long n, PllParams<T> P, long long B, T *__restrict__ phi,
                                           PllSeam<T> *__restrict__ seams, unsigned *__restrict__ counters, T *ckpt,
                                           const PllTailList *__restrict__ list, int run)
{
    // A walker is one lone wavefront for half a second: it wants a SIMD's issue slots to itself.  (i) The ring is declared at
    // PDT_TAIL_LDS_CLAIM bytes, more than half a CU's LDS, so that no two walkers are ever placed on one CU (the acquisition's two
    // wavefronts claim their SIMDs' register files, EXCL).  (ii) In a batch (blockIdx.z = the capture) the g-th stretch of every
    // capture would sit in the same workgroup index, which the dispatcher maps to ONE shader engine of ONE XCD (index mod 8, then
    // (index / 8) mod 4): sixteen passes' noise tails on eight CUs, two to a SIMD or waiting for a CU (measured: 1.1 s instead of
    // 0.5).  The index is rotated by 41 per capture -- the next XCD and the next shader engine.
    __shared__ __attribute__((aligned(16))) unsigned char ring[PDT_TAIL_LDS_CLAIM];
    static_assert(PDT_TAIL_LDS_CLAIM >= PDT_PLL_RING_PF * PDT_RING_SLOT, "the claim holds the ring");
    const unsigned g = ((unsigned)blockIdx.x + 41u * (unsigned)blockIdx.z) % (unsigned)PDT_TAIL_MAX;
    if (threadIdx.x != 0 || g >= list->n) return;
    const long long nb = (n + B - 1) / B;
    const long long NC = pll_ckpt_count(B);
    unsigned walked = 0;
    int closed_run = 0;
    for (long long r = list->start[g]; r < nb; r++) {
        const PllSeam<T> prev = seams[r - 1];
        const PllSeam<T> cur = seams[r];
        if (bits_equal(prev.phase1, cur.phase0) && bits_equal(prev.freq1, cur.freq0)) {
            // merged with the stored trajectory: this block stands as it is (the same start state gives the same samples).  A weak
            // stretch -- a fade that leaves a little of the signal -- is patchy: a few blocks merge, the next seam is open again; the
            // walker hops over the blocks that stand and goes on, and ends behind `run` closed seams in a row (where the search
            // would begin the next stretch)
            if (++closed_run >= run) break;
            continue;
        }
        closed_run = 0;
        T phase = prev.phase1, freq = prev.freq1;
        const long long start = r * B;
        const long long end = ((r + 1) * B < n) ? (r + 1) * B : n;
        T *col = ckpt ? ckpt + ((r >> 6) * NC) * 64 + (r & 63) : (T *)nullptr;
        pll_phase_range<T, true, SLOW, true, 32>(theta, phi, B, start, end, phase, freq, P.alpha_trk, P.beta_trk, P.max_freq, P.min_freq,
                                                 nullptr, nullptr, ring, col);
        PllSeam<T> upd;
        upd.phase0 = prev.phase1;
        upd.freq0 = prev.freq1;
        upd.phase1 = phase;
        upd.freq1 = freq;
        seams[r] = upd;
        walked++;
    }
    if (walked) atomicAdd(&counters[1], walked);
}

// elementwise mix for the samples after the lock (:106-113), and the lock-detector input
// term lockSigAlpha*(re*t_real + im*t_imag) (:194-220) when the lock stream is wanted.  The phases arrive in the LT
// layout: a workgroup takes RG rows of one tile (coalesced), transposes them through LDS and then works along the 64
// lane-rows in natural order -- I/Q in, mixed samples out, both in runs of RG * 16 bytes.
// AVGTERM: instead of the mixed samples, only the input term of the averagePhase EMA, averagePhaseAlpha * |arctan2(out)|
// (:117-124, :145-152), goes to lock_term -- the quality figure's stream after the lock (pdt_keep_quality).
template <typename T, bool LOCKSIG, bool AVGTERM = false>
__device__ __forceinline__ void k_pll_mix(IqSrc pcm, const T *__restrict__ phi_lt, long long n, long long B,
                                                  PllParams<T> P, const PllLockInfo<T> *__restrict__ info,
                                                  T *__restrict__ out, T *__restrict__ lock_term)
{
    constexpr int VN = Lt<T>::VN, ROW = Lt<T>::ROW, RG = Lt<T>::RG;
    __shared__ Vec16<T> s_v[64 * (RG + 1)];
    const long long lock_at = info->lock_sample;
    if (lock_at < 0) return;
    const long long S = lock_at + 1;
    const long long groups = B / (VN * RG);
    const long long w = (long long)blockIdx.x / groups, g = (long long)blockIdx.x - w * groups;
    if ((w << 6) * B >= n) return;
    if (((w << 6) + 63) * B + (g * RG + RG) * VN <= S) return;            // everything here precedes the lock
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
        const int qd = pass * 4 + ((int)threadIdx.x >> 6), l = (int)threadIdx.x & 63;
        s_v[l * (RG + 1) + qd] = *reinterpret_cast<const Vec16<T> *>(phi_lt + (w << 6) * B + (g * RG + qd) * ROW + l * VN);
    }
    __syncthreads();
    auto one = [&](T a, T b, T ph, T &o, T &lt) {
        T t_real, t_imag;
        Real<T>::sincos(ph, t_imag, t_real);
        const T c = t_real, d = -t_imag;
        o = a * d + b * c;
        if (AVGTERM) {
            const T o_re = a * c - b * d;
            lt = (T)0.00005 * Real<T>::abs(arctan2_ref(o, o_re));
        }
        if (LOCKSIG) {
            const T mag2 = a * a + b * b;
            const T inv = (T)q_rsqrt((float)mag2);
            const T re = a * inv, im = b * inv;
            lt = P.lock_alpha * (re * t_real + im * t_imag);
        }
    };
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
        const int l = pass * 16 + ((int)threadIdx.x >> 4), qd = (int)threadIdx.x & 15;
        const long long i = ((w << 6) + l) * B + (g * RG + qd) * VN;
        if (i >= n || i + VN <= S) continue;
        const Vec16<T> ph = s_v[l * (RG + 1) + qd];
        if (i >= S && i + VN <= n) {
            T a[VN], b[VN];
            iq_vec<T>(pcm, i, a, b);
            Vec16<T> o, lt;
#pragma unroll
            for (int e = 0; e < VN; e++) {
                lt.v[e] = 0;
                one(a[e], b[e], ph.v[e], o.v[e], lt.v[e]);
            }
            if (!AVGTERM) *reinterpret_cast<Vec16<T> *>(out + i) = o;       // i is a multiple of VN
            if (LOCKSIG || AVGTERM) *reinterpret_cast<Vec16<T> *>(lock_term + i) = lt;
        } else {
#pragma unroll
            for (int e = 0; e < VN; e++) {
                const long long k = i + e;
                if (k >= S && k < n) {
                    T a, b, o, lt = 0;
                    IqSample<T>::get(pcm, k, a, b);
                    one(a, b, ph.v[e], o, lt);
                    if (!AVGTERM) out[k] = o;
                    if (LOCKSIG || AVGTERM) lock_term[k] = lt;
                }
            }
        }
    }
}

// Lock-detector EMA over the precomputed input terms (ARGOS): L = L*(1-a) + u_i, evaluated
// in double and narrowed like the reference (:220).  Same block/warm-up/seam scheme.
template <typename T> struct EmaSeam { T v0, v1; };

// the EMA over [i0, i1).  STORE 0: nothing is written (warm-ups, zero responses); 1: every value (the lock stream Squelch
// reads); 2: only the value after the last sample of every chunk of `chunk` samples (the averagePhase the chunk loop prints,
// CarrierTrackingPLL.c:277: k_chunk_info reads nothing else of that stream) -- the full stream was 3.6 GB of 16-byte stores per
// lane at an hour of 250 ksps.  `ring` (PDT_EMA_PF KiB of LDS for this wavefront, or nullptr): the look-ahead as an LDS ring
// with hand-placed waits (ring_issue; the register ring's loop header waits for every load in flight, so each trip of 32
// vectors exposed a memory latency: ~35 ns a step where the arithmetic is 13).
#define PDT_EMA_PF 64
template <typename T, int STORE, int PF = 32>
__device__ __forceinline__ void ema_range(const T *__restrict__ term, T *__restrict__ out, long long i0, long long i1, T &L, double k,
                                          unsigned char *ring = nullptr, long long chunk = 0)
{
    constexpr int VN = Vec16<T>::N;
    long long i = i0;
    // STORE 2: samples from i up to and including the next chunk end
    long long togo = 0;
    if (STORE == 2) togo = (i / chunk + 1) * chunk - i;
    auto one = [&](T x, long long at) {
        L = (T)((double)L * k + (double)x);
        if (STORE == 1) out[at] = L;
        if (STORE == 2) {
            if (--togo == 0) {
                out[at] = L;
                togo = chunk;
            }
        }
    };
    for (; i < i1 && (i % VN) != 0; i++) one(term[i], i);
    auto vec = [&](const Vec16<T> &x, long long at) {
        if (STORE == 2) {
            if (togo > VN) {                                   // no chunk ends inside this vector
#pragma unroll
                for (int w = 0; w < VN; w++) L = (T)((double)L * k + (double)x.v[w]);
                togo -= VN;
            } else {
#pragma unroll
                for (int w = 0; w < VN; w++) one(x.v[w], at + w);
            }
        } else {
            Vec16<T> yv;
#pragma unroll
            for (int w = 0; w < VN; w++) {
                L = (T)((double)L * k + (double)x.v[w]);
                yv.v[w] = L;
            }
            if (STORE == 1) *reinterpret_cast<Vec16<T> *>(out + at) = yv;
        }
    };
    if (ring && i + (long long)PDT_EMA_PF * VN <= i1) {
        constexpr int RPF = PDT_EMA_PF;
        const unsigned ring0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)ring);
        const unsigned char *mine = ring + 16 * (threadIdx.x & 63);
#pragma unroll
        for (int u = 0; u < RPF; u++) ring_issue(term + i + u * VN, ring0 + u * PDT_RING_SLOT);
        ring_wait<RPF - 1>();
        Vec16<T> cur = *reinterpret_cast<const Vec16<T> *>(mine), nxt;
        for (; i + RPF * VN <= i1; i += RPF * VN) {
#pragma unroll
            for (int u = 0; u < RPF; u++) {
                ring_wait<RPF - 2>();
                nxt = *reinterpret_cast<const Vec16<T> *>(mine + ((u + 1) % RPF) * PDT_RING_SLOT);
                vec(cur, i + u * VN);
                ring_issue(term + i + (RPF + u) * VN, ring0 + u * PDT_RING_SLOT);   // (up to RPF vectors past i1: slack of the buffers)
                cur = nxt;
            }
        }
        ring_wait<0>();
        // the ring holds the RPF vectors from i on: what is left of the range (fewer than RPF whole vectors) is among them
        for (int u = 0; i + VN <= i1; i += VN, u++) {
            if (u > 0) cur = *reinterpret_cast<const Vec16<T> *>(mine + u * PDT_RING_SLOT);
            vec(cur, i);
        }
    } else if (i + PF * VN <= i1) {
        Vec16<T> buf[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) buf[u] = *reinterpret_cast<const Vec16<T> *>(term + i + u * VN);
        for (; i + PF * VN <= i1; i += PF * VN) {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                vec(buf[u], i + u * VN);
                long long q = i + (PF + u) * VN;      // reload after the last use (see pll_phase_range)
                asm volatile("" : "+v"(q));
                buf[u] = *reinterpret_cast<const Vec16<T> *>(term + q);
            }
        }
    }
    for (; i < i1; i++) one(term[i], i);
}

// Response of every block to a zero start state: with it the state at every block boundary follows from the state at the lock
// by composing affine maps (k_lock_ema_guess) -- not bit for bit (the true recurrence rounds at every sample), but to a few
// tens of ulps, so that a walker started from it agrees with the truth after 8 time constants instead of 45.
template <typename T>
__device__ __forceinline__ void k_lock_ema_zero(const T *__restrict__ term, long long n, T lock_alpha,
                                                const PllLockInfo<T> *__restrict__ info, long long B, double *__restrict__ zresp)
{
    const long long lock_at = info->lock_sample;
    if (lock_at < 0) return;
    const long long S = lock_at + 1;
    const long long j = S / B + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long start = j * B;
    if (start >= n) return;
    if (start < S) start = S;
    const long long end = ((j + 1) * B < n) ? (j + 1) * B : n;
    T L = 0;
    __shared__ __attribute__((aligned(16))) unsigned char ring[PDT_EMA_PF * PDT_RING_SLOT];
    ema_range<T, 0>(term, (T *)nullptr, start, end, L, 1.0 - (double)lock_alpha, ring);
    zresp[j - S / B] = (double)L;
}

// The same zero responses, a WAVEFRONT per block (round 4; the averagePhase EMA's blocks are 110 000 samples at an hour of
// 250 ksps: one lane per block walked them for 3 ms before the real walkers could start).  The response only steers a guess, so
// it is taken as the linear functional it is in exact arithmetic, sum over i of term[i] k^(end - 1 - i), in double: lane l folds
// the 16-byte vectors l, l + 64, ... of the block by Horner's rule with k^256 (coalesced KiB loads), weighs its sum with one
// power of k, the lanes add up.  It differs from the rounded recurrence's response by that recurrence's own rounding noise
// (tens of float ulps over a time constant), which is what the walkers' 16 time constants of warm-up are there to absorb.
template <typename T>
__device__ __forceinline__ void k_lock_ema_zero_wave(const T *__restrict__ term, long long n, T lock_alpha,
                                                     const PllLockInfo<T> *__restrict__ info, long long B, double *__restrict__ zresp)
{
    constexpr int VN = Vec16<T>::N;
    const long long lock_at = info->lock_sample;
    if (lock_at < 0) return;
    const long long S = lock_at + 1;
    const long long j = S / B + (long long)blockIdx.x;
    long long start = j * B;
    if (start >= n) return;
    if (start < S) start = S;
    const long long end = ((j + 1) * B < n) ? (j + 1) * B : n;
    const int lane = threadIdx.x & 63;
    const double k = 1.0 - (double)lock_alpha, lnk = log(k);
    double kv = 1.0;                                        // k^(64 VN): from one of a lane's vectors to its next
    for (int u = 0; u < 64 * VN; u++) kv *= k;
    const long long a0 = (start + VN - 1) / VN * VN;        // first aligned sample
    const long long nv = (end > a0) ? (end - a0) / VN : 0;  // whole vectors
    double A = 0.0;
    long long v_last = -1;
    for (long long v = lane; v < nv; v += 64) {
        const Vec16<T> x = *reinterpret_cast<const Vec16<T> *>(term + a0 + v * VN);
        double c = 0.0;
#pragma unroll
        for (int w = 0; w < VN; w++) c = c * k + (double)x.v[w];
        A = A * kv + c;
        v_last = v;
    }
    double R = 0.0;
    if (v_last >= 0) R = A * exp(lnk * (double)(end - 1 - (a0 + v_last * VN + VN - 1)));
    if (lane == 0) {                                        // the few samples in front of and behind the whole vectors
        for (long long i = start; i < a0 && i < end; i++) R += (double)term[i] * exp(lnk * (double)(end - 1 - i));
        for (long long i = a0 + nv * VN; i < end; i++) R += (double)term[i] * exp(lnk * (double)(end - 1 - i));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) R += __shfl_xor(R, d);
    if (lane == 0) zresp[j - S / B] = R;
}

// guess[r] = approximate state in front of block r (r counted from the block that holds the lock): guess[r + 1] =
// k^len(r) guess[r] + zresp[r].  One workgroup: every thread composes the maps of a run of blocks, a scan over the threads'
// composites gives each run its start value, the thread walks its run again.
template <typename T, bool AVG = false>
__device__ __forceinline__ void k_lock_ema_guess(const double *__restrict__ zresp, long long n, T lock_alpha,
                                                 const PllLockInfo<T> *__restrict__ info, long long B, double kB,
                                                 double *__restrict__ guess)
{
    __shared__ double s_a[1024], s_b[1024];
    const long long lock_at = info->lock_sample;
    if (lock_at < 0) return;
    const long long S = lock_at + 1;
    const long long j0 = S / B;
    const long long nb = (S < n) ? ((n - 1) / B - j0 + 1) : 0;
    const double k = 1.0 - (double)lock_alpha;
    const double k0 = pow(k, (double)((j0 + 1) * B - S));             // block 0 starts at the lock, not at a block boundary
    const int t = threadIdx.x, NT = blockDim.x;
    const long long per = (nb + NT - 1) / NT;
    const long long r0 = (long long)t * per, r1 = (r0 + per < nb) ? r0 + per : nb;
    double A = 1.0, Bv = 0.0;
    for (long long r = r0; r < r1; r++) {
        const double a = (r == 0) ? k0 : kB;
        A = a * A;
        Bv = a * Bv + zresp[r];
    }
    s_a[t] = A;
    s_b[t] = Bv;
    __syncthreads();
    for (int d = 1; d < NT; d <<= 1) {                                  // inclusive scan of affine maps, later o earlier
        double pa = 1.0, pb = 0.0;
        if (t >= d) { pa = s_a[t - d]; pb = s_b[t - d]; }
        __syncthreads();
        if (t >= d) {
            s_b[t] = s_a[t] * pb + s_b[t];
            s_a[t] = s_a[t] * pa;
        }
        __syncthreads();
    }
    const double L0 = (double)(AVG ? info->st.avg_phase : info->st.locksig);
    double g = (t == 0) ? L0 : s_a[t - 1] * L0 + s_b[t - 1];
    for (long long r = r0; r < r1; r++) {
        guess[r] = g;
        g = ((r == 0) ? k0 : kB) * g + zresp[r];
    }
}

template <typename T, bool AVG = false>
__device__ __forceinline__ void k_lock_ema(const T *__restrict__ term, long long n, T lock_alpha,
                                                  const PllLockInfo<T> *__restrict__ info, long long B, long long W,
                                                  T *__restrict__ lock_out, EmaSeam<T> *__restrict__ seams,
                                                  const double *__restrict__ guess,
                                                  long long chunk /* AVG: store the value behind every chunk's last sample only */)
{
    __shared__ __attribute__((aligned(16))) unsigned char ring[PDT_EMA_PF * PDT_RING_SLOT];
    const long long lock_at = info->lock_sample;
    if (lock_at < 0) return;
    const long long S = lock_at + 1;
    const long long j0 = S / B;
    const long long j = j0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long start = j * B;
    if (start >= n) return;
    if (start < S) start = S;
    const long long end = ((j + 1) * B < n) ? (j + 1) * B : n;
    long long ws = start - W;
    const T L_lock = AVG ? info->st.avg_phase : info->st.locksig;
    T L = L_lock;
    if (ws < S) ws = S;
    else if (guess) {                                  // W is a whole number of blocks: start the warm-up from the guess there
        const long long jw = ws / B;
        L = (T)guess[jw - j0];
        ws = jw * B;
        if (ws < S) { ws = S; L = L_lock; }
    }
    const double k = 1.0 - (double)lock_alpha;
    ema_range<T, 0>(term, lock_out, ws, start, L, k, ring);
    EmaSeam<T> sm;
    sm.v0 = L;
    if (AVG && chunk > 0) {
        ema_range<T, 2>(term, lock_out, start, end, L, k, ring, chunk);
        if (end == n) lock_out[n - 1] = L;                 // the short last chunk ends with the stream
    } else
        ema_range<T, 1>(term, lock_out, start, end, L, k, ring);
    sm.v1 = L;
    seams[j - j0] = sm;
}

template <typename T>
__device__ __forceinline__ void k_lock_ema_fix(const T *__restrict__ term, long long n, T lock_alpha,
                                                      const PllLockInfo<T> *__restrict__ info, long long B,
                                                      T *__restrict__ lock_out, EmaSeam<T> *__restrict__ seams,
                                                      unsigned *__restrict__ fixes_out)
{
    const long long lock_at = info->lock_sample;
    if (lock_at < 0) return;
    const long long S = lock_at + 1;
    const long long j0 = S / B;
    const long long nb = (S < n) ? ((n - 1) / B - j0 + 1) : 0;
    const double k = 1.0 - (double)lock_alpha;
    unsigned fixes = 0;
    long long r = 1;
    while (r < nb) {
        const long long mine = r + threadIdx.x;
        bool bad = false;
        if (mine < nb) bad = !bits_equal(seams[mine - 1].v1, seams[mine].v0);
        const unsigned long long mask = __ballot(bad);
        if (mask == 0) { r += 64; continue; }
        const long long rb = r + (long long)__builtin_ctzll(mask);
        fixes++;
        const T truth = seams[rb - 1].v1;
        if (threadIdx.x == 0) {
            T L = truth;
            const long long start = (j0 + rb) * B;
            const long long end = ((j0 + rb + 1) * B < n) ? (j0 + rb + 1) * B : n;
            for (long long i = start; i < end; i++) {
                L = (T)((double)L * k + (double)term[i]);
                lock_out[i] = L;
            }
            seams[rb].v0 = truth;
            seams[rb].v1 = L;
        }
        __threadfence_block();
        __syncthreads();
        r = rb + 1;
    }
    if (threadIdx.x == 0) *fixes_out += fixes;
}

// Squelch on its own (common/AGC.c:24-46): x[i] = 0 where lock[i] < thr.  The file programs apply it after the AGC (fused
// into k_agc_block); the sound-card twin applies it between PLL and FIR (POESTIPdemodPortAudio/main.c:370).
// Elementwise, 4 samples per thread.
template <typename T>
__device__ __forceinline__ void k_squelch(T *__restrict__ x, const T *__restrict__ lock, long long n, T thr)
{
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const long long i = i0 + u;
        if (i < n && lock[i] < thr) x[i] = 0;
    }
}

// decay branch of one AGC step as an affine map of the gain, g -> A g + B (see the AGC section below)
struct AgcMap { double A, B; };

// ------------------------------------------------------------------------------------------
// FIR (reference: common/LowPassFilter.c:13-71 interpolating, :76-125 in place)
// ------------------------------------------------------------------------------------------
// Interpolating form.  Output g (global index since stream start), M = g / interp,
// r = g % interp.  The reference's ring holds input m in slot interp*(m mod K), K = N/interp,
// and accumulates slots in ascending order, i.e. the K most recent inputs in ascending
// (m mod K) order, starting from +0 with separate multiply and add.  Input m pairs with
// tap h[N-1-(g - m*interp)]; inputs before the stream start are +0.
#define PDT_FIR_THREADS 256
template <typename T>
__device__ __forceinline__ void k_fir_interp(const T *__restrict__ in, long long n_in, int interp, int K,
                                                                 const T *__restrict__ taps, T *__restrict__ out,
                                                                 int outs_per_thread)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *s_taps = reinterpret_cast<T *>(smem_raw);
    const int N = K * interp;
    T *s_in = s_taps + N;
    const long long n_out = n_in * interp;
    const long long tile = (long long)PDT_FIR_THREADS * outs_per_thread;
    const long long g0 = (long long)blockIdx.x * tile;
    if (g0 >= n_out) return;
    const long long g1 = (g0 + tile < n_out) ? g0 + tile : n_out;
    const long long m_lo = g0 / interp - (K - 1);          // oldest input any output of the tile needs
    const long long m_hi = (g1 - 1) / interp;
    const int n_stage = (int)(m_hi - m_lo + 1);
    for (int t = threadIdx.x; t < N; t += PDT_FIR_THREADS) s_taps[t] = taps[t];
    for (int t = threadIdx.x; t < n_stage; t += PDT_FIR_THREADS) {
        const long long m = m_lo + t;
        s_in[t] = (m >= 0) ? in[m] : (T)0;
    }
    __syncthreads();
    for (int u = 0; u < outs_per_thread; u++) {
        const long long g = g0 + (long long)u * PDT_FIR_THREADS + threadIdx.x;
        if (g >= g1) break;
        const long long M = g / interp;
        const int r = (int)(g - M * interp);
        int k = (int)(M % K);                              // age (in inputs) of the slot-0 sample
        T y = 0;
        for (int t = 0; t < K; t++) {
            const long long m = M - k;
            y = y + s_taps[N - 1 - r - k * interp] * s_in[(int)(m - m_lo)];
            k = (k == 0) ? K - 1 : k - 1;
        }
        out[g] = y;
    }
}

// f(integral_constant<c>) for c = W, W + STEP, ... < K
template <int W, int STEP, int K, int I = 0, typename F>
__device__ __forceinline__ void fir_residues(F &f)
{
    if constexpr (W + STEP * I < K) {
        f(std::integral_constant<int, W + STEP * I>{});
        fir_residues<W, STEP, K, I + 1>(f);
    }
}

// Register-tiled interpolating form (K = taps per polyphase branch and INTERP known at compile time).
// The accumulation ORDER of an output depends on M mod K (the ring rotates), so the work is laid
// out by residue class: a workgroup owns 64*K consecutive inputs starting at a multiple of K, and a
// wavefront takes one residue c = M mod K at a time, lane j computing the INTERP outputs of
// M = m0 + c + K*j.  Ring slot t then holds input m0 + K*j + t (t <= c) or m0 + K*(j-1) + t (t > c),
// every lane of the wavefront walks the slots 0..K-1 in the reference's order, and the tap of slot
// t is the same for all lanes: it comes from a table rotated per residue, rot[c][t][r] =
// h[N-1-r-((c-t) mod K)*INTERP] (built once on the host) and is fetched with scalar loads into
// SGPRs (broadcast LDS reads of the taps made the kernel LDS-bandwidth bound).  The
// slot loop is fully unrolled; each staged input is read once for INTERP multiply-adds (separate
// multiply and add, as the reference).  Outputs go through LDS and leave with coalesced stores.
// HBM traffic = the algorithmic 4 B in + 4*INTERP B out per input sample.
template <typename T, int INTERP, int K>
__device__ __forceinline__ void k_fir_interp_rt(const T *__restrict__ in, long long n_in,
                                                                    const T *__restrict__ rot /* host-built rotated taps */,
                                                                    T *__restrict__ out, AgcMap *__restrict__ tile_maps,
                                                                    T agc_decay)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr int TI = 64 * K;                               // inputs (= values of M) per workgroup
    constexpr int RS = INTERP;                               // row stride of the tap table
    constexpr int LS = K + 1;                                // padded row of K inputs: lanes hit distinct banks
    T *s_out = reinterpret_cast<T *>(smem_raw);              // TI * INTERP outputs
    T *s_in = s_out + TI * INTERP;                           // 65 rows of K inputs: row q = inputs m0 + K*(q-1) ..
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int j = threadIdx.x & 63;
    const long long n_out = n_in * INTERP;
    constexpr int n_tile = TI * INTERP;                      // multiple of 4
    const long long n_tiles = (n_in + TI - 1) / TI;
    // persistent workgroups: the tap table is loaded once, then tile after tile
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long m0 = tile * TI;                      // multiple of K
        __syncthreads();                                     // previous tile's outputs have left s_out / s_in
        for (int t = threadIdx.x; t < 65 * K; t += PDT_FIR_THREADS) {
            const long long m = m0 - K + t;
            const int q = t / K;
            s_in[q * LS + (t - q * K)] = (m >= 0 && m < n_in) ? in[m] : (T)0;
        }
        __syncthreads();
        // The residue is a compile-time constant of each copy of the body (26 copies; wavefront w takes residues w, w + 4, ...):
        // which of the lane's two rows slot t is read from, the LDS offsets and the tap addresses are then all immediates.
        // With a run-time residue every read cost a scalar compare + select and a vector add for its address: 110 vector
        // and 69 scalar instructions per 64 outputs at INTERP 1 (PMC) against the 52 of the arithmetic.
        const T *w0 = s_in + j * LS;                             // slot t >  c: input m0 + K*(j-1) + t; t <= c: one row further
        auto residue = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            // wave-uniform address, read-only table: constant address space, so that the loads are scalar loads into SGPRs
            // whatever the compiler can prove about the stores around them
            // (each residue's K * INTERP taps start on a 64-byte boundary: the loads can be as wide as the ISA has them)
            constexpr int CS = (K * INTERP + 15) & ~15;
            const __attribute__((address_space(4))) T *h =
                (const __attribute__((address_space(4))) T *)__builtin_assume_aligned(rot + c * CS, 64);
            T y[INTERP];
#pragma unroll
            for (int r = 0; r < INTERP; r++) y[r] = 0;
            T x[K];                                              // the lane's ring, all reads in flight together
#pragma unroll
            for (int t = 0; t < K; t++) x[t] = w0[(t <= c) ? LS + t : t];
            // ... and the residue's taps: one block copy from a wave-uniform address = a handful of wide scalar loads
            T hv[K][INTERP];
#pragma unroll
            for (int t = 0; t < K; t++)
#pragma unroll
                for (int r = 0; r < INTERP; r++) hv[t][r] = h[t * RS + r];
            if constexpr (std::is_same<T, float>::value && INTERP >= 2) {
                // Round 6: the INTERP outputs of an input position two at a time in packed arithmetic (v_pk_mul_f32 / v_pk_add_f32:
                // per component exactly the reference's separate multiply and add, in its slot order, from +0 -- as k_mix_fir
                // pairs a lane's two outputs at INTERP 1).  At INTERP 8 the kernel is bound by its vector instructions -- 52 per
                // output, 540 M outputs for an hour at 18.75 ksps: 1.83 ms where the 36 B per input sample are 0.42 ms of HBM
                // time (profiles/r6) -- and the taps of outputs r, r + 1 lie side by side in the table: one SGPR pair.
                typedef float v2f __attribute__((ext_vector_type(2)));
                v2f y2[INTERP / 2];
#pragma unroll
                for (int q = 0; q < INTERP / 2; q++) { y2[q].x = 0; y2[q].y = 0; }
#pragma unroll
                for (int t = 0; t < K; t++) {
                    v2f xx;
                    xx.x = x[t]; xx.y = x[t];
#pragma unroll
                    for (int q = 0; q < INTERP / 2; q++) {
                        v2f hh;
                        hh.x = hv[t][2 * q]; hh.y = hv[t][2 * q + 1];
                        y2[q] = y2[q] + hh * xx;
                    }
                    if constexpr (INTERP & 1) y[INTERP - 1] = y[INTERP - 1] + hv[t][INTERP - 1] * x[t];
                }
#pragma unroll
                for (int q = 0; q < INTERP / 2; q++) { y[2 * q] = y2[q].x; y[2 * q + 1] = y2[q].y; }
            } else {
#pragma unroll
            for (int t = 0; t < K; t++) {
#pragma unroll
                for (int r = 0; r < INTERP; r++) y[r] = y[r] + hv[t][r] * x[t];
            }
            }
#pragma unroll
            for (int r = 0; r < INTERP; r++) s_out[(c + K * j) * INTERP + r] = y[r];
        };
        constexpr int NWV = PDT_FIR_THREADS / 64;
        static_assert(NWV == 4, "the residue dispatch below is written for four wavefronts");
        if (wave == 0) fir_residues<0, NWV, K>(residue);
        else if (wave == 1) fir_residues<1, NWV, K>(residue);
        else if (wave == 2) fir_residues<2, NWV, K>(residue);
        else fir_residues<3, NWV, K>(residue);
        __syncthreads();
        const long long g0 = m0 * INTERP;                    // multiple of 4
        if (tile_maps) {
            // fused: the AGC's affine model of this tile (k_agc_affine's job) while its outputs are still in LDS.
            // Every thread composes PER consecutive outputs in order, the 256 partial maps are reduced in
            // order -- shuffles inside a wavefront, LDS across the four -- and thread 0 writes the tile map.
            constexpr int PER = (n_tile + PDT_FIR_THREADS - 1) / PDT_FIR_THREADS;
            const int cnt = (int)((n_out - g0 < n_tile) ? (n_out - g0) : n_tile);
            const double r = (double)agc_decay;
            double A = 1.0, Bc = 0.0;
            const int i0 = threadIdx.x * PER;
#pragma unroll 4
            for (int u = 0; u < PER; u++) {
                const int i = i0 + u;
                if (i < cnt) {
                    const double a = 1.0 - r * (double)Real<T>::abs(s_out[i]);
                    A = a * A;
                    Bc = a * Bc + r;
                }
            }
            for (int d = 1; d < 64; d <<= 1) {               // ordered: lane i absorbs lanes i+1 .. i+2d-1
                const double Ar = __shfl_down(A, d), Br = __shfl_down(Bc, d);
                if (((threadIdx.x & 63) & (2 * d - 1)) == 0) {
                    Bc = Ar * Bc + Br;
                    A = Ar * A;
                }
            }
            double *s_red = reinterpret_cast<double *>(s_in);            // the input rows are no longer needed
            if ((threadIdx.x & 63) == 0) { s_red[2 * wave] = A; s_red[2 * wave + 1] = Bc; }
            __syncthreads();
            if (threadIdx.x == 0) {
                double tA = 1.0, tB = 0.0;
                for (int w = 0; w < PDT_FIR_THREADS / 64; w++) {
                    tB = s_red[2 * w] * tB + s_red[2 * w + 1];
                    tA = s_red[2 * w] * tA;
                }
                AgcMap m;
                m.A = tA;
                m.B = tB;
                tile_maps[tile] = m;
            }
        }
        if (g0 + n_tile <= n_out) {
            constexpr int VN = Vec16<T>::N;
            for (int t = threadIdx.x * VN; t < n_tile; t += PDT_FIR_THREADS * VN)
                *reinterpret_cast<Vec16<T> *>(out + g0 + t) = *reinterpret_cast<const Vec16<T> *>(s_out + t);
        } else {
            for (int t = threadIdx.x; t < n_tile; t += PDT_FIR_THREADS)
                if (g0 + t < n_out) out[g0 + t] = s_out[t];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Mix + FIR in one kernel (POES float build, INTERP 1 -- sample rates from 150 ksps up: the hour-long 250 ksps captures)
//
// After the lock the PLL output of sample i is Im(x_i e^{-j phi_i}) (CarrierTrackingPLL.c:106-113) and nothing but the filter
// reads it (LowPassFilter.c:43-70).  Unfused, k_pll_mix writes that stream (4 B/sample) and k_fir_interp_rt reads it back;
// here it only ever exists in LDS:
//   * a workgroup takes one run of 208 positions (8 ring revolutions of K = 26) of each of the 64 blocks of an LT tile: the
//     phases arrive as 59 whole rows of the tile (59 KiB of consecutive bytes: the run and 28 positions in front of it -- the
//     filter's K - 1 = 25 older inputs, rounded to whole phase vectors), I/Q as 64 stretches of 944 bytes;
//   * phase A transposes the phases into LDS (row = block), phase B replaces each by the mixed sample (branch-free sincosf,
//     64 consecutive positions of a row per wavefront instruction; samples up to the lock come from the acquisition's
//     output), optionally also writing it out (pdt_keep_pll);
//   * phase C filters: lane = block, so all lanes of a wavefront are at the same ring phase (blocks and runs start at multiples
//     of K: B is a multiple of lcm(64, 208) = 832) and take their taps -- rotated per residue on the host, as for
//     k_fir_interp_rt -- from SGPRs; a wavefront computes the outputs s = 26 w + c and s + 104 of every row for the 26
//     residues c in turn, the lane's ring of K inputs in registers, the two outputs of a lane side by side in packed
//     arithmetic (v_pk_mul_f32 / v_pk_add_f32: per component exactly the separate multiply and add of the reference, in its
//     slot order, starting from +0); 26 packed instructions per output against 52, and one LDS read per output;
//   * phase D composes the AGC's affine map of every row's run (the guess of the gain at the AGC block boundaries: any
//     grouping is legal there) and stores the outputs, 64 rows of 832 consecutive bytes.
// HBM traffic: 4 B (I/Q) + 4 B (phase) in (+ 12 % for the runs' halos), 4 B out.
// ------------------------------------------------------------------------------------------
#define PDT_MF_RUN 208
#define PDT_MF_HALO 28
#define PDT_MF_COLS (PDT_MF_RUN + PDT_MF_HALO)
#define PDT_MF_LSI 237                 // LDS row strides (odd: 64 rows at one column hit 64 banks)
typedef float pdt_v2f __attribute__((ext_vector_type(2)));

// QT (round 4): the kernel also writes the input term of the averagePhase EMA, averagePhaseAlpha * |arctan2(out)|
// (CarrierTrackingPLL.c:117-124), for every sample it mixes -- both components of the mixed sample are in registers here, so the
// per-chunk reports (pdt_keep_quality) no longer need a second pass over phases and I/Q with its own sincosf (k_pll_mix<AVGTERM>,
// 3 ms beside this kernel on the other stream: together they took 7 ms).
template <int K, int FMT, int NWV = 4, bool QT = false>
__device__ __forceinline__ void k_mix_fir(IqSrc pcm, const float *__restrict__ phi_lt, const float *pll_pre /* may alias pll_out */,
                                          long long n, long long B, const PllLockInfo<float> *__restrict__ info,
                                          const float *__restrict__ rot /* host-built rotated taps */, float *__restrict__ out,
                                          float *pll_out /* nullptr: the PLL output is not kept; the host passes the same buffer as
                                                            pll_pre (reads before the lock, writes behind it): no __restrict__ */,
                                          AgcMap *__restrict__ run_maps, float agc_decay, float *__restrict__ term_out = nullptr,
                                          long long pll_from = 0 /* pll_out is written from this sample on only (a stream segment
                                                                    keeps just the tail the next segment's filter starts from) */)
{
    static_assert(PDT_MF_RUN == 8 * K && PDT_MF_HALO >= K - 1 && PDT_MF_HALO % 4 == 0, "run = 8 ring revolutions, halo = whole phase vectors");
    static_assert(NWV == 4 || NWV == 8, "four or eight wavefronts per workgroup");
    constexpr int RPW = 64 / NWV;                         // rows a wavefront owns in phases B and D
    constexpr int NSEG = 2 * NWV;                         // stretches of a row's run the AGC map is composed from
    // 59 KiB + 8 (16) KiB: two workgroups per CU (one loads while the other computes)
    __shared__ __attribute__((aligned(16))) float s_in[64 * PDT_MF_LSI];
    __shared__ double s_maps[NSEG * 64 * 2];
    const int lane = (int)threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const long long runs = B / PDT_MF_RUN;
    const long long w = (long long)blockIdx.x / runs, g = (long long)blockIdx.x - w * runs;
    const long long p0 = g * PDT_MF_RUN;
    const long long blk0 = w << 6;
    if (blk0 * B + p0 >= n) return;                       // (the other rows of the tile lie further on)
    const long long lock_at = info->lock_sample;
    const long long S = (lock_at < 0) ? n : lock_at + 1;  // first sample mixed with a tracked phase

    // ---- every global load of the workgroup is issued here, back to back: the I/Q words phase B will need (as they are in the
    // capture: 4 or 8 bytes per sample), then (A) the phases of the run and its halo, transposed into LDS (row = block).
    // Phase B's work list: a wavefront owns 16 rows x 236 columns = 59 x 64 (row, column) pairs, pair number 64 it + lane.
    typedef typename std::conditional<FMT == 0, int, float2>::type Raw;
    constexpr int NIT = (RPW * PDT_MF_COLS + 63) / 64;
    constexpr bool RAGGED = NIT * 64 != RPW * PDT_MF_COLS;       // (eight wavefronts: the last instruction is half empty)
    // interior workgroup: every sample it touches lies behind the lock and inside the capture (no test per sample)
    const bool interior = (blk0 * B + p0 - PDT_MF_HALO >= S) && ((blk0 + 63) * B + p0 + PDT_MF_RUN <= n);
    Raw raw[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int idx = it * 64 + lane, rr = idx / PDT_MF_COLS, col = idx - rr * PDT_MF_COLS;
        const long long i = (blk0 + wave * RPW + rr) * B + p0 - PDT_MF_HALO + col;
        if constexpr (FMT == 0) raw[it] = 0;
        else raw[it] = make_float2(0.0f, 0.0f);
        if (interior && (!RAGGED || idx < RPW * PDT_MF_COLS)) raw[it] = reinterpret_cast<const Raw *>(pcm.p)[i];   // (the other workgroups load as they go)
    }
    {
        const float *tile = phi_lt + blk0 * B;
        constexpr int NQ = (PDT_MF_COLS / 4 + NWV - 1) / NWV;          // phase vectors per wavefront
        float4 v[NQ];
#pragma unroll
        for (int u = 0; u < NQ; u++) {
            const int q = wave + NWV * u;
            const long long p = p0 - PDT_MF_HALO + 4 * q;              // position of the vector's first sample in its block
            const long long i0 = (blk0 + lane) * B + p;
            v[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (q < PDT_MF_COLS / 4 && (interior || (i0 >= 0 && i0 < n && i0 + 4 > S))) {
                const float *src;
                if (p >= 0) src = tile + (p >> 2) * 256 + lane * 4;
                else if (lane > 0) src = tile + ((B + p) >> 2) * 256 + (lane - 1) * 4;        // the end of the block in front
                else src = tile - 64 * B + ((B + p) >> 2) * 256 + 63 * 4;                     // ... which is the previous tile's last
                v[u] = *reinterpret_cast<const float4 *>(src);
            }
        }
#pragma unroll
        for (int u = 0; u < NQ; u++) {
            const int q = wave + NWV * u;
            if (q < PDT_MF_COLS / 4) {
                float *d = s_in + lane * PDT_MF_LSI + 4 * q;
                d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
            }
        }
    }
    __syncthreads();
    // ---- B: phase -> mixed sample, in place
    {
        float *wrows = s_in + wave * RPW * PDT_MF_LSI;
        auto mix = [&](const Raw &r, float ph, float &term) {
            float a, b, sn, cs;
            if constexpr (FMT == 0) {                                  // I | Q << 16, value / 32768 (wave.c:127-172)
                a = (float)(short)(r & 0xffff) / 32768.0f;
                b = (float)(short)(r >> 16) / 32768.0f;
            } else {
                a = r.x;
                b = r.y;
            }
            sincosf_flat(ph, sn, cs);
            const float c = cs, d = -sn;
            const float o = a * d + b * c;
            if (QT) {
                const float o_re = a * c - b * d;
                term = 0.00005f * __builtin_fabsf(arctan2_ref(o, o_re));
            }
            return o;
        };
        if (interior) {
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int idx = it * 64 + lane, rr = idx / PDT_MF_COLS, col = idx - rr * PDT_MF_COLS;
                float *cell = wrows + idx + rr;                        // rr * LSI + col, LSI = COLS + 1
                if (RAGGED && idx >= RPW * PDT_MF_COLS) continue;
                float tq = 0.0f;
                const float x = mix(raw[it], *cell, tq);
                *cell = x;
                if (pll_out && col >= PDT_MF_HALO) {
                    const long long io = (blk0 + wave * RPW + rr) * B + p0 - PDT_MF_HALO + col;
                    if (io >= pll_from) pll_out[io] = x;
                }
                if (QT && col >= PDT_MF_HALO) term_out[(blk0 + wave * RPW + rr) * B + p0 - PDT_MF_HALO + col] = tq;
                if ((it & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // four evaluations interleaved, not fifty-nine (registers)
            }
        } else {
#pragma unroll 1
            for (int it = 0; it < NIT; it++) {
                const int idx = it * 64 + lane, rr = idx / PDT_MF_COLS, col = idx - rr * PDT_MF_COLS;
                const long long i = (blk0 + wave * RPW + rr) * B + p0 - PDT_MF_HALO + col;
                float *cell = wrows + idx + rr;
                if (RAGGED && idx >= RPW * PDT_MF_COLS) continue;
                float x = 0.0f;
                if (i >= S && i < n) {
                    float tq = 0.0f;
                    x = mix(reinterpret_cast<const Raw *>(pcm.p)[i], *cell, tq);
                    if (pll_out && col >= PDT_MF_HALO && i >= pll_from) pll_out[i] = x;
                    if (QT && col >= PDT_MF_HALO) term_out[i] = tq;
                } else if (i >= 0 && i < n)
                    x = pll_pre[i];                                    // up to the lock: the acquisition's output
                *cell = x;
            }
        }
    }
    __syncthreads();
    // ---- C: the filter
    const long long row_base = (blk0 + lane) * B + p0;                    // natural index of this lane's output 0
    const int n_valid = (int)((n - row_base >= PDT_MF_RUN) ? PDT_MF_RUN : ((n - row_base > 0) ? n - row_base : 0));
    double mA[2] = {1.0, 1.0}, mB[2] = {0.0, 0.0};                        // AGC maps of this lane's two stretches of 26 outputs
    const bool full_rows = (blk0 + 63) * B + p0 + PDT_MF_RUN <= n;        // every row's run lies inside the capture
    {
        float *xin = s_in + lane * PDT_MF_LSI + PDT_MF_HALO;              // xin[m] = input m of the run, m >= -HALO
        // four wavefronts: wavefront w takes the ring revolution that starts at output 26 w (and the one 104 further on), all 26
        // residues; eight: half a revolution each -- residues 0..12 (even w) or 13..25 (odd w) of the revolution at 26 (w / 2)
        const int s0 = (NWV == 4) ? wave * K : (wave >> 1) * K;
        const bool upper = NWV == 8 && (wave & 1);                        // starts at residue 13
        constexpr int H2 = PDT_MF_RUN / 2;
        const double rdec = (double)agc_decay;
        constexpr int NRES = (NWV == 4) ? K : K / 2;                       // residues a wavefront takes
        pdt_v2f x[K], nx[NRES];
        // the ring in front of output s0 (a multiple of K): slot t >= 1 holds input s0 - K + t; slot 0 is filled by residue 0.
        // The K inputs that enter the ring while the wavefront works are taken now as well: the outputs then go where the
        // inputs were (another wavefront's inputs, read before the barrier)
        // (in front of residue c0 the slots below c0 already hold this revolution's inputs)
#pragma unroll
        for (int t = 0; t < K; t++) {
            const int m = (upper && t < K / 2) ? s0 + t : s0 - K + t;
            x[t].x = xin[m];
            x[t].y = xin[m + H2];
        }
        const int c0 = upper ? K / 2 : 0;
#pragma unroll
        for (int t = 0; t < NRES; t++) { nx[t].x = xin[s0 + c0 + t]; nx[t].y = xin[s0 + c0 + H2 + t]; }
        __syncthreads();
        auto residue = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            x[c] = nx[(NWV == 4) ? c : c % (K / 2)];
            constexpr int CS = (K + 15) & ~15;
            const __attribute__((address_space(4))) float *h =
                (const __attribute__((address_space(4))) float *)__builtin_assume_aligned(rot + c * CS, 64);
            float hv[K];
#pragma unroll
            for (int t = 0; t < K; t++) hv[t] = h[t];
            pdt_v2f y;
            y.x = 0; y.y = 0;
#pragma unroll
            for (int t = 0; t < K; t++) {
                pdt_v2f hh;
                hh.x = hv[t]; hh.y = hv[t];
                y = y + hh * x[t];
            }
            xin[s0 + c] = y.x;
            xin[s0 + H2 + c] = y.y;
            if (run_maps) {
                // (the gain guess is free to round as it likes: fused operations, half the instructions)
                const double a0 = __builtin_fma(-rdec, (double)__builtin_fabsf(y.x), 1.0);
                const double a1 = __builtin_fma(-rdec, (double)__builtin_fabsf(y.y), 1.0);
                if (full_rows) {
                    mB[0] = __builtin_fma(a0, mB[0], rdec);
                    mA[0] = a0 * mA[0];
                    mB[1] = __builtin_fma(a1, mB[1], rdec);
                    mA[1] = a1 * mA[1];
                } else {
                    if (s0 + c < n_valid) {
                        mB[0] = __builtin_fma(a0, mB[0], rdec);
                        mA[0] = a0 * mA[0];
                    }
                    if (s0 + H2 + c < n_valid) {
                        mB[1] = __builtin_fma(a1, mB[1], rdec);
                        mA[1] = a1 * mA[1];
                    }
                }
            }
        };
        if constexpr (NWV == 4) fir_residues<0, 1, K>(residue);
        else if (upper) fir_residues<K / 2, 1, K>(residue);
        else fir_residues<0, 1, K / 2>(residue);
    }
    // ---- D: the rows' AGC maps (the stretches of each in the order of their outputs), then the outputs
    if (run_maps) {
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            s_maps[((hf * NWV + wave) * 64 + lane) * 2] = mA[hf];
            s_maps[((hf * NWV + wave) * 64 + lane) * 2 + 1] = mB[hf];
        }
    }
    __syncthreads();
    if (run_maps && (int)threadIdx.x < 64 && row_base < n) {
        double tA = 1.0, tB = 0.0;
#pragma unroll
        for (int sg = 0; sg < NSEG; sg++) {
            const double A = s_maps[(sg * 64 + lane) * 2], Bc = s_maps[(sg * 64 + lane) * 2 + 1];
            tB = A * tB + Bc;
            tA = A * tA;
        }
        AgcMap m;
        m.A = tA;
        m.B = tB;
        run_maps[row_base / PDT_MF_RUN] = m;
    }
#pragma unroll 4
    for (int rr = 0; rr < RPW; rr++) {
        const int l = wave * RPW + rr;
        const long long ob = (blk0 + l) * B + p0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int col = lane + 64 * k;
            if (col < PDT_MF_RUN && ob + col < n) out[ob + col] = s_in[l * PDT_MF_LSI + PDT_MF_HALO + col];
        }
    }
}

// In-place form (ARGOS): y[i] = sum_{k<N} h[k] * x[i-(N-1-k)], oldest first.
template <typename T>
__device__ __forceinline__ void k_fir_plain(const T *__restrict__ in, long long n, int N,
                                                                const T *__restrict__ taps, T *__restrict__ out,
                                                                int outs_per_thread)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *s_taps = reinterpret_cast<T *>(smem_raw);
    T *s_in = s_taps + N;
    const long long tile = (long long)PDT_FIR_THREADS * outs_per_thread;
    const long long g0 = (long long)blockIdx.x * tile;
    if (g0 >= n) return;
    const long long g1 = (g0 + tile < n) ? g0 + tile : n;
    const long long m_lo = g0 - (N - 1);
    const int n_stage = (int)(g1 - m_lo);
    for (int t = threadIdx.x; t < N; t += PDT_FIR_THREADS) s_taps[t] = taps[t];
    for (int t = threadIdx.x; t < n_stage; t += PDT_FIR_THREADS) {
        const long long m = m_lo + t;
        s_in[t] = (m >= 0) ? in[m] : (T)0;
    }
    __syncthreads();
    for (int u = 0; u < outs_per_thread; u++) {
        const long long g = g0 + (long long)u * PDT_FIR_THREADS + threadIdx.x;
        if (g >= g1) break;
        const int base = (int)(g - (N - 1) - m_lo);
        T y = 0;
        for (int k = 0; k < N; k++) y = y + s_taps[k] * s_in[base + k];
        out[g] = y;
    }
}

// ------------------------------------------------------------------------------------------
// StaticGain (reference: common/AGC.c:48-75): half-weight EMA of |x| over the first chunk.
// Magnitudes in parallel, the 2-op recurrence on one lane.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void k_static_gain(IqSrc pcm, long long n0, T *__restrict__ mag_scratch,
                                                      T desired, double override_norm, T *__restrict__ norm_out)
{
    if (override_norm != 0.0) {
        if (threadIdx.x == 0) *norm_out = (T)override_norm;
        return;
    }
    // Fast path.  avg <- (avg + m)/2 forgets its past within a few dozen steps, and the step is a
    // monotone map of avg (rounding is monotone), while avg always lies in [0, max m].  So two chains
    // over the last TAIL magnitudes, started from 0 and from the chunk's largest magnitude, bracket
    // the true chain; when they end on the same float -- they always do -- that float IS the
    // reference's result, whatever came before.  Otherwise the full chain below runs.
    constexpr int TAIL = 192;
    constexpr int CAP = 16384;
    __shared__ T s_mag[CAP];
    __shared__ T s_red[256];
    if (n0 > TAIL) {
        T mx = 0;
        for (long long i = threadIdx.x; i < n0; i += blockDim.x) {
            T a, b;
            IqSample<T>::get(pcm, i, a, b);
            const T m = Real<T>::hypot(a, b);
            mx = (m > mx) ? m : mx;
            if (i >= n0 - TAIL) s_mag[i - (n0 - TAIL)] = m;
        }
        s_red[threadIdx.x] = mx;
        __syncthreads();
        for (int w = 128; w >= 1; w >>= 1) {
            if ((int)threadIdx.x < w) s_red[threadIdx.x] = (s_red[threadIdx.x + w] > s_red[threadIdx.x]) ? s_red[threadIdx.x + w] : s_red[threadIdx.x];
            __syncthreads();
        }
        T a = (threadIdx.x == 0) ? (T)0 : s_red[0];
        if (threadIdx.x < 2) {
            for (int k = 0; k < TAIL; k++) {
                a = a + s_mag[k];
                a = a * (T)0.5;
            }
        }
        __syncthreads();
        if (threadIdx.x < 2) s_red[threadIdx.x] = a;
        __syncthreads();
        const bool same = bits_equal(s_red[0], s_red[1]);
        if (same) {
            if (threadIdx.x == 0) *norm_out = desired / s_red[0];
            return;
        }
        __syncthreads();
    }
    T avg = 0;
    for (long long base = 0; base < n0 || base == 0; base += CAP) {
        const long long cnt = (n0 - base < CAP) ? (n0 - base) : CAP;
        __syncthreads();
        for (long long i = threadIdx.x; i < cnt; i += blockDim.x) {
            T a, b;
            IqSample<T>::get(pcm, base + i, a, b);
            s_mag[i] = Real<T>::hypot(a, b);
        }
        for (long long i = cnt + threadIdx.x; i < ((cnt + 7) & ~7ll); i += blockDim.x) s_mag[i] = 0;
        __syncthreads();
        if (threadIdx.x == 0) {
            if (base == 0) avg = (n0 > 0) ? s_mag[0] : (T)0;      // n0 == 0: the reference reads a zeroed buffer
            long long i = 0;
            for (; i + 8 <= cnt; i += 8) {
                T m[8];
#pragma unroll
                for (int u = 0; u < 8; u++) m[u] = s_mag[i + u];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    avg = avg + m[u];
                    avg = avg * (T)0.5;                           // == (T)((double)avg / 2.0), exact
                }
            }
            for (; i < cnt; i++) {
                avg = avg + s_mag[i];
                avg = avg * (T)0.5;
            }
        }
        if (n0 == 0) break;
    }
    if (threadIdx.x == 0) *norm_out = desired / avg;
    (void)mag_scratch;
}

// ------------------------------------------------------------------------------------------
// NormalizingAGC (+Squelch) (reference: common/AGC.c:78-132, :24-46)
// ------------------------------------------------------------------------------------------
template <typename T> struct AgcParams {
    T attack, decay;
    T squelch_thr;
    int squelch;
    T *raw_out;      // optional: the AGC output before Squelch (what ARGOSdemod -r dumps, main.c:273-274); nullptr = not kept
};

template <typename T> __device__ __forceinline__ T agc_step(T x, T &gain, const AgcParams<T> &P)
{
    x = x * gain;
    const T err = Real<T>::abs(x) - (T)1.0;
    // both candidate updates are formed off the critical path; the selects are exact
    const T g_att = gain - err * P.attack;
    const T g_dec = gain - err * P.decay;
    T g = (Real<T>::abs(err) > gain) ? g_att : g_dec;
    g = (g < (T)0) ? (T)10e-5 : g;
    g = (g > (T)5000) ? (T)5000 : g;
    gain = g;
    return x;
}

template <typename T> struct AgcSeam { T g0, g1; };

// run the AGC over [i0, i1) with 16-byte vector loads/stores on the aligned body
// look-ahead depth (16-byte vectors per lane) of the AGC walkers' LDS ring: 256 samples, about 1.7 us of calm arithmetic
#ifndef PDT_AGC_PF
#define PDT_AGC_PF 64
#endif
// A lone wavefront issues about one instruction every four clocks whatever the dependences, so a walker's pace is its
// instruction count.  The exact step costs ten vector instructions (two of them selects behind compares, four the
// range clamps); over a CALM batch of 16 samples -- every |x| <= 1, the gain in [2.5, 4000] at its start, decay <= 0.04
// -- none of the three conditionals can act and four instructions remain:
//   * |x| g <= g and rounding is monotone, so err = fl(|fl(x g)| - 1) lies in [-1, g - 1] and |err| > g is false
//     (g >= 1.3 throughout, below): the decay branch;
//   * one step moves the gain into [g - (g - 1) decay, g + decay] (up to rounding), so over 16 steps it stays within
//     [0.52 g0, g0 + 0.64], i.e. inside [1.3, 4001]: neither clamp;
//   * a NaN sample gives gain = NaN on either path (the compare and both clamp tests are false for NaN), an infinite one
//     fails the |x| <= 1 test, a NaN gain fails the range test.
// Everything else runs the exact step, as before.
template <typename T, int NB>
__device__ __forceinline__ bool agc_calm(const Vec16<T> *b, T gain, T decay)
{
    constexpr int VN = Vec16<T>::N;
    T xm = 0;
#pragma unroll
    for (int k = 0; k < NB; k++)
#pragma unroll
        for (int w = 0; w < VN; w++) xm = Real<T>::max(xm, Real<T>::abs(b[k].v[w]));
    return (decay > (T)0) & (decay <= (T)0.04) & (gain >= (T)2.5) & (gain <= (T)4000) & (xm <= (T)1);
}

template <typename T> __device__ __forceinline__ T agc_step_calm(T x, T &gain, T decay)
{
    x = x * gain;
    const T err = Real<T>::abs(x) - (T)1.0;
    gain = gain - err * decay;
    return x;
}

template <typename T, bool STORE, int PF = PDT_AGC_PF>
__device__ __forceinline__ void agc_range(const T *__restrict__ in, const T *__restrict__ lock, T *__restrict__ out,
                                          long long i0, long long i1, T &gain, const AgcParams<T> &P, unsigned char *ring)
{
    constexpr int VN = Vec16<T>::N;
    constexpr int NB = 16 / VN;            // vectors per 16-sample batch
    constexpr int NBATCH = PF / NB;        // batches the ring holds
    static_assert(PF % NB == 0 && NBATCH >= 3, "the look-ahead ring holds whole batches");
    long long i = i0;
    for (; i < i1 && (i % VN) != 0; i++) {
        T y = agc_step(in[i], gain, P);
        if (STORE) {
            if (P.raw_out) P.raw_out[i] = y;
            if (P.squelch && lock[i] < P.squelch_thr) y = 0;
            out[i] = y;
        }
    }
    const long long nbt = (i1 - i) / 16;   // whole batches
    if (nbt > 0) {
        // ring in LDS (see ring_issue): batch k of the range lives in ring batch k % NBATCH; the loads run NBATCH
        // batches ahead of the arithmetic (up to PF vectors past i1: slack of the device buffers), the LDS reads one
        const unsigned ring0 = (unsigned)(size_t)ring;
        const unsigned char *mine = ring + 16 * (threadIdx.x & 63);
#pragma unroll
        for (int u = 0; u < PF; u++) ring_issue(in + i + u * VN, ring0 + u * PDT_RING_SLOT);
        ring_wait<PF - NB>();
        Vec16<T> xb[NB], xn[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) xb[k] = *reinterpret_cast<const Vec16<T> *>(mine + k * PDT_RING_SLOT);
        int rb = 0;                        // ring batch of the current batch
        for (long long bt = 0; bt < nbt; bt++, i += 16) {
            const int rnext = (rb + 1 == NBATCH) ? 0 : rb + 1;
            // requests younger than the next batch's: NBATCH - 2 batches (the current batch's slots are re-issued below)
            ring_wait<PF - 2 * NB>();
#pragma unroll
            for (int k = 0; k < NB; k++) xn[k] = *reinterpret_cast<const Vec16<T> *>(mine + (rnext * NB + k) * PDT_RING_SLOT);
            Vec16<T> yv[NB];
            if (agc_calm<T, NB>(xb, gain, P.decay)) {
#pragma unroll
                for (int k = 0; k < NB; k++)
#pragma unroll
                    for (int w = 0; w < VN; w++) yv[k].v[w] = agc_step_calm(xb[k].v[w], gain, P.decay);
            } else {
#pragma unroll
                for (int k = 0; k < NB; k++)
#pragma unroll
                    for (int w = 0; w < VN; w++) yv[k].v[w] = agc_step(xb[k].v[w], gain, P);
            }
#pragma unroll
            for (int k = 0; k < NB; k++) {
                if (STORE) {
                    if (P.raw_out) *reinterpret_cast<Vec16<T> *>(P.raw_out + i + k * VN) = yv[k];
                    if (P.squelch) {
                        const Vec16<T> lv = *reinterpret_cast<const Vec16<T> *>(lock + i + k * VN);
#pragma unroll
                        for (int w = 0; w < VN; w++)
                            if (lv.v[w] < P.squelch_thr) yv[k].v[w] = 0;
                    }
                    *reinterpret_cast<Vec16<T> *>(out + i + k * VN) = yv[k];
                }
                ring_issue(in + i + (PF + k) * VN, ring0 + (unsigned)(rb * NB + k) * PDT_RING_SLOT);
                xb[k] = xn[k];
            }
            rb = rnext;
        }
        ring_wait<0>();
    }
    for (; i + VN <= i1; i += VN) {
        const Vec16<T> xv = *reinterpret_cast<const Vec16<T> *>(in + i);
        Vec16<T> yv;
#pragma unroll
        for (int w = 0; w < VN; w++) yv.v[w] = agc_step(xv.v[w], gain, P);
        if (STORE) {
            if (P.raw_out) *reinterpret_cast<Vec16<T> *>(P.raw_out + i) = yv;
            if (P.squelch) {
                const Vec16<T> lv = *reinterpret_cast<const Vec16<T> *>(lock + i);
#pragma unroll
                for (int w = 0; w < VN; w++)
                    if (lv.v[w] < P.squelch_thr) yv.v[w] = 0;
            }
            *reinterpret_cast<Vec16<T> *>(out + i) = yv;
        }
    }
    for (; i < i1; i++) {
        T y = agc_step(in[i], gain, P);
        if (STORE) {
            if (P.raw_out) P.raw_out[i] = y;
            if (P.squelch && lock[i] < P.squelch_thr) y = 0;
            out[i] = y;
        }
    }
}

// Starting guesses for the block-parallel AGC.  In its steady regime (decay branch, no clamp) one
// AGC step is the affine map g -> g (1 - decay |x|) + decay; affine maps compose associatively,
// so the gain at every block boundary is available from a parallel composition in double
// precision -- up to the float rounding noise the real recurrence accumulates (~1e-6 relative).
// That is only a GUESS: each block still replays a warm-up with the exact float recurrence and
// its seam is verified bitwise; but a guess this close cuts the warm-up from ~45 to ~14 time
// constants.
// (struct AgcMap is declared with the FIR kernels above: the register-tiled FIR folds its tile maps)

#define PDT_AFF_TILE 8192
template <typename T>
__device__ __forceinline__ void k_agc_affine(const T *__restrict__ in, long long n, T decay, long long Bk,
                                                     AgcMap *__restrict__ maps)
{
    // one workgroup per AGC block: tiles of 8192 samples are staged in LDS with coalesced loads,
    // every thread composes 32 consecutive samples, the 256 partial maps are scanned in LDS
    // (ordered: thread t's samples precede thread t+1's) and folded into the running block map
    __shared__ T s_x[PDT_AFF_TILE];
    __shared__ double sA[256], sB[256];
    const long long j = blockIdx.x;
    const long long start = j * Bk;
    if (start >= n) return;
    const long long end = (start + Bk < n) ? start + Bk : n;
    const double r = (double)decay;
    double runA = 1.0, runB = 0.0;                     // block map so far (meaningful in thread 0)
    for (long long t0 = start; t0 < end; t0 += PDT_AFF_TILE) {
        const int cnt = (int)((end - t0 < PDT_AFF_TILE) ? (end - t0) : PDT_AFF_TILE);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += 256) s_x[t] = in[t0 + t];
        __syncthreads();
        const int per = PDT_AFF_TILE / 256;
        const int i0 = threadIdx.x * per;
        double A = 1.0, Bc = 0.0;
        for (int u = 0; u < per; u++) {
            const int i = i0 + u;
            if (i < cnt) {
                const double a = 1.0 - r * (double)Real<T>::abs(s_x[i]);
                A = a * A;
                Bc = a * Bc + r;
            }
        }
        sA[threadIdx.x] = A;
        sB[threadIdx.x] = Bc;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            double Ap = 1.0, Bp = 0.0;
            if ((int)threadIdx.x >= d) { Ap = sA[threadIdx.x - d]; Bp = sB[threadIdx.x - d]; }
            __syncthreads();
            if ((int)threadIdx.x >= d) {
                Bc = A * Bp + Bc;
                A = A * Ap;
                sA[threadIdx.x] = A;
                sB[threadIdx.x] = Bc;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const double tA = sA[255], tB = sB[255];   // map of the whole tile
            runB = tA * runB + tB;
            runA = tA * runA;
        }
    }
    if (threadIdx.x == 0) {
        AgcMap m;
        m.A = runA;
        m.B = runB;
        maps[j] = m;
    }
}

// map of AGC block j = its maps_per_block consecutive FIR-tile maps composed in order.  One thread per block (an hour at
// 250 ksps has 540 000 tile maps: composed inside k_agc_guess's single workgroup they cost 1.6 ms of dependent loads).
__device__ __forceinline__ void k_agc_blockmaps(const AgcMap *__restrict__ maps, long long nb, int maps_per_block, long long n_maps,
                                                AgcMap *__restrict__ bmaps)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nb) return;
    AgcMap bm;
    bm.A = 1.0; bm.B = 0.0;
    for (int q = 0; q < maps_per_block; q++) {
        const long long t = j * maps_per_block + q;
        if (t < n_maps) {
            const AgcMap m = maps[t];
            bm.B = m.A * bm.B + m.B;
            bm.A = m.A * bm.A;
        }
    }
    bmaps[j] = bm;
}

// gain at every block boundary = exclusive prefix composition of the block maps applied to the
// initial gain: one workgroup, each thread composes a contiguous slice, the slices are scanned in
// LDS (affine maps form a monoid), then every thread replays its slice from its prefix.
template <typename T>
__device__ __forceinline__ void k_agc_guess(const AgcMap *__restrict__ maps, long long nb, const T *__restrict__ norm,
                                                     double *__restrict__ guesses, int maps_per_block, long long n_maps)
{
    __shared__ double sA[1024], sB[1024];
    const long long per = (nb + 1023) / 1024;
    long long j0 = (long long)threadIdx.x * per, j1 = j0 + per;
    if (j0 > nb) j0 = nb;
    if (j1 > nb) j1 = nb;
    // map of block j = its maps_per_block consecutive (FIR-tile) maps composed in order (1 = maps are block maps)
    auto block_map = [&](long long j) {
        AgcMap bm;
        bm.A = 1.0; bm.B = 0.0;
        for (int q = 0; q < maps_per_block; q++) {
            const long long t = j * maps_per_block + q;
            if (t < n_maps) {
                const AgcMap m = maps[t];
                bm.B = m.A * bm.B + m.B;
                bm.A = m.A * bm.A;
            }
        }
        return bm;
    };
    double A = 1.0, Bc = 0.0;
    for (long long j = j0; j < j1; j++) {
        const AgcMap m = block_map(j);
        Bc = m.A * Bc + m.B;
        A = m.A * A;
    }
    sA[threadIdx.x] = A;
    sB[threadIdx.x] = Bc;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {            // inclusive Hillis-Steele scan of the slice maps
        double Ap = 1.0, Bp = 0.0;
        if ((int)threadIdx.x >= d) { Ap = sA[threadIdx.x - d]; Bp = sB[threadIdx.x - d]; }
        __syncthreads();
        if ((int)threadIdx.x >= d) {
            Bc = A * Bp + Bc;
            A = A * Ap;
            sA[threadIdx.x] = A;
            sB[threadIdx.x] = Bc;
        }
        __syncthreads();
    }
    // exclusive prefix of this thread's slice
    double Ae = 1.0, Be = 0.0;
    if (threadIdx.x > 0) { Ae = sA[threadIdx.x - 1]; Be = sB[threadIdx.x - 1]; }
    double g = Ae * (double)*norm + Be;
    for (long long j = j0; j < j1; j++) {
        double gg = g;
        if (!(gg > 1e-4)) gg = 1e-4;                // keep the model sane where the real AGC would clamp
        if (gg > 5000.0) gg = 5000.0;
        guesses[j] = gg;
        const AgcMap m = block_map(j);
        g = m.A * g + m.B;
    }
}

#define PDT_AGC_CKPT 1024     // samples between two gain checkpoints of a block (a multiple of 32)
template <typename T>
__device__ __forceinline__ void k_agc_block(const T *__restrict__ in, long long n, AgcParams<T> P,
                                                   const T *__restrict__ norm, long long B, long long W,
                                                   const double *__restrict__ guesses, const T *__restrict__ lock,
                                                   T *__restrict__ out, AgcSeam<T> *__restrict__ seams, double K,
                                                   T *__restrict__ ckpt /* the gain after every PDT_AGC_CKPT samples of a block
                                                                           (B / PDT_AGC_CKPT per block: k_agc_fix), or none */)
{
    __shared__ __attribute__((aligned(16))) unsigned char ring[PDT_AGC_PF * PDT_RING_SLOT];
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long start = j * B;
    if (start >= n) return;
    const long long end = (start + B < n) ? start + B : n;
    // The gain contracts with time constant tau ~ gain/decay samples.  Start from the affine-model
    // guess at a block boundary K time constants back (K = 14 for float, more bits to agree on in
    // double), never more than W samples; blocks whose warm-up would reach sample 0 replay from
    // the true initial gain.  Any choice is exact: seams are verified bitwise.
    long long ws = 0;
    T gain = *norm;
    if (j >= 1) {
        const double g_here = guesses[j];
        double need = K * g_here / (double)P.decay;
        if (need < 4096.0) need = 4096.0;
        if (need > (double)W) need = (double)W;
        long long m = (long long)((need + (double)B - 1.0) / (double)B);
        if (m < 1) m = 1;
        if (j - m >= 1) {
            ws = (j - m) * B;
            const double g0 = guesses[j - m];
            gain = (T)g0;
        }
    }
    agc_range<T, false>(in, lock, out, ws, start, gain, P, ring);
    AgcSeam<T> sm;
    sm.g0 = gain;
    if (ckpt) {
        // (round 4, the double-precision build: ARGOS' one open seam per capture -- gains equal to nine digits -- cost a whole
        // block walk of the slow f64 step, 0.46 of the step's 17.9 ms)
        const long long ncp = B / PDT_AGC_CKPT;
        long long at = start;
        for (long long c = 0; c < ncp && at + PDT_AGC_CKPT <= end; c++, at += PDT_AGC_CKPT) {
            agc_range<T, true>(in, lock, out, at, at + PDT_AGC_CKPT, gain, P, ring);
            ckpt[j * ncp + c] = gain;
        }
        agc_range<T, true>(in, lock, out, at, end, gain, P, ring);
    } else
        agc_range<T, true>(in, lock, out, start, end, gain, P, ring);
    sm.g1 = gain;
    seams[j] = sm;
}

// ---- round 4: the same walk with full-line transfers.
// What held k_agc_block up at an hour of 250 ksps (3.3 ms) was never its arithmetic (0.83 ms with its memory instructions
// taken out, tools/probes/agc_mem_probe.hip) but the shape of its accesses: every lane streams its own block, so a wavefront's
// 16-byte store touches 64 different lines, a sixth of each -- partial-line writes HBM takes at 2.1 TB/s (1.7 ms for the
// stores alone), and loads and stores of that shape do not overlap (1.4 + 1.7 = 3.2 ms).  Here a wavefront still owns 64
// consecutive blocks, one per lane, but samples move in SUPER-BATCHES of 32 per lane (128 B = one line per block), eight
// lanes to a line: transfer instruction s (of 8) carries the blocks l with (l & 7) == s; its lane t moves piece (t & 7) of
// block 8 (t >> 3) + s -- eight full lines per instruction -- to / from LDS slot s.  Slots are 1040 B apart, so that the
// walker lane l finds piece p of its own block at (l & 7) * 1040 + (l >> 3) * 128 + 16 p and all 16-byte LDS accesses, on
// the transfer face and on the walker face, are conflict-free.  Loads are LDS-direct, R super-batches ahead; outputs go
// through one staging super-batch.  Same arithmetic, same 16-sample calm batches: the outputs are those of agc_range bit
// for bit (10.8 GB in 1.84 ms = 5.9 TB/s in the probe; 1.47 ms with the tile-granular warm-up below).
#define PDT_TR_SLOT 1040
#define PDT_TR_SB (8 * PDT_TR_SLOT)
#ifndef PDT_AGC_TR_R
#define PDT_AGC_TR_R 4
#endif
typedef float pdt_f4 __attribute__((ext_vector_type(4)));

// walk nsb super-batches from element e0 (= the place of lane 0's block start + rel0 in the stream; every lane walks the
// same offsets of its own block).  STORE false = warm-up: a lane idles (keeps its gain) until rel reaches its own `from`.
// PARTIAL = the wavefront holds blocks that end at n (or lie behind it): every stored piece is checked against n.
// CLAMP = the walk reaches in front of the stream's first sample (long warm-ups of the first wavefronts): those lines are
// wanted by idle lanes only and are fetched from sample 0 on instead.
template <bool STORE, bool PARTIAL, bool CLAMP, int R>
__device__ __forceinline__ void agc_range_tr(const float *__restrict__ in, float *__restrict__ out, long long e0, int rel0, long long nsb,
                                             float &gain, int from, const AgcParams<float> &P, unsigned char *ring,
                                             const unsigned (&voff)[8], long long n, long long B, long long jb0,
                                             float *__restrict__ ckpt = nullptr)
{
    const int t = threadIdx.x & 63;
    const unsigned ring0 = (unsigned)(size_t)ring;
    const unsigned char *mine = ring + (t & 7) * PDT_TR_SLOT + (t >> 3) * 128;
    unsigned char *stage_mine = ring + R * PDT_TR_SB + (t & 7) * PDT_TR_SLOT + (t >> 3) * 128;
    const unsigned char *stage_row = ring + R * PDT_TR_SB + t * 16;
    const char *gin = (const char *)(in + e0);
    char *gout = (char *)(out + e0);
    auto issue = [&](long long sb, int slot) {
        const char *b = gin + sb * 128;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            unsigned vo = voff[s];
            if (CLAMP) {
                const long long e = e0 + sb * 32;                    // element the scalar base stands at (uniform)
                if (e + (long long)(vo >> 2) < 0) vo = (unsigned)((4 * (t & 7) - e) * 4);
            }
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(vo), "s"(b),
                         "s"(ring0 + slot * PDT_TR_SB + s * PDT_TR_SLOT) : "memory");
        }
    };
#pragma unroll
    for (int u = 0; u < R; u++) issue(u, u);         // (up to R super-batches past the range: slack of the device buffers)
    int slot = 0;
    int rel = rel0;
    for (long long sb = 0; sb < nsb; sb++, rel += 32) {
        // requests younger than this super-batch's loads: the loads of the R - 1 super-batches behind it (loads complete
        // in order among themselves; the counter also holds the stores, so this never waits for too little)
        ring_wait<(R - 1) * 8>();
        Vec16<float> x[8], y[8];
#pragma unroll
        for (int p = 0; p < 8; p++) x[p] = *reinterpret_cast<const Vec16<float> *>(mine + slot * PDT_TR_SB + p * 16);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float g = gain;
            if (agc_calm<float, 4>(x + 4 * h, g, P.decay)) {
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int w = 0; w < 4; w++) y[4 * h + k].v[w] = agc_step_calm(x[4 * h + k].v[w], g, P.decay);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int w = 0; w < 4; w++) y[4 * h + k].v[w] = agc_step(x[4 * h + k].v[w], g, P);
            }
            if (STORE && !PARTIAL) gain = g;
            else if (STORE) {
                // the wavefront that holds the stream's end: a lane's gain stops at its block's last sample inside the stream --
                // the state a stream segment hands to the next one (round 5: the per-lane walkers always stopped there; this
                // form walked every block to its nominal end, over whatever lies behind the stream, and nobody read that gain
                // while only whole captures came this way)
                const long long left = n - ((jb0 + t) * B + rel + 16 * h);       // samples of the stream from this half batch on
                if (left >= 16) gain = g;
                else if (left > 0) {
                    float g2 = gain;
#pragma unroll
                    for (int k = 0; k < 4; k++)
#pragma unroll
                        for (int w = 0; w < 4; w++)
                            if (4 * k + w < left) (void)agc_step(x[4 * h + k].v[w], g2, P);
                    gain = g2;
                }
            } else gain = (rel + 16 * h >= from) ? g : gain;
        }
        if (STORE) {
#pragma unroll
            for (int p = 0; p < 8; p++) *reinterpret_cast<Vec16<float> *>(stage_mine + p * 16) = y[p];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            char *ob = gout + sb * 128;
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const pdt_f4 v = *reinterpret_cast<const pdt_f4 *>(stage_row + s * PDT_TR_SLOT);
                if (!PARTIAL) {
                    asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(voff[s]), "v"(v), "s"(ob) : "memory");
                } else {
                    // (voff of a block behind the stream's end points at block 0: go by the block's real place)
                    const long long bstart = (jb0 + 8 * (t >> 3) + s) * B;
                    const long long pos = bstart + rel + 4 * (t & 7);
                    const long long lim = n - pos;                   // elements of the stream from this piece on
                    if (bstart < n) {
                        if (lim >= 4) *reinterpret_cast<pdt_f4 *>(out + pos) = v;
                        else
                            for (int w = 0; w < 4; w++)
                                if (w < lim) out[pos + w] = v[w];
                    }
                }
            }
        }
        // the gain after every PDT_AGC_CKPT samples of the block: a seam repair stops at the first one it reproduces
        if (STORE && ckpt && (sb & (PDT_AGC_CKPT / 32 - 1)) == PDT_AGC_CKPT / 32 - 1) ckpt[sb / (PDT_AGC_CKPT / 32)] = gain;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the slot's reads are done: it may be refilled
        issue(sb + R, slot);
        slot = (slot + 1 == R) ? 0 : slot + 1;
    }
    ring_wait<0>();
}

// Block-parallel AGC with full-line transfers (float, no Squelch, no raw copy; B a multiple of 32, streams on 128-byte
// boundaries).  Warm-ups start at a tile boundary (tile_len samples, a multiple of 32, maps_per_tile of the FIR kernel's
// affine maps to a tile, maps_per_block to a block) instead of a block boundary: the guess there is the block-boundary guess carried on through the tile maps in
// double, so a walker replays K time constants and not a whole block (9 984 instead of 28 288 samples at an hour of
// 250 ksps).  The lanes of a wavefront walk in step: the wavefront replays the longest warm-up among its lanes, a lane
// idles until its own begins.
template <int R>
__device__ __forceinline__ void k_agc_block_tr(const float *__restrict__ in, long long n, AgcParams<float> P,
                                               const float *__restrict__ norm, long long B, long long W,
                                               const double *__restrict__ guesses, const AgcMap *__restrict__ tmaps,
                                               int maps_per_block, int maps_per_tile, long long n_maps, long long tile_len,
                                               float *__restrict__ out, AgcSeam<float> *__restrict__ seams, double K,
                                               float *__restrict__ ckpt)
{
    __shared__ __attribute__((aligned(128))) unsigned char ring[(R + 1) * PDT_TR_SB];
    const int t = threadIdx.x & 63;
    const long long jb0 = (long long)blockIdx.x * 64;
    if (jb0 * B >= n) return;
    const long long j = jb0 + t;
    const long long start = j * B;
    const bool valid = start < n;
    unsigned voff[8];
#pragma unroll
    for (int s = 0; s < 8; s++) {
        // blocks of this wavefront that lie behind the stream's end transfer block 0's lines instead (never stored)
        long long b = 8 * (t >> 3) + s;
        if ((jb0 + b) * B >= n) b = 0;
        voff[s] = (unsigned)((b * B + 4 * (t & 7)) * 4);
    }
    // the lane's own warm-up: K time constants of the guessed gain, in whole tiles, never more than W samples
    long long ws = 0;
    float gain = *norm;
    if (valid && j >= 1) {
        const double g_here = guesses[j];
        double need = K * g_here / (double)P.decay;
        if (need < 4096.0) need = 4096.0;
        if (need > (double)W) need = (double)W;
        const long long m = (long long)((need + (double)tile_len - 1.0) / (double)tile_len);
        const long long q = (start - m * tile_len) / tile_len;       // tile the warm-up starts at (start is a tile boundary)
        if (start - m * tile_len > 0 && q >= 1) {
            ws = q * tile_len;
            const long long jb = ws / B;                             // the block boundary at or in front of it: its guess,
            double g = guesses[jb];                                  // carried on through the maps up to the tile
            for (long long u = jb * maps_per_block; u < q * maps_per_tile; u++)
                if (u < n_maps) {
                    const AgcMap mp = tmaps[u];
                    g = mp.A * g + mp.B;
                }
            if (!(g > 1e-4)) g = 1e-4;
            if (g > 5000.0) g = 5000.0;
            gain = (float)g;
        }
    }
    const bool partial = (jb0 + 64) * B > n;
    // longest warm-up of the wavefront (in samples before the block start; whole super-batches: starts and tiles are)
    int back = valid ? (int)(start - ws) : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int o = __shfl_xor(back, d);
        back = back > o ? back : o;
    }
    back = __builtin_amdgcn_readfirstlane(back);
    const int from = valid ? -(int)(start - ws) : 0x7fffffff;        // rel at which this lane's warm-up begins
    if (back > 0) {
        if (jb0 * B >= back) agc_range_tr<false, false, false, R>(in, out, jb0 * B - back, -back, back / 32, gain, from, P, ring, voff, n, B, jb0);
        else agc_range_tr<false, false, true, R>(in, out, jb0 * B - back, -back, back / 32, gain, from, P, ring, voff, n, B, jb0);
    }
    AgcSeam<float> sm;
    sm.g0 = gain;
    float *my_ckpt = (ckpt && valid) ? ckpt + j * (B / PDT_AGC_CKPT) : nullptr;
    if (!partial) agc_range_tr<true, false, false, R>(in, out, jb0 * B, 0, B / 32, gain, 0, P, ring, voff, n, B, jb0, my_ckpt);
    else agc_range_tr<true, true, false, R>(in, out, jb0 * B, 0, B / 32, gain, 0, P, ring, voff, n, B, jb0, my_ckpt);
    sm.g1 = gain;
    if (valid) seams[j] = sm;
}

// first seam that does not close (or the number of blocks): one workgroup, every thread a stride of seams.  Almost
// always there is none, and k_agc_fix then has nothing to scan (alone, its single wavefront took 64 seams per round trip
// to memory: 0.05 ms on 9 000 blocks).
template <typename T>
__device__ __forceinline__ void k_agc_scan(long long n, long long B, const AgcSeam<T> *__restrict__ seams,
                                                   long long *__restrict__ first_bad)
{
    __shared__ unsigned long long s_first;
    const long long nb = (n + B - 1) / B;
    if (threadIdx.x == 0) s_first = (unsigned long long)nb;
    __syncthreads();
    for (long long k = 1 + threadIdx.x; k < nb; k += blockDim.x)
        if (!bits_equal(seams[k - 1].g1, seams[k].g0)) {
            atomicMin(&s_first, (unsigned long long)k);
            break;
        }
    __syncthreads();
    if (threadIdx.x == 0) *first_bad = (long long)s_first;
}

template <typename T>
__device__ __forceinline__ void k_agc_fix(const T *__restrict__ in, long long n, AgcParams<T> P, long long B,
                                                 const T *__restrict__ lock, T *__restrict__ out,
                                                 AgcSeam<T> *__restrict__ seams, unsigned *__restrict__ counters,
                                                 const long long *__restrict__ first_bad, T *__restrict__ ckpt = nullptr)
{
    // same wave-parallel seam scan as k_pll_fix, from the first seam k_agc_scan found open
    __shared__ __attribute__((aligned(16))) unsigned char ring[PDT_AGC_PF * PDT_RING_SLOT];
    const long long nb = (n + B - 1) / B;
    unsigned fixes = 0;
    long long r = *first_bad;
    if (r < 1) r = 1;
    while (r < nb) {
        const long long mine = r + threadIdx.x;
        bool bad = false;
        if (mine < nb) bad = !bits_equal(seams[mine - 1].g1, seams[mine].g0);
        const unsigned long long mask = __ballot(bad);
        if (mask == 0) { r += 64; continue; }
        const long long rb = r + (long long)__builtin_ctzll(mask);
        fixes++;
        const T g_true = seams[rb - 1].g1;
        if (threadIdx.x == 0) {
            T gain = g_true;
            const long long start = rb * B;
            const long long end = (start + B < n) ? start + B : n;
            bool merged = false;
            if (ckpt) {
                // (k_agc_block_tr left the gain after every PDT_AGC_CKPT samples behind: the re-run and the stored walk differ by
                // the rounding noise of a start a few ulps off, so they agree again after a time constant or two -- from the
                // first checkpoint the re-run reproduces, the rest of the block, its end state included, stands as stored)
                const long long ncp = B / PDT_AGC_CKPT;
                T *cp = ckpt + rb * ncp;
                long long c = 0;
                for (; c < ncp && start + (c + 1) * PDT_AGC_CKPT <= end; c++) {
                    agc_range<T, true>(in, lock, out, start + c * PDT_AGC_CKPT, start + (c + 1) * PDT_AGC_CKPT, gain, P, ring);
                    if (bits_equal(cp[c], gain)) { merged = true; break; }
                    cp[c] = gain;
                }
                if (!merged) agc_range<T, true>(in, lock, out, start + c * PDT_AGC_CKPT, end, gain, P, ring);
            } else
            agc_range<T, true>(in, lock, out, start, end, gain, P, ring);
            seams[rb].g0 = g_true;
            if (!merged) seams[rb].g1 = gain;
        }
        __threadfence_block();
        __syncthreads();
        r = rb + 1;
    }
    if (threadIdx.x == 0) {
        counters[2] = (unsigned)nb;
        counters[3] = fixes;
    }
}

}  // namespace pdt
