// pdt_kernels_back.h -- back half of the chain on gfx950: Gardner symbol sampler,
// Manchester decision, sync-word search and frame extraction.
#pragma once
#include "pdt_device_math.h"

namespace pdt {

// one 16-byte vector of samples
template <typename T> struct alignas(16) Vec16 {
    static constexpr int N = 16 / sizeof(T);
    T v[N];
};

// ------------------------------------------------------------------------------------------
// Gardner clock recovery (reference: common/GardenerClockRecovery.c:5-114)
//
// The sampler is a leak-free integrator of a data-dependent, clipped error with
// nearest-sample picks: there is no contraction to lean on as in the PLL / AGC, and in double
// precision or odd chunk geometries this stage is a true sequential chain over the symbols of one
// capture -- one lane walks it (k_gardner, sequential mode); the other lanes of the workgroup only
// stage the current reference chunk (chunk-relative coordinates are part of the arithmetic: float
// spacing depends on the position inside the chunk) into LDS with coalesced loads.
// The float build normally goes through the exact parallel scheme further down (boundary-state
// tables), which uses the one regularity there is: trajectories that pick the same samples receive
// the same corrections and merge exactly.
// ------------------------------------------------------------------------------------------
template <typename T> struct GardnerParams {
    T step, kp, lim;
    long long n_total;       // interpolated samples in the capture
    long long chunk_out;     // reference chunk size in interpolated samples (chunk * interp)
    int argos_heap;          // 1 = reproduce the ARGOS heap adjacency (Q16) for reads past the chunk
    unsigned long long argos_field_bits;   // malloc size field seen as a double
    int argos_even;          // elements of malloc slack in front of that field (8-byte buffers: 1 when the chunk is even, else 0)
};

// value the reference would read at chunk-relative index idx >= n_cur of chunk c (Q3/Q16)
template <typename T>
__device__ __forceinline__ T gardner_beyond(const T *__restrict__ in, const T *__restrict__ lock, const GardnerParams<T> &P,
                                            long long c, long long n_cur, long long idx)
{
    const long long C = P.chunk_out;
    if (idx < C)                                   // stale data of the previous chunk (only in a short last chunk)
        return (c >= 1) ? in[(c - 1) * C + idx] : (T)0;
    if (!P.argos_heap) return (T)0;                // POES: over-allocated, never written -> zero pages
    long long k = idx - C;
    if (k < P.argos_even) return (T)0;             // malloc slack in front of the next chunk's size field (elements: 0 / 1 for
    k -= P.argos_even;                             // 8-byte buffers, 0 .. 3 for the float build's 4-byte ones)
    if (sizeof(T) == 8) {
        if (k == 0) return (T)__longlong_as_double((long long)P.argos_field_bits);
        k -= 1;
    } else {                                       // float build (the ARGOS sound-card twin): the 8-byte field is two elements
        if (k == 0) return (T)__uint_as_float((unsigned)(P.argos_field_bits & 0xffffffffull));
        if (k == 1) return (T)__uint_as_float((unsigned)(P.argos_field_bits >> 32));
        k -= 2;
    }
    // index into the lock-signal array of the current chunk
    if (k < n_cur) return lock[c * C + k];
    if (k < C) return (c >= 1) ? lock[(c - 1) * C + k] : (T)0;
    return (T)0;
}

// LDS budget (160 KiB per CU): input window + symbol staging buffers.
template <typename T> struct GardnerLds;
template <> struct GardnerLds<float> { static constexpr int LEN = 31232; static constexpr int OUT = 4096; };   // 122 + 32 KiB
template <> struct GardnerLds<double> { static constexpr int LEN = 14336; static constexpr int OUT = 3072; };  // 112 + 36 KiB

#define PDT_GARDNER_THREADS 64
#ifndef PDT_GEMIT_LEN
#define PDT_GEMIT_LEN 2048            // parallel emission: LDS window (samples) and symbol staging buffer per chunk
#endif
#ifndef PDT_GEMIT_OUT
#define PDT_GEMIT_OUT 256
#endif

template <typename T> __device__ __forceinline__ T uniform(T v);
template <> __device__ __forceinline__ int uniform<int>(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <> __device__ __forceinline__ unsigned uniform<unsigned>(unsigned v)
{
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

// neither NaN nor infinity
__device__ __forceinline__ bool is_finite_bits(float v) { return (__float_as_int(v) & 0x7f800000) != 0x7f800000; }
__device__ __forceinline__ bool is_finite_bits(double v) { return (__double2hiint(v) & 0x7ff00000) != 0x7ff00000; }
// rint(x) as an integer, read out of the mantissa of x + 1.5 * 2^(mantissa bits): the addition rounds to nearest-even at the
// integer position exactly as rint does (float: 0 <= x < 2^22; double: |x| < 2^31)
__device__ __forceinline__ int rint_index(float x) { return __float_as_int(x + 12582912.0f) - 0x4B400000; }
__device__ __forceinline__ int rint_index(double x) { return __double2loint(x + 6755399441055744.0); }
// (e > lim) ? lim : ((e < -lim) ? -lim : e) for every e that is not a NaN
__device__ __forceinline__ float clip_finite(float e, float lim) { return __builtin_amdgcn_fmed3f(e, -lim, lim); }
__device__ __forceinline__ double clip_finite(double e, double lim) { return __builtin_fmin(__builtin_fmax(e, -lim), lim); }

template <typename T> struct GardnerState {
    T ns, prev, half;      // sampler state (identical in every lane of the wavefront)
    T q_last;              // sampling instant of the last symbol after the error correction
    unsigned i_last;       // index the last symbol was taken at
};

// Walk ONE reference chunk with ONE wavefront.  All 64 lanes carry the same sampler state and
// execute the same instructions (LDS reads of one address broadcast), so every branch is
// wave-uniform and made scalar with readfirstlane: no exec-mask bookkeeping in the symbol
// loop.  The lanes differ only when they stage a window or flush symbols.
//
// Symbol loop = [one fully checked step] + [a counted batch of check-free steps].  The batch
// length is the number of steps that provably stay inside the chunk, the LDS window and the
// staging buffer: a step advances the sampling instant by at most step + 0.1.
// Returns the number of symbols of the chunk; EMIT stores them at sym[count0...].
template <typename T, bool EMIT, int LEN, int OUT>
__device__ __forceinline__ long long gardner_walk_chunk(const T *__restrict__ in, const T *__restrict__ lock,
                                                        const GardnerParams<T> &P, long long c, GardnerState<T> &S, T *win,
                                                        T *o_val, unsigned *o_idx, T *__restrict__ sym,
                                                        long long *__restrict__ symidx, long long count0, long long sym_cap)
{
    const int lane = threadIdx.x;
    const int NT = (int)blockDim.x;            // 64 in the per-chunk kernels; 256 in the sequential kernel, where every
                                               // wavefront walks the same trajectory and all of them stage and flush
    T ns = S.ns, prev = S.prev, half = S.half, q_last = S.q_last;
    unsigned i_last = S.i_last;
    const T hs = (T)((double)P.step / 2.0);       // exact: step/2 is representable
    const T kp = P.kp, lim = P.lim, step = P.step;
    const T adv = step + (T)0.101;                // upper bound of one step's advance
    const long long C = P.chunk_out;
    const long long base = c * C;
    const unsigned n_cur = (unsigned)((P.n_total - base < C) ? (P.n_total - base) : C);
    const T nT = (T)n_cur;
    long long count = count0;
    unsigned wbase = 0;
    bool chunk_done = false;
    while (!chunk_done) {
        // ---- stage the part of [wbase, wbase+LEN) the sampler can touch: chunk data with 8
        // independent loads in flight per lane, then the few past-the-end values (Q3/Q16)
        __syncthreads();
        int n_staged;
        {
            const unsigned avail = (n_cur > wbase) ? n_cur - wbase : 0u;
            const int n_data = (int)((avail < (unsigned)LEN) ? avail : (unsigned)LEN);
            const T *src = in + base + wbase;
            int t = lane;
            for (; t + 7 * NT < n_data; t += 8 * NT) {
                T r[8];
#pragma unroll
                for (int u = 0; u < 8; u++) r[u] = src[t + u * NT];
#pragma unroll
                for (int u = 0; u < 8; u++) win[t + u * NT] = r[u];
            }
            for (; t < n_data; t += NT) win[t] = src[t];
            int n_tail = n_data + 2 * (int)step + 24;          // furthest look-ahead of the mid-point
            if (n_tail > LEN) n_tail = LEN;
            for (int q = n_data + lane; q < n_tail; q += NT)
                win[q] = gardner_beyond(in, lock, P, c, (long long)n_cur, (long long)(wbase + (unsigned)q));
            n_staged = n_tail;
        }
        __syncthreads();
        const unsigned wend = wbase + (unsigned)LEN;
        const unsigned lim_idx = (n_cur < wend) ? n_cur : wend;   // first index the batch must not reach
        int nout = 0;
        for (;;) {
            // ---- checked step
            const T rn = Real<T>::rint(ns);
            const int in_chunk = uniform<int>((int)(rn < nT));
            if (!in_chunk) { chunk_done = true; break; }
            const unsigned i_abs = uniform<unsigned>((unsigned)rn);
            if (i_abs - wbase >= (unsigned)LEN || nout >= OUT) break;          // new window / flush
            const unsigned h_abs = uniform<unsigned>((unsigned)Real<T>::rint(half));
            const T cur = win[i_abs - wbase];
            T mid;
            if (h_abs - wbase < (unsigned)n_staged)
                mid = win[h_abs - wbase];
            else
                mid = (h_abs < n_cur) ? in[base + h_abs] : gardner_beyond(in, lock, P, c, (long long)n_cur, (long long)h_abs);
            if (EMIT) {
                o_val[nout] = cur;
                o_idx[nout] = i_abs;
            }
            nout++;
            {
                T err = kp * (cur - prev) * mid;
                err = (err > lim) ? lim : ((err < -lim) ? -lim : err);
                ns = ns - err;
                q_last = ns;
                half = ns + hs;   // == (T)((double)ns + (double)step/2.0): that double sum is exact
                ns = ns + step;
                prev = cur;
                i_last = i_abs;
            }
            // ---- check-free batch
            const T room = (T)lim_idx - (T)2 - ns;
            int K = (room > (T)0) ? (int)(room / adv) : 0;
            K = uniform<int>(K);
            if (K > OUT - nout) K = OUT - nout;
            const T wb = (T)wbase;
            T *ov = o_val + nout;
            unsigned *oi = o_idx + nout;
            int k = 0;
            if constexpr (sizeof(T) == 4) if (n_cur < (1u << 22) - 4096u) {
                // float: the wavefronts of a capture's chunks fill the chip's issue slots (about one instruction per four
                // clocks per wavefront), so the loop is written for instruction count: round-to-nearest-even through the
                // 1.5 * 2^23 bias (index = mantissa bits, exact for 0 <= x < 2^22, as in the table kernel), the clip as one
                // median (the value the two-sided select gives for every non-NaN error), four symbols per trip
                const int bias = 0x4B400000 + (int)wbase;
                auto one = [&](int kk) {
                    const int bn = __float_as_int((float)ns + 12582912.0f);
                    const int bh = __float_as_int((float)half + 12582912.0f);
                    const float c_k = (float)win[bn - bias];
                    const float m_k = (float)win[bh - bias];
                    i_last = (unsigned)(bn - 0x4B400000);
                    if (EMIT) {
                        ov[kk] = (T)c_k;
                        oi[kk] = i_last;
                    }
                    const float err = __builtin_amdgcn_fmed3f((float)kp * (c_k - (float)prev) * m_k, -(float)lim, (float)lim);
                    ns = (T)((float)ns - err);
                    q_last = ns;
                    half = (T)((float)ns + (float)hs);
                    ns = (T)((float)ns + (float)step);
                    prev = (T)c_k;
                };
                for (; k + 4 <= K; k += 4) {
                    one(k);
                    one(k + 1);
                    one(k + 2);
                    one(k + 3);
                }
                for (; k < K; k++) one(k);
            }
            for (; k < K; k++) {
                const T rnk = Real<T>::rint(ns);
                const T rhk = Real<T>::rint(half);
                const T c_k = win[(int)(rnk - wb)];
                const T m_k = win[(int)(rhk - wb)];
                i_last = (unsigned)rnk;
                if (EMIT) {
                    ov[k] = c_k;
                    oi[k] = i_last;
                }
                T err = kp * (c_k - prev) * m_k;
                err = (err > lim) ? lim : ((err < -lim) ? -lim : err);
                ns = ns - err;
                q_last = ns;
                half = ns + hs;
                ns = ns + step;
                prev = c_k;
            }
            nout += K;
        }
        // ---- flush staged symbols with coalesced stores
        if (EMIT) {
            __syncthreads();
            for (int t = lane; t < nout; t += NT) {
                const long long k = count + t;
                if (k < sym_cap) {
                    sym[k] = o_val[t];
                    symidx[k] = base + (long long)o_idx[t];
                }
            }
        }
        count += nout;
        if (!chunk_done) {
            const unsigned cur_i = uniform<unsigned>((unsigned)Real<T>::rint(ns));
            if (cur_i - wbase >= (unsigned)LEN) wbase = (cur_i > 64u) ? cur_i - 64u : 0u;   // keep the mid-point in view
        }
    }
    S.ns = ns - nT;                                // roll over; `half` is deliberately not (Q3)
    S.prev = prev;
    S.half = half;
    S.q_last = q_last;
    S.i_last = i_last;
    return count - count0;
}

// entry state of one reference chunk, produced by k_gardner_chain for the parallel path
template <typename T> struct GardnerEntry {
    T ns, prev, half;
    long long offset;        // symbols emitted before this chunk
};

// State a sequential sampler is entered with / leaves behind when a stream is demodulated segment by segment: the sampler
// state after the last full chunk before c_first, and the number of symbols that already sit in the symbol buffer (the two
// history symbols the Manchester stage needs).  A whole capture is {0, 0, 0, 0, 0}.
template <typename T> struct SamplerCarry {
    T a, b, c;               // Gardner: nextSample, prev, halfSample; M&M: nextSample, stepSize, sampleLast
    long long c_first;       // first chunk to walk
    long long count0;        // symbols already in the buffer: output starts there, the count includes them
};

// sequential mode (entries == nullptr): one wavefront walks the chunks from `carry.c_first` on, in order.
// parallel mode: block b owns chunk b and starts from the tabulated entry state.
template <typename T, int LEN, int OUT>
__device__ __forceinline__ void k_gardner(const T *__restrict__ in, const T *__restrict__ lock,
                                                                  GardnerParams<T> P, T *__restrict__ sym,
                                                                  long long *__restrict__ symidx,
                                                                  unsigned long long *__restrict__ nsym_out,
                                                                  long long sym_cap,
                                                                  const GardnerEntry<T> *__restrict__ entries,
                                                                  SamplerCarry<T> carry, SamplerCarry<T> *__restrict__ carry_out,
                                                                  long long c_off /* parallel mode: group of block 0 */,
                                                                  int span /* parallel mode: chunks per group (entries are per group); otherwise 1 */,
                                                                  const unsigned char *__restrict__ flags /* parallel mode, optional: only the groups marked 1 */)
{
    __shared__ T win[LEN];
    __shared__ T o_val[OUT];
    __shared__ unsigned o_idx[OUT];
    const long long C = P.chunk_out;
    const long long n_chunks = (P.n_total + C - 1) / C;
    GardnerState<T> S;
    S.ns = carry.a; S.prev = carry.b; S.half = carry.c; S.q_last = 0; S.i_last = 0;
    long long count = carry.count0;
    long long c_begin = carry.c_first, c_end = n_chunks;
    if (entries) {
        const long long g = blockIdx.x + c_off;
        if (flags && !flags[g]) return;                                      // (emitted chunk by chunk: k_gardner_emit_first / _rest)
        c_begin = g * span;
        c_end = (c_begin + span < n_chunks) ? c_begin + span : n_chunks;
        if (c_begin >= n_chunks || c_begin < carry.c_first) return;      // (a stream segment: the chunks in front are history)
        const GardnerEntry<T> e = entries[g];
        S.ns = e.ns;
        S.prev = e.prev;
        S.half = e.half;
        count = e.offset;
    }
    for (long long c = c_begin; c < c_end; c++)
        count += gardner_walk_chunk<T, true, LEN, OUT>(in, lock, P, c, S, win, o_val, o_idx, sym, symidx, count, sym_cap);
    if (threadIdx.x == 0 && (!entries || c_end == n_chunks)) *nsym_out = (unsigned long long)count;
    if (threadIdx.x == 0 && carry_out && (!entries || c_end == n_chunks)) {
        SamplerCarry<T> o;
        o.a = S.ns; o.b = S.prev; o.c = S.half; o.c_first = c_end; o.count0 = count;
        *carry_out = o;
    }
}

// Sequential sampler for small reference chunks (ARGOS: 2 400 samples, 60 symbols per chunk): with one window per
// chunk the walk of a chunk takes less time than staging it, so the staging is taken off the walker's
// path -- wavefronts 1..3 fill the second LDS buffer with chunk c+1 (data + the past-the-end values of
// Q3/Q16) while wavefront 0 walks chunk c out of the first, then the buffers swap.  The walk is the
// single-window case of gardner_walk_chunk, statement for statement.
template <typename T, int LEN, int OUT>
__device__ __forceinline__ void k_gardner_small(const T *__restrict__ in, const T *__restrict__ lock, GardnerParams<T> P,
                                                        T *__restrict__ sym, long long *__restrict__ symidx,
                                                        unsigned long long *__restrict__ nsym_out, long long sym_cap)
{
    __shared__ T win[2][LEN];
    __shared__ T o_val[OUT];
    __shared__ unsigned o_idx[OUT];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const long long C = P.chunk_out;
    const long long n_chunks = (P.n_total + C - 1) / C;
    const T hs = (T)((double)P.step / 2.0);
    const T kp = P.kp, lim = P.lim, step = P.step;
    const T adv = step + (T)0.101;
    const int tail = 2 * (int)step + 24;               // furthest look-ahead of the mid-point
    T ns = 0, prev = 0, half = 0;
    long long count = 0;

    // squelched stretches (ARGOS: three quarters of a capture) are exact zeros: there cur - prev = 0, the error term is +-0
    // whatever the mid-point sample is, and the sampler just adds its step -- such chunks take a loop without the error
    // arithmetic.  s_nz[b] = the staged chunk in buffer b holds a non-zero sample.
    __shared__ int s_nz[2];
    auto stage = [&](long long c, T *buf, int first, int stride) {
        const long long base = c * C;
        const int n_cur = (int)((P.n_total - base < C) ? (P.n_total - base) : C);
        int nz = 0;
        int t = first;
        for (; t + 7 * stride < n_cur; t += 8 * stride) {            // eight loads in flight per thread: one memory latency per chunk
            T r[8];
#pragma unroll
            for (int u = 0; u < 8; u++) r[u] = in[base + t + u * stride];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                nz |= (r[u] != (T)0) ? 1 : 0;
                buf[t + u * stride] = r[u];
            }
        }
        for (; t < n_cur; t += stride) {
            const T v = in[base + t];
            nz |= (v != (T)0) ? 1 : 0;
            buf[t] = v;
        }
        if (nz) atomicOr(&s_nz[c & 1], 1);
        int n_tail = n_cur + tail;
        if (n_tail > LEN) n_tail = LEN;
        for (int q = n_cur + first; q < n_tail; q += stride)
            buf[q] = gardner_beyond(in, lock, P, c, (long long)n_cur, (long long)q);
    };
    if (tid < 2) s_nz[tid] = 0;
    __syncthreads();
    if (n_chunks > 0) stage(0, win[0], tid, 256);
    __syncthreads();
    for (long long c = 0; c < n_chunks; c++) {
        const T *w = win[c & 1];
        const long long base = c * C;
        const int n_cur = (int)((P.n_total - base < C) ? (P.n_total - base) : C);
        int n_staged = n_cur + tail;
        if (n_staged > LEN) n_staged = LEN;
        int nout = 0;
        const bool calm = s_nz[c & 1] == 0;       // (read by every wavefront before the barrier at the end of this iteration)
        if (wave != 0) {
            if (c + 1 < n_chunks) stage(c + 1, win[(c + 1) & 1], tid - 64, 192);
        } else {
            const T nT = (T)n_cur;
            if (calm && prev == (T)0) {
                // every sample of the chunk is zero and so is the previous symbol: err = clip(kp * (0 - 0) * mid) = +-0,
                // nextSample - err = nextSample: only the additions of the reference step remain (GardenerClockRecovery.c:52-63)
                for (;;) {
                    const T rn = Real<T>::rint(ns);
                    const int in_chunk = uniform<int>((int)(rn < nT));
                    if (!in_chunk || nout >= OUT) break;
                    const unsigned i_abs = uniform<unsigned>((unsigned)rn);
                    o_val[nout] = w[i_abs];
                    o_idx[nout] = i_abs;
                    nout++;
                    half = ns + hs;
                    ns = ns + step;
                }
                prev = (nout > 0) ? w[o_idx[nout - 1]] : prev;
            } else
            for (;;) {
                // ---- checked step
                const T rn = Real<T>::rint(ns);
                const int in_chunk = uniform<int>((int)(rn < nT));
                if (!in_chunk) break;
                if (nout >= OUT) break;                                      // cannot happen: OUT covers a whole chunk
                const unsigned i_abs = uniform<unsigned>((unsigned)rn);
                const unsigned h_abs = uniform<unsigned>((unsigned)Real<T>::rint(half));
                const T cur = w[i_abs];
                T mid;
                if (h_abs < (unsigned)n_staged) mid = w[h_abs];
                else mid = (h_abs < (unsigned)n_cur) ? in[base + h_abs] : gardner_beyond(in, lock, P, c, (long long)n_cur, (long long)h_abs);
                o_val[nout] = cur;
                o_idx[nout] = i_abs;
                nout++;
                {
                    T err = kp * (cur - prev) * mid;
                    err = (err > lim) ? lim : ((err < -lim) ? -lim : err);
                    ns = ns - err;
                    half = ns + hs;
                    ns = ns + step;
                    prev = cur;
                }
                // ---- check-free batch
                const T room = (T)n_cur - (T)2 - ns;
                int K = (room > (T)0) ? (int)(room / adv) : 0;
                K = uniform<int>(K);
                if (K > OUT - nout) K = OUT - nout;
                for (int k = 0; k < K; k++) {
                    const T rnk = Real<T>::rint(ns);
                    const T rhk = Real<T>::rint(half);
                    const T c_k = w[(int)rnk];
                    const T m_k = w[(int)rhk];
                    o_val[nout + k] = c_k;
                    o_idx[nout + k] = (unsigned)rnk;
                    T err = kp * (c_k - prev) * m_k;
                    err = (err > lim) ? lim : ((err < -lim) ? -lim : err);
                    ns = ns - err;
                    half = ns + hs;
                    ns = ns + step;
                    prev = c_k;
                }
                nout += K;
            }
            // the walker flushes its own symbols (at most a few per lane)
            for (int t = lane; t < nout; t += 64) {
                const long long k = count + t;
                if (k < sym_cap) {
                    sym[k] = o_val[t];
                    symidx[k] = base + (long long)o_idx[t];
                }
            }
            ns = ns - nT;                              // roll over; `half` is deliberately not (Q3)
        }
        count += uniform<int>(nout);                   // only wavefront 0's copy matters
        __syncthreads();
        if (tid == 0) s_nz[c & 1] = 0;                 // this buffer is restaged (chunk c + 2) in the next iteration
        __syncthreads();
    }
    if (tid == 0) *nsym_out = (unsigned long long)count;
}

// need[c] = 1 when chunk c has to be walked sample by sample: a sample of chunk c or of chunk c - 1 is not +0.0.  In every
// other chunk all picks are +0.0 and so is the symbol before its first one: cur - prev = 0, the error term is +-0 whatever the
// mid-point sample is (finite: AGC output, lock signal or the heap's size field), nextSample - err = nextSample
// (GardenerClockRecovery.c:52-63) -- the sampler only adds its step.  One workgroup per chunk.
template <typename T>
__device__ __forceinline__ void k_chunk_need(const T *__restrict__ in, long long n_total, long long C, long long n_chunks,
                                             unsigned char *__restrict__ need)
{
    const long long c = blockIdx.x;
    if (c >= n_chunks) return;
    long long i0 = (c > 0 ? c - 1 : 0) * C, i1 = (c + 1) * C;
    if (i1 > n_total) i1 = n_total;
    int nz = 0;
    for (long long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const T v = in[i];
        if constexpr (sizeof(T) == 8) nz |= (__double_as_longlong((double)v) != 0ll) ? 1 : 0;
        else nz |= (__float_as_int((float)v) != 0) ? 1 : 0;
    }
    nz = __syncthreads_or(nz);
    if (threadIdx.x == 0) need[c] = nz ? 1 : 0;
}

// where a chunk without a walk starts: sampling instant (chunk-relative) and number of symbols before it
template <typename T> struct CalmEntry {
    T ns;
    long long count;
};

// symbols of the chunks the ring sampler did not walk: value +0.0 at rint(nextSample), nextSample advancing by the step
// from the entry the walker noted.  One lane per chunk.
template <typename T>
__device__ __forceinline__ void k_calm_emit(const unsigned char *__restrict__ need, const CalmEntry<T> *__restrict__ calm,
                                            GardnerParams<T> P, T *__restrict__ sym, long long *__restrict__ symidx,
                                            long long sym_cap)
{
    const long long C = P.chunk_out;
    const long long n_chunks = (P.n_total + C - 1) / C;
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks || need[c] != 0) return;
    const long long base = c * C;
    const T nT = (T)((P.n_total - base < C) ? (P.n_total - base) : C);
    T ns = calm[c].ns;
    long long k = calm[c].count;
    for (;;) {
        const T rn = Real<T>::rint(ns);
        if (!(rn < nT)) break;
        if (k < sym_cap) {
            sym[k] = (T)0;
            symidx[k] = base + (long long)(unsigned)rn;
        }
        k++;
        ns = ns + P.step;
    }
}

// Sequential sampler for small reference chunks with a ring of NB staged chunks (ARGOS: 2 400 samples, 60 symbols per
// chunk, 70 % of them squelched).  Wavefront 0 walks; wavefront 1 + s stages every NB-th chunk that needs a walk
// (k_chunk_need) into buffer s, several chunks ahead of the walker, so that the memory latency of a chunk (the whole cost of
// the two-buffer kernel above: 6.7 us per chunk) is hidden behind NB walks; chunks that need no walk are never read.
// Hand-over through two counters per buffer in LDS (uses staged / uses consumed), release / acquire at workgroup scope.
template <typename T, int LEN, int NB, int OUT>
__device__ __forceinline__ void k_gardner_ring(const T *__restrict__ in, const T *__restrict__ lock, GardnerParams<T> P,
                                               const unsigned char *__restrict__ need, T *__restrict__ sym,
                                               long long *__restrict__ symidx, unsigned long long *__restrict__ nsym_out,
                                               long long sym_cap, CalmEntry<T> *__restrict__ calm, T gran_scale)
{
    __shared__ __align__(16) T win[NB][LEN];
    __shared__ T o_val[OUT];
    __shared__ unsigned o_idx[OUT];
    __shared__ int s_ready[NB], s_done[NB], s_bad[NB];      // s_bad: the staged chunk holds a NaN or an infinity
    constexpr int VN = 16 / (int)sizeof(T);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const long long C = P.chunk_out;
    const long long n_chunks = (P.n_total + C - 1) / C;
    const int tail = 2 * (int)P.step + 24;               // furthest look-ahead of the mid-point
    if (tid < NB) { s_ready[tid] = 0; s_done[tid] = 0; }
    __syncthreads();
    if (wave > NB) return;

    if (wave != 0) {
        // ---- stager of buffer s
        const int s = wave - 1;
        T *buf = win[s];
        int j = 0;                                       // chunks in need of a walk seen so far
        for (long long c0 = 0; c0 < n_chunks; c0 += 64) {
            unsigned long long m = __ballot(c0 + lane < n_chunks && need[c0 + lane] != 0);
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const int mine = (j % NB) == s;
                const int use = j / NB;
                j++;
                if (!mine) continue;
                while (__hip_atomic_load(&s_done[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != use) __builtin_amdgcn_s_sleep(4);
                const long long c = c0 + b;
                const long long base = c * C;
                const int n_cur = (int)((P.n_total - base < C) ? (P.n_total - base) : C);
                int bad = 0;
                auto put = [&](T *dst, T v) {
                    bad |= is_finite_bits(v) ? 0 : 1;
                    *dst = v;
                };
                if ((C % VN) == 0) {
                    const int nv = n_cur / VN;                                     // whole 16-byte vectors, all in flight together
                    const Vec16<T> *src = reinterpret_cast<const Vec16<T> *>(in + base);
                    Vec16<T> *dst = reinterpret_cast<Vec16<T> *>(buf);
                    int t = lane;
                    for (; t + 9 * 64 < nv; t += 10 * 64) {
                        Vec16<T> r[10];
#pragma unroll
                        for (int u = 0; u < 10; u++) r[u] = src[t + u * 64];
#pragma unroll
                        for (int u = 0; u < 10; u++) {
#pragma unroll
                            for (int e = 0; e < VN; e++) bad |= is_finite_bits(r[u].v[e]) ? 0 : 1;
                            dst[t + u * 64] = r[u];
                        }
                    }
                    for (; t < nv; t += 64) {
                        const Vec16<T> r = src[t];
#pragma unroll
                        for (int e = 0; e < VN; e++) bad |= is_finite_bits(r.v[e]) ? 0 : 1;
                        dst[t] = r;
                    }
                    for (int q = nv * VN + lane; q < n_cur; q += 64) put(buf + q, in[base + q]);
                } else {
                    for (int q = lane; q < n_cur; q += 64) put(buf + q, in[base + q]);
                }
                int n_tail = n_cur + tail;
                if (n_tail > LEN) n_tail = LEN;
                for (int q = n_cur + lane; q < n_tail; q += 64)
                    put(buf + q, gardner_beyond(in, lock, P, c, (long long)n_cur, (long long)q));
                if (lane == 0) s_bad[s] = 0;
                if (bad) s_bad[s] = 1;
                __hip_atomic_store(&s_ready[s], use + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        return;
    }

    // ---- the walker
    __builtin_amdgcn_s_setprio(3);
    const T hs = (T)((double)P.step / 2.0);
    const T kp = P.kp, lim = P.lim, step = P.step;
    const T adv = step + (T)0.101;
    T ns = 0, prev = 0, half = 0;
    T memo_in = (T)-1e30, memo_nT = 0, memo_half = 0, memo_out = 0;   // the last chunk taken in one stride: entry -> picks, exit
    int memo_K = 0;
    long long count = 0;
    int jn = 0;                                          // walked chunks so far
    for (long long c0 = 0; c0 < n_chunks; c0 += 64) {
        const unsigned long long mask = __ballot(c0 + lane < n_chunks && need[c0 + lane] != 0);
        const int nb = (int)((n_chunks - c0 < 64) ? (n_chunks - c0) : 64);
        for (int b = 0; b < nb; b++) {
            const long long c = c0 + b;
            const long long base = c * C;
            const int n_cur = (int)((P.n_total - base < C) ? (P.n_total - base) : C);
            const T nT = (T)n_cur;
            int nout = 0;
            if (!((mask >> b) & 1ull)) {
                // only the additions of the reference step remain; every pick is +0.0.  The walker notes where the chunk
                // starts (sampling instant, symbol count); k_calm_emit writes its symbols afterwards, all chunks in parallel
                if (lane == 0) {
                    CalmEntry<T> e;
                    e.ns = ns;
                    e.count = count;
                    calm[c] = e;
                }
                // Round 5: the chunk in one stride where every one of those additions is exact.  gran_scale = 2^(p - E), p the
                // mantissa's length and 2^E above chunk + 2 steps (the host passes 0 unless the step is a multiple of 2^(E - p)):
                // a sampling instant that is a multiple of 2^(E - p) itself -- as it is after every roll-over from the top binade
                // -- stays one through every + step, all of them below 2^E: ns + k step is the k-th sum bit for bit, and the
                // number of picks is the K with rint(ns + (K - 1) step) < n <= rint(ns + K step), found from a quotient and
                // settled with the reference's own test (GardenerClockRecovery.c:25).  (60 dependent additions a chunk, 16 clocks
                // each, were a fifth of the ARGOS sampler's time.)
                // ... and not at all where the last such chunk started from the same instant (a pure function of it and of the
                // chunk's length): step 40.0 and chunks of 2 400 samples bring the instant back to where it was, gap after gap
                const T gs = ns * gran_scale;
                if (uniform<int>((int)(ns == memo_in && nT == memo_nT))) {
                    nout = memo_K;
                    if (nout > 0) { half = memo_half; ns = memo_out; }
                } else if (uniform<int>((int)(gran_scale != (T)0 && gs == Real<T>::rint(gs) && Real<T>::abs(gs) < (T)(sizeof(T) == 8 ? 4503599627370496.0 : 8388608.0)))) {
                    const T ns_in = ns;
                    const T q = (nT - (T)0.5 - ns) / step;
                    int K = uniform<int>((q > (T)0) ? (int)q : 0);
                    while (K > 0 && !uniform<int>((int)(Real<T>::rint(ns + (T)(K - 1) * step) < nT))) K--;
                    while (uniform<int>((int)(Real<T>::rint(ns + (T)K * step) < nT))) K++;
                    if (K > 0) {
                        half = (ns + (T)(K - 1) * step) + hs;
                        ns = ns + (T)K * step;
                        nout = K;
                    }
                    memo_in = ns_in; memo_nT = nT; memo_K = K; memo_half = half; memo_out = ns;
                } else
                for (;;) {
                    const T rn = Real<T>::rint(ns);
                    const int in_chunk = uniform<int>((int)(rn < nT));
                    if (!in_chunk) break;
                    nout++;
                    T before = ns;
                    ns = ns + step;
                    // check-free batch: K more steps stay inside the chunk
                    const T room = nT - (T)2 - ns;
                    int K = (room > (T)0) ? (int)(room / adv) : 0;
                    K = uniform<int>(K);
                    if (K > 0) {
                        int k = 0;
                        for (; k + 8 < K; k += 8) {
#pragma unroll
                            for (int u = 0; u < 8; u++) ns = ns + step;
                        }
                        for (; k + 1 < K; k++) ns = ns + step;
                        before = ns;
                        ns = ns + step;
                        nout += K;
                    }
                    half = before + hs;
                }
                prev = (nout > 0) ? (T)0 : prev;
            } else {
                const int s = jn % NB, use = jn / NB;
                jn++;
                while (__hip_atomic_load(&s_ready[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != use + 1) __builtin_amdgcn_s_sleep(1);
                const T *w = win[s];
                const bool exact = s_bad[s] != 0 || n_cur >= (1 << 22) - 4096;
                int n_staged = n_cur + tail;
                if (n_staged > LEN) n_staged = LEN;
                for (;;) {
                    // ---- checked step
                    const T rn = Real<T>::rint(ns);
                    const int in_chunk = uniform<int>((int)(rn < nT));
                    if (!in_chunk) break;
                    if (nout >= OUT) break;                                      // cannot happen: OUT covers a whole chunk
                    const unsigned i_abs = uniform<unsigned>((unsigned)rn);
                    const unsigned h_abs = uniform<unsigned>((unsigned)Real<T>::rint(half));
                    const T cur = w[i_abs];
                    T mid;
                    if (h_abs < (unsigned)n_staged) mid = w[h_abs];
                    else mid = (h_abs < (unsigned)n_cur) ? in[base + h_abs] : gardner_beyond(in, lock, P, c, (long long)n_cur, (long long)h_abs);
                    o_val[nout] = cur;
                    o_idx[nout] = i_abs;
                    nout++;
                    {
                        T err = kp * (cur - prev) * mid;
                        err = (err > lim) ? lim : ((err < -lim) ? -lim : err);
                        ns = ns - err;
                        half = ns + hs;
                        ns = ns + step;
                        prev = cur;
                    }
                    // ---- check-free batch
                    const T room = (T)n_cur - (T)2 - ns;
                    int K = (room > (T)0) ? (int)(room / adv) : 0;
                    K = uniform<int>(K);
                    if (K > OUT - nout) K = OUT - nout;
                    T *ov = o_val + nout;
                    unsigned *oi = o_idx + nout;
                    int k = 0;
                    if (!exact) {
                        // A lone wavefront pays the full latency of every dependent instruction (16 clocks for a double-precision
                        // add or multiply, ~50 for the LDS read; 190 per symbol): the loop is written for the length of the
                        // dependent chain.  The rounded index comes out of the mantissa (rint_index: one add instead of round +
                        // convert); the clip is max / min (the value the reference's two-sided select gives for every non-NaN
                        // error; the stager has looked at every value of this buffer); four symbols per trip with immediate
                        // store offsets.  Tried and slower: both samples of step k + 1 fetched by the 64 lanes during step k and
                        // picked with v_readlane (readfirstlane / readlane and their wait states cost more than the LDS latency
                        // they replace: 102 vs 80 ns per symbol); the clip folded into the update as a select between
                        // nextSample - e, nextSample - lim, nextSample + lim (two f64 compares: 96 ns); round 5: the three
                        // candidates of the next pick and of the next mid-point (the clipped error moves either by less than one
                        // sample) read a step ahead and picked with two compares each -- the compiler issues the reads next to
                        // their use, and even issued early the compare / select pairs cost what the LDS read did: 133 ns.
                        auto one = [&](int kk) {
                            const int ic = rint_index(ns), ih = rint_index(half);
                            const T c_k = w[ic], m_k = w[ih];
                            ov[kk] = c_k;
                            oi[kk] = (unsigned)ic;
                            const T err = clip_finite(kp * (c_k - prev) * m_k, lim);
                            ns = ns - err;
                            half = ns + hs;
                            ns = ns + step;
                            prev = c_k;
                        };
                        for (; k + 4 <= K; k += 4) {
                            one(k);
                            one(k + 1);
                            one(k + 2);
                            one(k + 3);
                        }
                    }
                    for (; k < K; k++) {
                        const T rnk = Real<T>::rint(ns);
                        const T rhk = Real<T>::rint(half);
                        const T c_k = w[(int)rnk];
                        const T m_k = w[(int)rhk];
                        ov[k] = c_k;
                        oi[k] = (unsigned)rnk;
                        T err = kp * (c_k - prev) * m_k;
                        err = (err > lim) ? lim : ((err < -lim) ? -lim : err);
                        ns = ns - err;
                        half = ns + hs;
                        ns = ns + step;
                        prev = c_k;
                    }
                    nout += K;
                }
                __hip_atomic_store(&s_done[s], use + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                for (int t = lane; t < nout; t += 64) {
                    const long long k = count + t;
                    if (k < sym_cap) {
                        sym[k] = o_val[t];
                        symidx[k] = base + (long long)o_idx[t];
                    }
                }
            }
            count += nout;
            ns = ns - nT;                              // roll over; `half` is deliberately not (Q3)
        }
    }
    if (lane == 0) *nsym_out = (unsigned long long)count;
}

// ------------------------------------------------------------------------------------------
// Mueller & Muller clock recovery (reference: common/MMClockRecovery.c:5-83), the alternative sampler
// the reference keeps at the same call site behind a comment (ARGOSdemod/main.c:277).  State
// (nextSample, stepSize, sampleLast) has two free floats, so there is no small boundary-state domain to
// tabulate: one wavefront walks the capture, chunk by chunk, like the sequential Gardner kernel.
// ------------------------------------------------------------------------------------------
template <typename T> struct MmParams {
    T step0, step_max, step_min, kp;   // Fs/baud, Fs/(baud - stepRange), Fs/(baud + stepRange)
    long long n_total, chunk_out;
};
// the reference rounds the sampling instant with rint() in the float build and with rintf() -- after
// narrowing to float -- in the double build (MMClockRecovery.c:25 vs :55)
template <typename T> __device__ __forceinline__ T mm_rint(T x);
template <> __device__ __forceinline__ float mm_rint<float>(float x) { return __builtin_rintf(x); }
template <> __device__ __forceinline__ double mm_rint<double>(double x) { return (double)__builtin_rintf((float)x); }

template <typename T, int LEN, int OUT>
__device__ __forceinline__ void k_mm(const T *__restrict__ in, MmParams<T> P, T *__restrict__ sym,
                                                             long long *__restrict__ symidx,
                                                             unsigned long long *__restrict__ nsym_out, long long sym_cap,
                                                             SamplerCarry<T> carry, int have_carry, SamplerCarry<T> *__restrict__ carry_out)
{
    __shared__ T win[LEN];
    __shared__ T o_val[OUT];
    __shared__ unsigned o_idx[OUT];
    const int lane = threadIdx.x;
    const long long C = P.chunk_out;
    const long long n_chunks = (C > 0) ? (P.n_total + C - 1) / C : 0;
    T next = 0, step = P.step0, last = 0;
    long long count = 0, c_first = 0;
    if (have_carry) { next = carry.a; step = carry.b; last = carry.c; count = carry.count0; c_first = carry.c_first; }
    for (long long c = c_first; c < n_chunks; c++) {
        const long long base = c * C;
        const unsigned n_cur = (unsigned)((P.n_total - base < C) ? (P.n_total - base) : C);
        const T nT = (T)n_cur;
        unsigned wbase = 0;
        bool chunk_done = false;
        while (!chunk_done) {
            __syncthreads();
            {
                const unsigned avail = (n_cur > wbase) ? n_cur - wbase : 0u;
                const int n_data = (int)((avail < (unsigned)LEN) ? avail : (unsigned)LEN);
                const T *src = in + base + wbase;
                for (int t = lane; t < n_data; t += PDT_GARDNER_THREADS) win[t] = src[t];
            }
            __syncthreads();
            const unsigned wend = wbase + (unsigned)LEN;
            int nout = 0;
            for (;;) {
                const T rn = mm_rint<T>(next);
                const int in_chunk = uniform<int>((int)(rn < nT));
                if (!in_chunk) { chunk_done = true; break; }
                const unsigned i_abs = uniform<unsigned>((unsigned)rn);
                if (i_abs - wbase >= (unsigned)LEN || nout >= OUT) break;      // new window / flush
                const T cur = win[i_abs - wbase];
                o_val[nout] = cur;
                o_idx[nout] = i_abs;
                nout++;
                const T err = (T)((last > 0) - (last < 0)) * cur - (T)((cur > 0) - (cur < 0)) * last;     // :37 / :67
                step = step + P.kp * err;
                step = (step > P.step_max) ? P.step_max : step;
                step = (step < P.step_min) ? P.step_min : step;
                next = next + step;
                last = cur;
            }
            __syncthreads();
            for (int t = lane; t < nout; t += PDT_GARDNER_THREADS) {
                const long long k = count + t;
                if (k < sym_cap) {
                    sym[k] = o_val[t];
                    symidx[k] = base + (long long)o_idx[t];
                }
            }
            count += nout;
            if (!chunk_done) {
                const unsigned cur_i = uniform<unsigned>((unsigned)mm_rint<T>(next));
                if (cur_i - wbase >= (unsigned)LEN) wbase = cur_i;
            }
            (void)wend;
        }
        next = next - nT;                              // roll over to the next chunk (:80)
    }
    if (threadIdx.x == 0) *nsym_out = (unsigned long long)count;
    if (threadIdx.x == 0 && carry_out) {
        SamplerCarry<T> o;
        o.a = next; o.b = step; o.c = last; o.c_first = n_chunks; o.count0 = count;
        *carry_out = o;
    }
}

// ------------------------------------------------------------------------------------------
// Exact parallel Gardner ("boundary-state table" method, float only)
//
// The sampler state that crosses a chunk boundary is (ns, prev, half).  All three are functions
// of q = the sampling instant of the chunk's last symbol after the error correction and before
// the step is added (ns_end = q + step, half = q + step/2, prev = in[rint(ns_last)] with
// rint(ns_last) in {floor q, floor q + 1}).  At the end of a chunk q lies in
// [n - step - 0.6, n - 0.4), a single float binade for the usual chunk sizes, so q takes only
// (step + ~1.5)/ulp distinct values (4.7 k for n = 30000): the set of *possible* boundary states
// is small and known in advance, although which one occurs depends on the whole past.
//   level 1  (parallel over chunks x candidate states): run every full chunk from candidate
//            entry states; record the exit state's index and the symbol count.  Only the
//            consistent (q, last-pick) combinations are enumerated, and when 64 scout
//            trajectories started at spread-out phases over the tail of the previous chunk all
//            end within one sample of each other (the timing loop is locked) only the
//            candidates around that instant are run;
//   level 2  (one wavefront): follow the true chain  k_{c+1} = table_c[k_c]  from the known
//            start; a state that was not tabulated (outside the scouts' band) is simply walked;
//   level 3  (parallel over chunks): re-run each chunk from its now known entry state and emit
//            its symbols at the prefix-summed offset (k_gardner in parallel mode).
// Every float operation of the true trajectory is executed exactly as in the sequential loop,
// so the result is bit-identical by construction.
// ------------------------------------------------------------------------------------------
struct GardnerDomain {
    float q_min;       // smallest candidate q (multiple of u)
    float u;           // grid spacing (ulp of the binade that contains the chunk end)
    int n_q;           // number of q values; table row = 2 * n_q cells (two choices of the last pick)
    int n_cand;        // consistent (q, pick) combinations, listed in cand_k in increasing q
    int pad_q;         // candidates are tabulated within this many grid points of a scout's end point
    int idx_bits;      // table cell = candidate index (low idx_bits bits) | symbol count of the chunk (the rest)
    int span;          // chunks per table row: boundary states are tabulated in front of every span-th chunk only (row r =
                       // chunks [r span, (r + 1) span): entry key -> exit key after the last of them | symbols of all of them)
};

#ifndef PDT_GTAB_TAIL
#define PDT_GTAB_TAIL 4096           // most samples of the previous chunk the scouts can run over (LDS)
#endif
#ifndef PDT_GTAB_THREADS
#define PDT_GTAB_THREADS 64           // table kernel: threads per block (one or two candidates per lane)
#endif
#ifndef PDT_GTAB_WIN
#define PDT_GTAB_WIN 2048             // table kernel: LDS window in floats (8 KiB: every chunk of a 10-minute capture resident)
#endif
#define PDT_GTAB_MISS 0xffffffffu    // cell not tabulated

__device__ __forceinline__ void gardner_entry_from_candidate(const float *__restrict__ in, const GardnerParams<float> &P,
                                                             const GardnerDomain &D, long long c, int k, float &ns,
                                                             float &prev, float &half)
{
    // candidate k of the boundary between chunk c-1 (full) and chunk c
    const float q = D.q_min + (float)(k >> 1) * D.u;          // exact: both are multiples of u in one binade
    const float hs = (float)((double)P.step / 2.0);
    const float ns_end = q + P.step;
    ns = ns_end - (float)P.chunk_out;
    half = q + hs;
    const long long i_last = (long long)floorf(q) + (k & 1);
    prev = in[(c - 1) * P.chunk_out + i_last];
}

// exit state -> table cell (candidate index | symbol count, split at D.idx_bits); MISS if outside the domain
__device__ __forceinline__ unsigned gardner_encode_exit(const GardnerDomain &D, float q_last, unsigned i_last, unsigned count)
{
    const float mf = (q_last - D.q_min) / D.u;               // exact small integer for in-domain states
    const int m = (int)mf;
    const int v = (int)i_last - (int)floorf(q_last);
    const bool ok = (count > 0) && (mf == (float)m) && (m >= 0) && (m < D.n_q) && (v == 0 || v == 1) &&
                    (count < (1u << (32 - D.idx_bits)) - 1u);
    return ok ? (((unsigned)(2 * m + v)) | (count << D.idx_bits)) : PDT_GTAB_MISS;
}


// one candidate trajectory carried by a lane
struct GardnerLane {
    float ns, prev, half, q_last;
    unsigned i_last, count;
    int k;
    bool active;
};

// wrel[rint(x)] for an LDS window `wrel` (indexed with chunk-relative sample indices) in three instructions: the magic add
// (rint_index), one shift-add and the LDS read.  The integer image of the magic constant is folded into the window's byte
// address once per window -- (asint(x + 1.5 * 2^23) << 2) + (wrel - (0x4B400000 << 2)), all modulo 2^32 -- where the plain form
// spends a shift and a three-operand add per read.  (Round 4: the symbol step of the table / span / emission walkers went
// from 19 to 14 issue slots with this and with the sampler's two additions kept out of a packed operation, Makefile.)
typedef const __attribute__((address_space(3))) float pdt_lds_cf;
__device__ __forceinline__ float lds_rel_at(const float *wrel, float x)
{
    const unsigned base = (unsigned)(size_t)wrel - (0x4B400000u << 2);
    return *(pdt_lds_cf *)(size_t)((__float_as_uint(x + 12582912.0f) << 2) + base);    // (an LDS address: 32 bits on the device)
}

__device__ __forceinline__ void gardner_lane_step(GardnerLane &L, const float *wrel, float kp, float lim, float hs, float step)
{
    const float cur = lds_rel_at(wrel, L.ns);
    const float mid = lds_rel_at(wrel, L.half);
    const float err = __builtin_amdgcn_fmed3f(kp * (cur - L.prev) * mid, -lim, lim);
    L.ns = L.ns - err;
    L.half = L.ns + hs;
    L.ns = L.ns + step;
    L.prev = cur;
}

__device__ __forceinline__ void gardner_lane_tail(GardnerLane &L, const float *wrel, float stop, float kp, float lim, float hs,
                                                  float step)
{
    for (;;) {
        const float rn = __builtin_rintf(L.ns);
        if (!(rn < stop)) break;
        const float cur = wrel[(int)rn];
        const float mid = wrel[rint_index(L.half)];
        const float err = __builtin_amdgcn_fmed3f(kp * (cur - L.prev) * mid, -lim, lim);
        L.ns = L.ns - err;
        L.q_last = L.ns;
        L.half = L.ns + hs;
        L.ns = L.ns + step;
        L.prev = cur;
        L.i_last = (unsigned)rn;
        L.count++;
    }
}

// scouts: one wavefront per chunk.  64 trajectories over the tail of chunk c-1, started one 64th
// of a symbol apart.  In lock they collapse onto a few tight clusters (trajectories that pick the
// same samples receive the same corrections and merge exactly), and the true trajectory -- which has
// been running far longer -- ends on or within a few grid points of one of them.  The chunk's
// candidate list is the union of the neighbourhoods [m - pad, m + pad] of the scouts' end points m;
// when the scouts end more than 1.5 samples apart (no timing lock) the full domain is tabulated.
// (The table itself is preset to "not tabulated" by a memset before this kernel.)
// candidates [j_lo, j_hi) of cand_k, or of the chunk's own list; [k_lo, k_hi] = the keys a look-up in this chunk's table row
// may ask for: the row is not initialised outside it (nor at keys inside it that are no candidates: those are unreachable)
struct GardnerBand { int j_lo, j_hi, listed; unsigned k_lo, k_hi; };
// table cell of chunk c for entry key `key`, PDT_GTAB_MISS outside the chunk's band
__device__ __forceinline__ unsigned gardner_cell(const unsigned *__restrict__ table, size_t stride, const GardnerBand &b,
                                                 long long c, unsigned key);
#define PDT_GTAB_LIST 2048                        // capacity of a chunk's candidate list

__device__ __forceinline__ void k_gardner_scout(const float *__restrict__ in, GardnerParams<float> P, GardnerDomain D,
                                                       long long n_tab_chunks, const int *__restrict__ m_first,
                                                       const unsigned *__restrict__ cand_k, unsigned *__restrict__ table,
                                                       GardnerBand *__restrict__ bands, unsigned *__restrict__ clist,
                                                       unsigned *__restrict__ stats /* [1] full-domain chunks [3] candidates */,
                                                       int tail_max /* samples of the previous chunk the scouts run over, <= PDT_GTAB_TAIL */)
{
    __shared__ float tail[PDT_GTAB_TAIL];
    __shared__ int s_sorted[64], s_lo[64], s_hi[64], s_off[65];
    const long long r = blockIdx.x;                              // table row; its first chunk:
    const long long c = r * D.span;
    if (r >= n_tab_chunks) return;
    const long long C = P.chunk_out;
    const long long base = c * C;
    const int n_cur = (int)C;
    const int tail_n = (C < tail_max) ? (int)C : tail_max;
    const int lane = threadIdx.x;
    GardnerBand bd;
    bd.j_lo = 0;
    bd.j_hi = (c == 0) ? 1 : D.n_cand;
    bd.listed = 0;
    bd.k_lo = (c == 0) ? 0u : cand_k[0];                    // chunk 0: the single start state is cell 0 of its row
    bd.k_hi = (c == 0) ? 0u : cand_k[bd.j_hi - 1];
    if (c >= 1) {
        const float hs = (float)((double)P.step / 2.0);
        const float kp = P.kp, lim = P.lim, step = P.step, nT = (float)n_cur;
        for (int t = lane; t < tail_n; t += 64 * 8) {
            float r[8];
#pragma unroll
            for (int u = 0; u < 8; u++) r[u] = (t + u * 64 < tail_n) ? in[base - tail_n + t + u * 64] : 0.0f;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (t + u * 64 < tail_n) tail[t + u * 64] = r[u];
        }
        __syncthreads();
        const float t0 = (float)(n_cur - tail_n);
        float ns = t0 + 8.0f + step * (float)lane * (1.0f / 64.0f);
        float prev = 0, half = ns - hs, q_last = ns;
        for (;;) {
            const float rn = __builtin_rintf(ns);
            if (!(rn < nT)) break;
            const int i_cur = (int)(rn - t0);
            int i_half = (int)(__builtin_rintf(half) - t0);
            i_half = (i_half < 0) ? 0 : i_half;
            const float cur = tail[i_cur];
            const float mid = tail[(i_half < tail_n) ? i_half : tail_n - 1];
            const float err = __builtin_amdgcn_fmed3f(kp * (cur - prev) * mid, -lim, lim);
            ns = ns - err;
            q_last = ns;
            half = ns + hs;
            ns = ns + step;
            prev = cur;
        }
        const int m = (int)floorf((q_last - D.q_min) / D.u);
        // rank sort of the 64 end points
        int rank = 0;
        for (int j = 0; j < 64; j++) {
            const int mj = __builtin_amdgcn_readlane(m, j);
            rank += (mj < m || (mj == m && j < lane)) ? 1 : 0;
        }
        s_sorted[rank] = m;
        __syncthreads();
        const int mine = s_sorted[lane];
        const int before = (lane > 0) ? s_sorted[lane - 1] : mine;
        const int m_min = s_sorted[0], m_max = s_sorted[63];
        const int one = (int)(1.0f / D.u);                        // grid points per sample
        const bool locked = (m_max - m_min) <= one + one / 2 && m_min >= 0 && m_max < D.n_q;
        // a lane opens an interval when its neighbourhood does not touch the previous scout's
        const bool opens = lane == 0 || (mine - before) > 2 * D.pad_q + 1;
        const unsigned long long open_mask = __ballot(opens);
        int cnt = 0;
        if (opens) {
            const unsigned long long above = (lane < 63) ? (open_mask >> (lane + 1)) : 0ull;
            const int last = above ? (lane + __builtin_ctzll(above)) : 63;       // last scout of this interval
            int m_lo = mine - D.pad_q, m_hi = s_sorted[last] + D.pad_q;
            m_lo = (m_lo < 0) ? 0 : m_lo;
            m_hi = (m_hi > D.n_q - 1) ? D.n_q - 1 : m_hi;
            m_lo = (m_lo > D.n_q - 1) ? D.n_q - 1 : m_lo;
            m_hi = (m_hi < m_lo) ? m_lo : m_hi;
            s_lo[lane] = m_first[m_lo];
            cnt = m_first[m_hi + 1] - s_lo[lane];
        }
        s_hi[lane] = cnt;
        __syncthreads();
        if (lane == 0) {
            int acc = 0;
            for (int j = 0; j < 64; j++) { s_off[j] = acc; acc += s_hi[j]; }
            s_off[64] = acc;
        }
        __syncthreads();
        const int total = s_off[64];
        if (locked && total <= PDT_GTAB_LIST) {
            unsigned *mylist = clist + (size_t)r * PDT_GTAB_LIST;
            for (int j = 0; j < 64; j++) {
                const int nj = s_hi[j];
                if (nj == 0) continue;
                const int jl = s_lo[j], off = s_off[j];
                for (int t = lane; t < nj; t += 64) mylist[off + t] = cand_k[jl + t];
            }
            bd.j_lo = 0;
            bd.j_hi = total;
            bd.listed = 1;
            // the listed keys lie in [k_lo, k_hi]; the table kernel writes the listed ones, every other key of that range
            // must read as a miss: clear the range (a few thousand cells; the whole table is never initialised)
            int last = 0;
            for (int j = 0; j < 64; j++) last = (s_hi[j] > 0) ? j : last;
            bd.k_lo = cand_k[s_lo[0]];
            bd.k_hi = cand_k[s_lo[last] + s_hi[last] - 1];
            unsigned *row = table + (size_t)r * (size_t)(2 * D.n_q);
            for (unsigned k = bd.k_lo + (unsigned)lane; k <= bd.k_hi; k += 64u) row[k] = PDT_GTAB_MISS;
        }
    }
    // (no global counters here: 3 000 atomics on one address drained for 0.1 ms after the last wavefront had finished;
    // k_gardner_chain sums the candidates and the full-domain chunks from the bands)
    if (lane == 0) bands[r] = bd;
}

// level 1: block (r, p) runs slice p of row r's candidate list through the row's first chunk, window by window, merging as it
// goes.  Candidate trajectories of one chunk collapse onto each other as they go
// (same picks -> same corrections -> identical state from then on): of ~150 entry states only a
// handful of distinct trajectories are left after a few hundred symbols.  A workgroup of two
// wavefronts therefore starts with up to 256 candidates (two per lane), and at every window seam,
// while more than 64 trajectories are alive, identical states (ns, prev, half) are found through a
// hash table in LDS, the duplicates are dropped -- each original candidate remembers which survivor
// carries it and by how many symbols its own count differs -- and the survivors are packed into the
// low lanes.  From 128 survivors on one trajectory per lane is walked, from 64 on one wavefront.
// Every walked trajectory executes exactly the arithmetic of the sequential loop; merging only
// avoids repeating identical work, so the table is the same as without it.
#define PDT_GTM_THREADS 128
#define PDT_GTM_SLOTS 256
#define PDT_GTM_HASH 512
template <int WIN>
__device__ __forceinline__ void k_gardner_table_merge(const float *__restrict__ in, GardnerParams<float> P,
                                                                          GardnerDomain D, long long n_tab_chunks,
                                                                          const unsigned *__restrict__ cand_k,
                                                                          const GardnerBand *__restrict__ bands,
                                                                          const unsigned *__restrict__ clist,
                                                                          unsigned *__restrict__ table,
                                                                          unsigned *__restrict__ stats /* [0] bad */)
{
    static_assert(WIN * 4 >= PDT_GTM_SLOTS * 4 * 6 + PDT_GTM_HASH * 4, "exchange arrays alias the sample window");
    __shared__ __attribute__((aligned(16))) float win[WIN];
    __shared__ unsigned short own[PDT_GTM_SLOTS], dupof[PDT_GTM_SLOTS], newidx[PDT_GTM_SLOTS];
    __shared__ int extra[PDT_GTM_SLOTS];
    __shared__ unsigned s_wcnt[2 * (PDT_GTM_THREADS / 64)];
    // exchange arrays, valid only between two windows (they overlay the sample window)
    float *x_ns = win, *x_prev = win + PDT_GTM_SLOTS, *x_half = win + 2 * PDT_GTM_SLOTS, *x_q = win + 3 * PDT_GTM_SLOTS;
    unsigned *x_cnt = reinterpret_cast<unsigned *>(win + 4 * PDT_GTM_SLOTS);
    unsigned *x_il = reinterpret_cast<unsigned *>(win + 5 * PDT_GTM_SLOTS);
    unsigned *htab = reinterpret_cast<unsigned *>(win + 6 * PDT_GTM_SLOTS);

    const long long r = blockIdx.x;                 // table row
    if (r >= n_tab_chunks) return;
    const long long c = r * D.span;                 // its first chunk (always a full one)
    const GardnerBand bd = bands[r];
    const int j0 = bd.j_lo + (int)blockIdx.y * PDT_GTM_SLOTS;
    const int j_hi = bd.j_hi;
    if (j0 >= j_hi) return;
    const unsigned *cand = bd.listed ? (clist + (size_t)blockIdx.x * PDT_GTAB_LIST) : cand_k;
    unsigned *row = table + (size_t)r * (size_t)(2 * D.n_q);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long C = P.chunk_out;
    const long long base = c * C;
    const int n_cur = (int)C;
    const float hs = (float)((double)P.step / 2.0);
    const float kp = P.kp, lim = P.lim, step = P.step, nT = (float)n_cur;
    const int margin = 2 * (int)step + 24;          // look-ahead the staged data must cover past a stop point
    const int back = (int)step + 8;                 // a mid-point lies at most this far behind a stop point

    int n_alive = (j_hi - j0 < PDT_GTM_SLOTS) ? (j_hi - j0) : PDT_GTM_SLOTS;     // trajectories at positions [0, n_alive)
    const int n_slots = n_alive;                                                  // original candidates of this block
    // position p lives in lane p & 127, slot p >> 7
    GardnerLane L[2];
    int my_k[2];
#pragma unroll
    for (int l = 0; l < 2; l++) {
        const int p = tid + l * PDT_GTM_THREADS;
        L[l].ns = L[l].prev = L[l].half = L[l].q_last = 0;
        L[l].i_last = L[l].count = 0;
        L[l].k = 0;
        L[l].active = false;
        my_k[l] = 0;
        if (p < n_slots) {
            if (c >= 1) {                                   // chunk 0: the single start state is cell 0, all zero
                my_k[l] = (int)cand[j0 + p];
                gardner_entry_from_candidate(in, P, D, c, my_k[l], L[l].ns, L[l].prev, L[l].half);
            }
            own[p] = (unsigned short)p;
            extra[p] = 0;
        }
    }
    // positions beyond n_alive shadow position 0 so that every lane stays inside the staged windows
    if (tid == 0) { x_ns[0] = L[0].ns; x_prev[0] = L[0].prev; x_half[0] = L[0].half; }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 2; l++)
        if (tid + l * PDT_GTM_THREADS >= n_alive) { L[l].ns = x_ns[0]; L[l].prev = x_prev[0]; L[l].half = x_half[0]; }

    int wbase = 0;
    float enter_hi = step + 1.2f;                 // upper bound of ns when entering the window
    for (;;) {
        // ---- stage [wbase, wbase + WIN)
        __syncthreads();
        for (int t0 = 0; t0 < WIN; t0 += PDT_GTM_THREADS * 8) {
            float r[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int t = t0 + u * PDT_GTM_THREADS + tid;
                const int idx = wbase + t;
                r[u] = (t < WIN && idx < n_cur) ? in[base + idx] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int t = t0 + u * PDT_GTM_THREADS + tid;
                const int idx = wbase + t;
                if (t < WIN) {
                    if (idx >= n_cur && idx < n_cur + margin)
                        r[u] = gardner_beyond(in, (const float *)nullptr, P, c, (long long)n_cur, (long long)idx);
                    win[t] = r[u];
                }
            }
        }
        __syncthreads();
        const int wend = wbase + WIN;
        const bool last_window = (wend - margin >= n_cur);
        const float stop = last_window ? nT : (float)(wend - margin);    // lanes leave the window at rint(ns) >= stop
        const float *wrel = win - wbase;          // indexed with chunk-relative indices
        const bool two = n_alive > PDT_GTM_THREADS;                       // block-uniform
        const int waves_on = two ? (PDT_GTM_THREADS / 64) : ((n_alive + 63) >> 6);
        if (wave < waves_on) {
            if (wbase == 0) {
                // first symbol: the mid-point index is the stale one of the previous chunk (Q3)
#pragma unroll
                for (int l = 0; l < 2; l++) {
                    const float rn = __builtin_rintf(L[l].ns);
                    if (rn < nT) {
                        const unsigned i_cur = (unsigned)rn;
                        const unsigned i_half = (unsigned)__builtin_rintf(L[l].half);
                        const float cur = win[i_cur];
                        float mid;
                        if (i_half < (unsigned)WIN) mid = win[i_half];
                        else mid = (i_half < (unsigned)n_cur) ? in[base + i_half]
                                                              : gardner_beyond(in, (const float *)nullptr, P, c, (long long)n_cur, (long long)i_half);
                        const float err = __builtin_amdgcn_fmed3f(kp * (cur - L[l].prev) * mid, -lim, lim);
                        L[l].ns = L[l].ns - err;
                        L[l].q_last = L[l].ns;
                        L[l].half = L[l].ns + hs;
                        L[l].ns = L[l].ns + step;
                        L[l].prev = cur;
                        L[l].i_last = i_cur;
                        L[l].count = 1;
                    }
                }
            }
            if (L[0].count >= 1 && L[1].count >= 1) {
                int k_min = (int)((stop - 4.0f - enter_hi) / (step + 0.101f)) - 1;
                if (k_min < 0) k_min = 0;
                if (two) {
                    for (int it = 0; it < k_min; it++) {
                        gardner_lane_step(L[0], wrel, kp, lim, hs, step);
                        gardner_lane_step(L[1], wrel, kp, lim, hs, step);
                    }
                    L[0].count += (unsigned)k_min;
                    L[1].count += (unsigned)k_min;
                    gardner_lane_tail(L[0], wrel, stop, kp, lim, hs, step);
                    gardner_lane_tail(L[1], wrel, stop, kp, lim, hs, step);
                } else {
                    for (int it = 0; it < k_min; it++) gardner_lane_step(L[0], wrel, kp, lim, hs, step);
                    L[0].count += (unsigned)k_min;
                    gardner_lane_tail(L[0], wrel, stop, kp, lim, hs, step);
                }
            }
        }
        if (last_window) break;
        enter_hi = stop + step + 1.2f;
        wbase = wend - margin - back;
        if (n_alive <= 64) continue;
        // ---- merge identical trajectories (the sample window is dead until the next staging)
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 2; l++) {
            const int p = tid + l * PDT_GTM_THREADS;
            if (p < n_alive) { x_ns[p] = L[l].ns; x_prev[p] = L[l].prev; x_half[p] = L[l].half; x_cnt[p] = L[l].count; }
        }
        for (int t = tid; t < PDT_GTM_HASH; t += PDT_GTM_THREADS) htab[t] = 0xffffffffu;
        __syncthreads();
        bool surv[2];
#pragma unroll
        for (int l = 0; l < 2; l++) {
            const int p = tid + l * PDT_GTM_THREADS;
            surv[l] = false;
            if (p < n_alive) {
                const unsigned a = __float_as_uint(L[l].ns), b = __float_as_uint(L[l].half), d = __float_as_uint(L[l].prev);
                unsigned h = (a * 2654435761u) ^ (b * 40503u) ^ (d * 2246822519u);
                h = (h >> 7) & (PDT_GTM_HASH - 1);
                for (;;) {
                    const unsigned old = atomicCAS(&htab[h], 0xffffffffu, (unsigned)p);
                    if (old == 0xffffffffu) { dupof[p] = (unsigned short)p; surv[l] = true; break; }
                    if (__float_as_uint(x_ns[old]) == a && __float_as_uint(x_half[old]) == b && __float_as_uint(x_prev[old]) == d) {
                        dupof[p] = (unsigned short)old;
                        break;
                    }
                    h = (h + 1) & (PDT_GTM_HASH - 1);
                }
            }
        }
        // survivors get consecutive new positions: slot-0 survivors of wave 0, wave 1, then slot-1 survivors
        unsigned long long bal[2];
        bal[0] = __ballot(surv[0]);
        bal[1] = __ballot(surv[1]);
        if ((tid & 63) == 0) {
            s_wcnt[wave] = (unsigned)__popcll(bal[0]);
            s_wcnt[PDT_GTM_THREADS / 64 + wave] = (unsigned)__popcll(bal[1]);
        }
        __syncthreads();
        // every original candidate follows its carrier to the survivor and books the count difference
#pragma unroll
        for (int l = 0; l < 2; l++) {
            const int q = tid + l * PDT_GTM_THREADS;
            if (q < n_slots) {
                const unsigned o = own[q];
                const unsigned dd = dupof[o];
                if (dd != o) {
                    extra[q] += (int)x_cnt[o] - (int)x_cnt[dd];
                    own[q] = (unsigned short)dd;
                }
            }
        }
        unsigned before[2] = {0, 0};
        unsigned total = 0;
        for (int g = 0; g < 2 * (PDT_GTM_THREADS / 64); g++) {
            const unsigned v = s_wcnt[g];
            if (g < wave) before[0] += v;
            if (g < PDT_GTM_THREADS / 64 + wave) before[1] += v;
            total += v;
        }
        const unsigned long long lt = (1ull << (tid & 63)) - 1ull;
        unsigned np[2];
        np[0] = before[0] + (unsigned)__popcll(bal[0] & lt);
        np[1] = before[1] + (unsigned)__popcll(bal[1] & lt);
#pragma unroll
        for (int l = 0; l < 2; l++)
            if (surv[l]) newidx[tid + l * PDT_GTM_THREADS] = (unsigned short)np[l];
        __syncthreads();
        // move the surviving states to their new positions (all reads of the old arrays are done)
#pragma unroll
        for (int l = 0; l < 2; l++) {
            if (surv[l]) {
                const unsigned q = np[l];
                x_ns[q] = L[l].ns; x_prev[q] = L[l].prev; x_half[q] = L[l].half; x_q[q] = L[l].q_last;
                x_cnt[q] = L[l].count; x_il[q] = L[l].i_last;
            }
            const int q = tid + l * PDT_GTM_THREADS;
            if (q < n_slots) own[q] = newidx[own[q]];
        }
        __syncthreads();
        n_alive = (int)total;
#pragma unroll
        for (int l = 0; l < 2; l++) {
            const int p = tid + l * PDT_GTM_THREADS;
            const int src = (p < n_alive) ? p : 0;
            L[l].ns = x_ns[src]; L[l].prev = x_prev[src]; L[l].half = x_half[src]; L[l].q_last = x_q[src];
            L[l].count = x_cnt[src]; L[l].i_last = x_il[src];
        }
    }
    // ---- exit states of the survivors, then one table cell per original candidate
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 2; l++) {
        const int p = tid + l * PDT_GTM_THREADS;
        if (p < n_alive) { x_q[p] = L[l].q_last; x_il[p] = L[l].i_last; x_cnt[p] = L[l].count; }
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 2; l++) {
        const int q = tid + l * PDT_GTM_THREADS;
        if (q < n_slots) {
            const unsigned o = own[q];
            const unsigned cnt = (unsigned)((int)x_cnt[o] + extra[q]);
            const unsigned cell = gardner_encode_exit(D, x_q[o], x_il[o], cnt);
            if (cell == PDT_GTAB_MISS) atomicAdd(&stats[0], 1u);  // exit outside the enumerated domain (never expected)
            row[my_k[l]] = cell;
        }
    }
}

// Rows that span several chunks (D.span > 1).  k_gardner_table* has filled row r with the exits of the row's FIRST chunk.  The
// candidates of a chunk collapse onto a handful of trajectories, so those cells hold only a few distinct exit keys (thousands
// where the timing loop is not locked -- noise in front of a pass: the scouts then list the whole domain).  Three kernels:
//   k_gardner_span_keys   one workgroup per row: the distinct exit keys of the row's cells (a bitmap of the key domain in LDS),
//                         written in ascending order to a shared key list, and cut into work items of up to 64 keys;
//   k_gardner_span_walk   persistent wavefronts take the work items (an atomic cursor): one lane per key walks the
//                         other span - 1 chunks of its row -- every float operation that of the sequential loop, the roll-over at
//                         every chunk end included -- and leaves, per key, the exit after the row's last chunk | symbols of those
//                         chunks;
//   k_gardner_span_join   one workgroup per row: every cell becomes  exit key after the row's last chunk | symbols of all its
//                         chunks  (the key's place in the row's sorted key list by binary search).
// Scouts and candidate walks are then needed in front of every span-th chunk only, and the chain hops span chunks per look-up.
// A row whose keys do not fit the shared list any more is left untabulated (its band is emptied: the chain walks it).
#define PDT_GSPAN_BITMAP_WORDS 8192              // 2 n_q <= 262 144 keys
// (Measured at an hour of 250 ksps, 5 625 rows of ~20 distinct exits: one row per wavefront 2.7 ms; four rows per wavefront,
// 16 keys each, 3.6 ms -- the walk of a lone wavefront is bound by the latency of its symbol step, ~600 clocks, not by issue slots,
// so what counts is the number of resident wavefronts.  The same packing for the emission: 2.65 against 2.4 ms.)
#define PDT_GSUB 1                                   // rows per wavefront
#define PDT_GSUB_WIN 2048                            // LDS window (floats)
#define PDT_GSUB_KEYS (64 / PDT_GSUB)                // keys per work item
struct GardnerSpanRow { unsigned off, n; };     // the row's keys: [off, off + n) of the key list; n = ~0u: not tabulated
struct GardnerSpanItem { unsigned row, first, cnt; };
struct GardnerSpanRec { float ns, prev, half; unsigned count; };   // sampler state in front of a chunk, symbols since the row's second chunk
struct GardnerSpanCtl { unsigned long long keys; unsigned items, cursor, overflow, pad_; };     // keys: a 64-bit cursor (never wraps)

__device__ __forceinline__ void k_gardner_span_keys(GardnerDomain D, long long n_rows, const unsigned *__restrict__ cand_k,
                                                    GardnerBand *__restrict__ bands, const unsigned *__restrict__ clist,
                                                    const unsigned *__restrict__ table, unsigned *__restrict__ keys,
                                                    unsigned cap_keys, GardnerSpanRow *__restrict__ rows,
                                                    GardnerSpanItem *__restrict__ items, GardnerSpanCtl *__restrict__ ctl)
{
    __shared__ unsigned s_bits[PDT_GSPAN_BITMAP_WORDS];
    __shared__ unsigned s_scan[256];
    __shared__ unsigned s_off, s_item;
    const long long r = blockIdx.x;
    if (r >= n_rows) return;
    const long long c0 = r * D.span;
    const GardnerBand bd = bands[r];
    const unsigned *cand = bd.listed ? (clist + (size_t)r * PDT_GTAB_LIST) : cand_k;
    const unsigned *row = table + (size_t)r * (size_t)(2 * D.n_q);
    const int tid = threadIdx.x;
    const int n_words = (2 * D.n_q + 31) >> 5;
    const unsigned kmask = (1u << D.idx_bits) - 1u;
    for (int t = tid; t < n_words; t += 256) s_bits[t] = 0u;
    __syncthreads();
    for (int j = bd.j_lo + tid; j < bd.j_hi; j += 256) {
        const unsigned cell = row[(c0 == 0) ? 0u : cand[j]];            // (chunk 0: the single start state is cell 0)
        if (cell != PDT_GTAB_MISS) {
            const unsigned k1 = cell & kmask;
            atomicOr(&s_bits[k1 >> 5], 1u << (k1 & 31u));
        }
    }
    __syncthreads();
    // every thread owns a contiguous range of words: count, scan, write in ascending key order
    const int per = (n_words + 255) / 256;
    const int w0 = tid * per, w1 = (w0 + per < n_words) ? w0 + per : n_words;
    unsigned mine = 0;
    for (int t = w0; t < w1; t++) mine += (unsigned)__popc(s_bits[t]);
    s_scan[tid] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned v = (tid >= d) ? s_scan[tid - d] : 0u;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    const unsigned total = s_scan[255];
    if (tid == 0) {
        // Space in the shared key list is handed out by one fetch-and-add per row, and a reservation that does not fit is NOT
        // handed back: from the first overflow on the cursor stands behind cap_keys and every later row fails as well (its band
        // is emptied, the chain walks it) -- successful ranges can never overlap, which a give-back by subtraction allowed
        // (ADVICE r4).  The cursor is 64 bits wide: however many rows fail, it cannot wrap.  (Measured on the way: a
        // compare-and-swap loop that only reserves what fits -- 5 625 rows arriving together retry each other quadratically, 24 ms
        // for this kernel at an hour of 250 ksps; a look at the overflow flag in front of the add -- 0.20 -> 0.44 ms, the load
        // queues behind the other rows' atomics on the same line.)
        unsigned off = ~0u;
        {
            const unsigned long long got = atomicAdd(&ctl->keys, (unsigned long long)total);
            if (got + total <= (unsigned long long)cap_keys) off = (unsigned)got;
        }
        if (off == ~0u) atomicAdd(&ctl->overflow, 1u);
        else s_item = atomicAdd(&ctl->items, (total + (unsigned)PDT_GSUB_KEYS - 1u) / (unsigned)PDT_GSUB_KEYS);
        s_off = off;
    }
    __syncthreads();
    const unsigned off = s_off;
    if (off == ~0u) {
        if (tid == 0) {
            GardnerSpanRow rw;
            rw.off = 0; rw.n = ~0u;
            rows[r] = rw;
            GardnerBand nb = bd;                                         // no key of this row is tabulated any more
            nb.k_lo = 1u; nb.k_hi = 0u;
            bands[r] = nb;
        }
        return;
    }
    unsigned at = off + s_scan[tid] - mine;
    for (int t = w0; t < w1; t++) {
        unsigned bits = s_bits[t];
        while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1u;
            keys[at++] = ((unsigned)t << 5) + (unsigned)b;
        }
    }
    for (unsigned q = tid; q < (total + (unsigned)PDT_GSUB_KEYS - 1u) / (unsigned)PDT_GSUB_KEYS; q += 256u) {
        GardnerSpanItem it;
        it.row = (unsigned)r;
        it.first = off + (unsigned)PDT_GSUB_KEYS * q;
        it.cnt = (total - (unsigned)PDT_GSUB_KEYS * q < (unsigned)PDT_GSUB_KEYS) ? total - (unsigned)PDT_GSUB_KEYS * q : (unsigned)PDT_GSUB_KEYS;
        items[s_item + q] = it;
    }
    if (tid == 0) {
        GardnerSpanRow rw;
        rw.off = off; rw.n = total;
        rows[r] = rw;
    }
}

// A wavefront as NSUB sub-groups of 64 / NSUB lanes, each sub-group in a chunk of its own (of another row / group): the
// walk of a chunk by a wavefront costs the same whether one lane or all of them carry a trajectory, and a row has only a
// handful -- so four rows share the instruction stream.  All chunks are full ones, so the window schedule (and with it every
// loop bound) is the same for all sub-groups; only the window a lane reads from is its sub-group's.  EMIT: the sub-group's first
// lane also stores every symbol (value, global sample index) at sym_at + its running count.
#define PDT_GSUB_OUT 64                               // symbols a sub-group can stage per window (WIN / (step - 0.1) + 2 must fit)
template <int WIN, int NSUB, bool EMIT>
__device__ __forceinline__ void gardner_sub_chunk(float *win, const float *__restrict__ in, const GardnerParams<float> &P,
                                                  const long long (&c_sub)[NSUB], GardnerLane &L, float *__restrict__ sym,
                                                  long long *__restrict__ symidx, long long sym_at, long long sym_cap,
                                                  float *o_val = nullptr, unsigned *o_idx = nullptr /* EMIT: LDS, NSUB x GSUB_OUT each */)
{
    constexpr int SUBW = 64 / NSUB;
    const int lane = (int)threadIdx.x;
    const int mysub = lane / SUBW;
    // EMIT: every lane of a sub-group carries the same trajectory; the symbols of a window are staged in LDS (all lanes write the
    // same word: no mask) and leave with the sub-group's lanes side by side at the window's end
    float *ov = o_val + mysub * PDT_GSUB_OUT;
    unsigned *oi = o_idx + mysub * PDT_GSUB_OUT;
    int nout = 0;
    const long long C = P.chunk_out;
    const int n_cur = (int)C;
    const float hs = (float)((double)P.step / 2.0);
    const float kp = P.kp, lim = P.lim, step = P.step, nT = (float)n_cur;
    const int margin = 2 * (int)step + 24;          // look-ahead the staged data must cover past a stop point
    const int back = (int)step + 8;                 // a mid-point lies at most this far behind a stop point
    long long my_base = 0;
#pragma unroll
    for (int u = 0; u < NSUB; u++) my_base = (mysub == u) ? c_sub[u] * C : my_base;
    int wbase = 0;
    float enter_hi = step + 1.2f;                   // upper bound of ns when entering the window
    auto emit = [&](float cur, unsigned i_cur) {
        if (EMIT) {
            ov[nout] = cur;
            oi[nout] = i_cur;
            nout++;
        }
    };
    auto flush = [&]() {
        if (EMIT) {
            // (the LDS words were written by this very wavefront, in program order: no barrier needed)
            const long long k0 = sym_at + (long long)L.count - (long long)nout;
            if (L.active)
                for (int t = lane % SUBW; t < nout; t += SUBW) {
                    const long long k = k0 + t;
                    if (k < sym_cap) {
                        sym[k] = ov[t];
                        symidx[k] = my_base + (long long)oi[t];
                    }
                }
            nout = 0;
        }
    };
#pragma unroll 1
    for (;;) {
        __syncthreads();
        if (wbase + WIN <= n_cur) {
            // a window inside the chunk: 16-byte loads (the chunk's samples are 4-byte aligned only), no test
            struct __attribute__((packed, aligned(4))) Q { float v[4]; };
#pragma unroll
            for (int u = 0; u < NSUB; u++) {
                const float *src = in + c_sub[u] * C + wbase;
                float *wu = win + u * WIN;
#pragma unroll
                for (int v = 0; v < WIN / 256; v++) {
                    const int t = (v * 64 + lane) * 4;
                    const Q r = *reinterpret_cast<const Q *>(src + t);
                    *reinterpret_cast<float4 *>(wu + t) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
                }
            }
        } else {
            // the chunk's last window: the samples, then the few values the sampler can ask for past the end (Q3)
#pragma unroll 1
            for (int t = lane; t < NSUB * WIN; t += 64) {
                const int u = t / WIN, tt = t - u * WIN;
                long long cu = c_sub[0];
#pragma unroll
                for (int q = 1; q < NSUB; q++) cu = (u == q) ? c_sub[q] : cu;
                const int idx = wbase + tt;
                float r = 0.0f;
                if (idx < n_cur) r = in[cu * C + idx];
                else if (idx < n_cur + margin) r = gardner_beyond(in, (const float *)nullptr, P, cu, (long long)n_cur, (long long)idx);
                win[t] = r;
            }
        }
        __syncthreads();
        const int wend = wbase + WIN;
        const bool last_window = (wend - margin >= n_cur);
        const float stop = last_window ? nT : (float)(wend - margin);    // lanes leave the window at rint(ns) >= stop
        const float *wrel = win + mysub * WIN - wbase;                    // indexed with chunk-relative indices
        if (wbase == 0) {
            // first symbol: the mid-point index is the stale one of the previous chunk (Q3)
            const float rn = __builtin_rintf(L.ns);
            const unsigned i_cur = (unsigned)rn;
            const unsigned i_half = (unsigned)__builtin_rintf(L.half);
            const float cur = wrel[i_cur];
            float mid;
            if (i_half < (unsigned)WIN) mid = wrel[i_half];
            else mid = (i_half < (unsigned)n_cur) ? in[my_base + i_half]
                                                  : gardner_beyond(in, (const float *)nullptr, P, my_base / C, (long long)n_cur, (long long)i_half);
            emit(cur, i_cur);
            const float err = __builtin_amdgcn_fmed3f(kp * (cur - L.prev) * mid, -lim, lim);
            L.ns = L.ns - err;
            L.q_last = L.ns;
            L.half = L.ns + hs;
            L.ns = L.ns + step;
            L.prev = cur;
            L.i_last = i_cur;
            L.count += 1;
        }
        int k_min = (int)((stop - 4.0f - enter_hi) / (step + 0.101f)) - 1;
        if (k_min < 0) k_min = 0;
#pragma unroll 4
        for (int it = 0; it < k_min; it++) {
            if (EMIT) {
                const int ic = rint_index(L.ns);
                const float cur = lds_rel_at(wrel, L.ns);
                const float mid = lds_rel_at(wrel, L.half);
                emit(cur, (unsigned)ic);
                const float err = __builtin_amdgcn_fmed3f(kp * (cur - L.prev) * mid, -lim, lim);
                L.ns = L.ns - err;
                L.half = L.ns + hs;
                L.ns = L.ns + step;
                L.prev = cur;
                L.count += 1;
            } else
                gardner_lane_step(L, wrel, kp, lim, hs, step);
        }
        if (!EMIT) L.count += (unsigned)k_min;
        for (;;) {
            const float rn = __builtin_rintf(L.ns);
            if (!(rn < stop)) break;
            const float cur = wrel[(int)rn];
            const float mid = wrel[rint_index(L.half)];
            emit(cur, (unsigned)rn);
            const float err = __builtin_amdgcn_fmed3f(kp * (cur - L.prev) * mid, -lim, lim);
            L.ns = L.ns - err;
            L.q_last = L.ns;
            L.half = L.ns + hs;
            L.ns = L.ns + step;
            L.prev = cur;
            L.i_last = (unsigned)rn;
            L.count++;
        }
        flush();
        if (last_window) break;
        enter_hi = stop + step + 1.2f;
        wbase = wend - margin - back;
    }
}


template <int WIN>
__device__ __forceinline__ void k_gardner_span_walk(const float *__restrict__ in, GardnerParams<float> P, GardnerDomain D,
                                                    const unsigned *__restrict__ keys, const GardnerSpanItem *__restrict__ items,
                                                    GardnerSpanCtl *__restrict__ ctl, unsigned *__restrict__ tails,
                                                    GardnerSpanRec *__restrict__ recs /* per key: the state in front of each of those chunks */)
{
    __shared__ __attribute__((aligned(16))) float win[PDT_GSUB * WIN];
    __shared__ float s_ref[3];
    __shared__ unsigned s_next;
    const int lane = threadIdx.x;
    const int mysub = lane / PDT_GSUB_KEYS, sublane = lane % PDT_GSUB_KEYS;
    const float nT = (float)P.chunk_out;
    const unsigned n_items = ctl->items;                                 // (final: the key kernel has finished)
    for (;;) {
        __syncthreads();
        if (lane == 0) s_next = atomicAdd(&ctl->cursor, (unsigned)PDT_GSUB);
        __syncthreads();
        const unsigned q0 = s_next;
        if (q0 >= n_items) break;
        // sub-group u takes item q0 + u; the ones past the end shadow item q0
        long long c_sub[PDT_GSUB];
        GardnerSpanItem mine = items[q0];
#pragma unroll
        for (int u = 0; u < PDT_GSUB; u++) {
            const GardnerSpanItem it = (q0 + (unsigned)u < n_items) ? items[q0 + u] : items[q0];
            c_sub[u] = (long long)it.row * D.span + 1;
            if (mysub == u) {
                mine = it;
                if (q0 + (unsigned)u >= n_items) mine.cnt = 0;
            }
        }
        GardnerLane L;
        L.q_last = 0; L.i_last = 0; L.count = 0; L.k = 0;
        L.active = (unsigned)sublane < mine.cnt;
        L.ns = L.prev = L.half = 0;
        long long my_c = 0;
#pragma unroll
        for (int u = 0; u < PDT_GSUB; u++) my_c = (mysub == u) ? c_sub[u] : my_c;
        if (L.active) gardner_entry_from_candidate(in, P, D, my_c, (int)keys[mine.first + sublane], L.ns, L.prev, L.half);
        if (lane == 0) { s_ref[0] = L.ns; s_ref[1] = L.prev; s_ref[2] = L.half; }       // (item q0's first key: always there)
        __syncthreads();
        // idle lanes start from that state, over their own sub-group's samples: whatever the samples, a symbol step advances by
        // step +- 0.1, so they stay inside the staged windows like everybody else
        if (!L.active) { L.ns = s_ref[0]; L.prev = s_ref[1]; L.half = s_ref[2]; }
        for (int g = 1; g < D.span; g++) {
            long long cc[PDT_GSUB];
#pragma unroll
            for (int u = 0; u < PDT_GSUB; u++) cc[u] = c_sub[u] + (g - 1);
            if (L.active) {
                GardnerSpanRec rc;
                rc.ns = L.ns; rc.prev = L.prev; rc.half = L.half; rc.count = L.count;
                recs[(size_t)(mine.first + sublane) * (size_t)(D.span - 1) + (size_t)(g - 1)] = rc;
            }
            gardner_sub_chunk<WIN, PDT_GSUB, false>(win, in, P, cc, L, nullptr, nullptr, 0, 0);
            if (g + 1 < D.span) L.ns = L.ns - nT;                         // roll over; `half` is deliberately not (Q3)
        }
        if (L.active) tails[mine.first + sublane] = gardner_encode_exit(D, L.q_last, L.i_last, L.count);
    }
}

// ---- emission for rows of several chunks: four chunks per wavefront
// The wavefront-per-group emission spends a whole wavefront's instruction stream on one trajectory.  Here a wavefront carries four
// (quarter wavefronts, gardner_sub_chunk), and -- so that there are enough wavefronts to cover the latency of a symbol step --
// the unit of work is a CHUNK, not a group: k_gardner_emit_first walks (and emits) the first chunk of every group from the
// group's entry state (the chain's); its exit names the trajectory the span walkers recorded, whose states in front of the
// group's other chunks become those chunks' entry states; k_gardner_emit_rest then takes all the other chunks, four to a
// wavefront.  flags[g] = 1: the group could not be resolved (no table row, keys not listed, exit not among them) and goes
// through the wavefront-per-group kernel.
#define PDT_GEMIT_SUB_WIN 512
template <int WIN>
__device__ __forceinline__ void k_gardner_emit_first(const float *__restrict__ in, GardnerParams<float> P, GardnerDomain D,
                                                     long long n_rows, const GardnerEntry<float> *__restrict__ entries,
                                                     const unsigned *__restrict__ keys, const GardnerSpanRow *__restrict__ rows,
                                                     const GardnerSpanRec *__restrict__ recs,
                                                     GardnerEntry<float> *__restrict__ centries, unsigned char *__restrict__ flags,
                                                     float *__restrict__ sym, long long *__restrict__ symidx, long long sym_cap,
                                                     long long g_first = 0 /* a stream segment: the rows in front are history (no entry states) */)
{
    __shared__ __attribute__((aligned(16))) float win[4 * WIN];
    const int lane = threadIdx.x;
    const int mysub = lane >> 4;
    const long long g0 = g_first + (long long)blockIdx.x * 4;
    if (g0 >= n_rows) return;
    long long c_sub[4];
#pragma unroll
    for (int u = 0; u < 4; u++) c_sub[u] = ((g0 + u < n_rows) ? g0 + u : g0) * D.span;      // (groups past the end shadow group g0)
    const long long my_g = (g0 + mysub < n_rows) ? g0 + mysub : g0;
    const GardnerEntry<float> e = entries[my_g];
    GardnerLane L;
    L.ns = e.ns; L.prev = e.prev; L.half = e.half; L.q_last = 0; L.i_last = 0; L.count = 0; L.k = 0;
    L.active = g0 + mysub < n_rows;
    __shared__ float o_val[4 * PDT_GSUB_OUT];
    __shared__ unsigned o_idx[4 * PDT_GSUB_OUT];
    gardner_sub_chunk<WIN, 4, true>(win, in, P, c_sub, L, sym, symidx, e.offset, sym_cap, o_val, o_idx);
    if (L.active && (lane & 15) == 0) {
        unsigned char flag = 1;
        const GardnerSpanRow rw = rows[my_g];
        const unsigned cell = gardner_encode_exit(D, L.q_last, L.i_last, L.count);
        if (rw.n != ~0u && rw.n > 0 && cell != PDT_GTAB_MISS) {
            const unsigned k1 = cell & ((1u << D.idx_bits) - 1u);
            unsigned lo = 0, hi = rw.n;
            while (lo < hi) {
                const unsigned mid = (lo + hi) >> 1;
                if (keys[rw.off + mid] < k1) lo = mid + 1;
                else hi = mid;
            }
            if (lo < rw.n && keys[rw.off + lo] == k1) {
                const GardnerSpanRec *rc = recs + (size_t)(rw.off + lo) * (size_t)(D.span - 1);
                for (int q = 1; q < D.span; q++) {
                    GardnerEntry<float> ce;
                    ce.ns = rc[q - 1].ns; ce.prev = rc[q - 1].prev; ce.half = rc[q - 1].half;
                    ce.offset = e.offset + (long long)L.count + (long long)rc[q - 1].count;
                    centries[my_g * D.span + q] = ce;
                }
                flag = 0;
            }
        }
        flags[my_g] = flag;
    }
}

template <int WIN>
__device__ __forceinline__ void k_gardner_emit_rest(const float *__restrict__ in, GardnerParams<float> P, GardnerDomain D,
                                                    long long n_rows, const GardnerEntry<float> *__restrict__ centries,
                                                    const unsigned char *__restrict__ flags, float *__restrict__ sym,
                                                    long long *__restrict__ symidx, long long sym_cap, long long g_first = 0)
{
    __shared__ __attribute__((aligned(16))) float win[4 * WIN];
    const int lane = threadIdx.x;
    const int mysub = lane >> 4;
    const long long per = D.span - 1;                       // chunks of a group that are not its first
    const long long total = n_rows * per;
    const long long i0 = g_first * per + (long long)blockIdx.x * 4;
    if (i0 >= total) return;
    // the chunk of every sub-group; one whose group is flagged (or past the end) shadows a chunk that is walked -- if there is none
    // in this wavefront, there is nothing to do
    long long c_sub[4];
    bool live[4];
    long long c_any = -1;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const long long i = i0 + u;
        const long long g = (i < total) ? i / per : 0;
        live[u] = i < total && !flags[g];
        c_sub[u] = g * D.span + 1 + (i - g * per);
        if (live[u] && c_any < 0) c_any = c_sub[u];
    }
    if (c_any < 0) return;
#pragma unroll
    for (int u = 0; u < 4; u++) c_sub[u] = live[u] ? c_sub[u] : c_any;
    long long my_c = c_any;
    bool mine_live = false;
#pragma unroll
    for (int u = 0; u < 4; u++)
        if (mysub == u) { my_c = c_sub[u]; mine_live = live[u]; }
    const GardnerEntry<float> e = centries[my_c];
    GardnerLane L;
    L.ns = e.ns; L.prev = e.prev; L.half = e.half; L.q_last = 0; L.i_last = 0; L.count = 0; L.k = 0;
    L.active = mine_live;
    __shared__ float o_val[4 * PDT_GSUB_OUT];
    __shared__ unsigned o_idx[4 * PDT_GSUB_OUT];
    gardner_sub_chunk<WIN, 4, true>(win, in, P, c_sub, L, sym, symidx, e.offset, sym_cap, o_val, o_idx);
}

__device__ __forceinline__ void k_gardner_span_join(GardnerDomain D, long long n_rows, const unsigned *__restrict__ cand_k,
                                                    const GardnerBand *__restrict__ bands, const unsigned *__restrict__ clist,
                                                    unsigned *__restrict__ table, const unsigned *__restrict__ keys,
                                                    const unsigned *__restrict__ tails, const GardnerSpanRow *__restrict__ rows,
                                                    unsigned *__restrict__ stats /* [0] bad */)
{
    const long long r = blockIdx.x;
    if (r >= n_rows) return;
    const GardnerSpanRow rw = rows[r];
    if (rw.n == ~0u) return;
    const long long c0 = r * D.span;
    const GardnerBand bd = bands[r];
    const unsigned *cand = bd.listed ? (clist + (size_t)r * PDT_GTAB_LIST) : cand_k;
    unsigned *row = table + (size_t)r * (size_t)(2 * D.n_q);
    const unsigned kmask = (1u << D.idx_bits) - 1u;
    for (int j = bd.j_lo + (int)threadIdx.x; j < bd.j_hi; j += (int)blockDim.x) {
        const unsigned k = (c0 == 0) ? 0u : cand[j];
        const unsigned cell = row[k];
        if (cell == PDT_GTAB_MISS) continue;
        const unsigned k1 = cell & kmask;
        unsigned lo = 0, hi = rw.n;                                      // first key >= k1 (it is there)
        while (lo < hi) {
            const unsigned mid = (lo + hi) >> 1;
            if (keys[rw.off + mid] < k1) lo = mid + 1;
            else hi = mid;
        }
        unsigned out = PDT_GTAB_MISS;
        if (lo < rw.n && keys[rw.off + lo] == k1) {
            const unsigned tail = tails[rw.off + lo];
            if (tail != PDT_GTAB_MISS) {
                const unsigned cnt = (cell >> D.idx_bits) + (tail >> D.idx_bits);
                if (cnt < (1u << (32 - D.idx_bits)) - 1u) out = (tail & kmask) | (cnt << D.idx_bits);
            }
        }
        if (out == PDT_GTAB_MISS) atomicAdd(&stats[0], 1u);
        row[k] = out;
    }
}

__device__ __forceinline__ unsigned gardner_cell(const unsigned *__restrict__ table, size_t stride, const GardnerBand &b,
                                                 long long c, unsigned key)
{
    const unsigned cell = table[(size_t)c * stride + key];          // valid address for every key < 2 n_q; content only inside the band
    return (key >= b.k_lo && key <= b.k_hi) ? cell : PDT_GTAB_MISS;
}

// level 2: follow the chain  k_{c+1} = table_c[k_c]  (k_0 = 0).  A chain of n dependent HBM lookups
// would cost ~0.5 us each, so it is cut into segments of G chunks:
//   k_gardner_segmap   composes, for every tabulated entry state of a segment's first chunk, the G
//                      table steps of the segment (parallel over segments x states);
//   k_gardner_chain    one wavefront hops from segment to segment through those composite maps;
//                      where a hop is impossible (state not tabulated, end of the capture) it
//                      falls back to chunk-by-chunk stepping and simply WALKS a chunk whose entry
//                      state was outside the scouts' band;
//   k_gardner_segfill  re-traces every hopped segment from its now known start, in parallel,
//                      to produce the per-chunk entry states and symbol offsets.
struct GardnerSegCell { unsigned next, count; };
struct GardnerSegStart { unsigned key; unsigned hopped; long long offset; };

__device__ __forceinline__ void k_gardner_segmap(const unsigned *__restrict__ table, GardnerDomain D, long long n_groups,
                                                          int G, GardnerSegCell *__restrict__ segmap,
                                                          const GardnerBand *__restrict__ bands)
{
    // (everything here is in units of table rows = groups of D.span chunks; n_groups = rows + 1, the last group has no row)
    const long long s = blockIdx.x;
    const long long c0 = s * G;
    if (c0 + G > n_groups - 1) return;                  // only whole segments whose groups all have a table row
    const size_t stride = (size_t)(2 * D.n_q);
    __shared__ unsigned s_klo[64], s_khi[64];
    if ((int)threadIdx.x < G) {
        const GardnerBand b = bands[c0 + threadIdx.x];
        s_klo[threadIdx.x] = b.k_lo;
        s_khi[threadIdx.x] = b.k_hi;
    }
    __syncthreads();
    // entry keys of the segment's first chunk: its band only (everything else is a miss before the first step).  A segment that
    // starts on a row whose scouts did not settle (the whole domain listed: tens of thousands of keys, G dependent look-ups each)
    // is not composed: the chain steps through it row by row
    const bool wide = s_khi[0] - s_klo[0] > 4096u;
    for (unsigned k0 = s_klo[0] + threadIdx.x; k0 <= s_khi[0]; k0 += 1024u) {
        unsigned k = k0, total = 0;
        bool ok = !wide;
        for (int g = 0; g < G && ok; g++) {
            const unsigned cell = (k >= s_klo[g] && k <= s_khi[g]) ? table[(size_t)(c0 + g) * stride + k] : PDT_GTAB_MISS;
            if (cell == PDT_GTAB_MISS) { ok = false; break; }
            k = cell & ((1u << D.idx_bits) - 1u);
            total += cell >> D.idx_bits;
        }
        GardnerSegCell sc;
        sc.next = ok ? k : PDT_GTAB_MISS;
        sc.count = total;
        segmap[(size_t)s * stride + (size_t)k0] = sc;
    }
}

// where the chain stands between two launches of k_gardner_chain (long captures run it range by range, so that the entry
// states and the symbols of one range are produced -- segfill, emission, on the side stream -- while the chain hops on)
struct GardnerChainState {
    long long c, off;
    unsigned key, walked, i_last;
    int have_key;
    float ns, prev, half, q_last;
};

__device__ __forceinline__ void k_gardner_chain(const float *__restrict__ in, GardnerParams<float> P,
                                                                        GardnerDomain D, long long n_chunks /* groups of D.span chunks: `chunk` below = group */,
                                                                        const unsigned *__restrict__ table,
                                                                        const GardnerSegCell *__restrict__ segmap, int G,
                                                                        GardnerSegStart *__restrict__ segstart,
                                                                        GardnerEntry<float> *__restrict__ entries,
                                                                        unsigned *__restrict__ stats /* [2] walked chunks */,
                                                                        const GardnerBand *__restrict__ bands, long long n_tab,
                                                                        SamplerCarry<float> carry, int have_carry,
                                                                        long long c_stop /* a multiple of G, or n_chunks */,
                                                                        GardnerChainState *__restrict__ state, int first,
                                                                        const unsigned *__restrict__ span_keys /* D.span > 1: the rows' */,
                                                                        const unsigned *__restrict__ span_tails /* distinct first-chunk exits */,
                                                                        const GardnerSpanRow *__restrict__ span_rows /* and where they lead */)
{
    __shared__ float win[GardnerLds<float>::LEN];
    if (first) {   // statistics of the scouts: candidates evaluated ([3]) and chunks tabulated over the full domain ([1])
        __shared__ unsigned s_sum[2];
        if (threadIdx.x < 2) s_sum[threadIdx.x] = 0;
        __syncthreads();
        unsigned cand = 0, full = 0;
        for (long long k = threadIdx.x; k < n_tab; k += blockDim.x) {
            const GardnerBand b = bands[k];
            cand += (unsigned)(b.j_hi - b.j_lo);
            full += (k >= 1 && !b.listed) ? 1u : 0u;
        }
        atomicAdd(&s_sum[0], cand);
        atomicAdd(&s_sum[1], full);
        __syncthreads();
        if (threadIdx.x == 0) { stats[3] = s_sum[0]; stats[1] = s_sum[1]; }
    }
    const size_t stride = (size_t)(2 * D.n_q);
    GardnerState<float> S;
    S.ns = 0; S.prev = 0; S.half = 0; S.q_last = 0; S.i_last = 0;
    long long off = carry.count0;
    unsigned walked = 0, key = 0;
    bool have_key = true;                                // chunk 0: the single start state is cell 0 of row 0
    long long c = carry.c_first;
    if (have_carry) {
        // a stream segment goes on from the sampler state the previous one left: not a tabulated boundary state as such, so the
        // first chunk is walked and the chain takes to the tables from its exit on
        S.ns = carry.a; S.prev = carry.b; S.half = carry.c;
        have_key = false;
    }
    if (!first) {
        const GardnerChainState st = *state;
        c = st.c; off = st.off; key = st.key; walked = st.walked; have_key = st.have_key != 0;
        S.ns = st.ns; S.prev = st.prev; S.half = st.half; S.q_last = st.q_last; S.i_last = st.i_last;
    }
    bool finished = false;
    while (c < n_chunks && c < c_stop) {
        // ---- hop over a whole segment
        if (have_key && (c % G) == 0 && c + G <= n_chunks - 1 && c + G <= c_stop) {
            const long long s = c / G;
            // (both loads issued together: the cell's address is valid whatever the key, its content only inside the band)
            const GardnerBand b0 = bands[c];
            const GardnerSegCell sc = segmap[(size_t)s * stride + key];
            const unsigned nxt = uniform<unsigned>((key >= b0.k_lo && key <= b0.k_hi) ? sc.next : PDT_GTAB_MISS);
            if (nxt != PDT_GTAB_MISS) {
                if (threadIdx.x == 0) {
                    GardnerSegStart ss;
                    ss.key = key; ss.hopped = 1; ss.offset = off;
                    segstart[s] = ss;
                }
                key = nxt;
                off += (long long)uniform<unsigned>(sc.count);
                c += G;
                continue;
            }
        }
        // ---- one chunk
        if (have_key && c >= 1) gardner_entry_from_candidate(in, P, D, c * D.span, (int)key, S.ns, S.prev, S.half);
        if (threadIdx.x == 0) {
            GardnerEntry<float> e;
            e.ns = S.ns; e.prev = S.prev; e.half = S.half; e.offset = off;
            entries[c] = e;
        }
        if (c + 1 >= n_chunks) { finished = true; break; }
        unsigned cell = have_key ? gardner_cell(table, stride, bands[c], c, key) : PDT_GTAB_MISS;
        cell = uniform<unsigned>(cell);
        if (cell == PDT_GTAB_MISS) {
            long long cnt = 0;                               // (a group that has a successor consists of full chunks)
            bool through = false;
            for (long long cc = c * D.span; cc < (c + 1) * D.span && !through; cc++) {
                cnt += gardner_walk_chunk<float, false, GardnerLds<float>::LEN, GardnerLds<float>::OUT>(
                    in, (const float *)nullptr, P, cc, S, win, (float *)nullptr, (unsigned *)nullptr, (float *)nullptr, (long long *)nullptr, 0, 0);
                if (cc == c * D.span && D.span > 1 && span_rows && c < n_tab) {
                    // a row of several chunks: once through its first chunk the trajectory is, as a rule, one of the handful the
                    // span kernels walked on from there -- the rest of the row is then one look-up
                    const GardnerSpanRow rw = span_rows[c];
                    const unsigned c1 = gardner_encode_exit(D, S.q_last, S.i_last, (unsigned)cnt);
                    if (rw.n != ~0u && rw.n > 0 && c1 != PDT_GTAB_MISS) {
                        const unsigned k1 = c1 & ((1u << D.idx_bits) - 1u);
                        unsigned lo = 0, hi = rw.n;
                        while (lo < hi) {
                            const unsigned mid = (lo + hi) >> 1;
                            if (span_keys[rw.off + mid] < k1) lo = mid + 1;
                            else hi = mid;
                        }
                        if (lo < rw.n && span_keys[rw.off + lo] == k1) {
                            const unsigned tail = span_tails[rw.off + lo];
                            const unsigned total = (unsigned)cnt + (tail >> D.idx_bits);
                            if (tail != PDT_GTAB_MISS && total < (1u << (32 - D.idx_bits)) - 1u) {
                                cell = uniform<unsigned>((tail & ((1u << D.idx_bits) - 1u)) | (total << D.idx_bits));
                                through = true;
                            }
                        }
                    }
                }
            }
            walked++;
            if (!through) cell = gardner_encode_exit(D, S.q_last, S.i_last, (unsigned)cnt);
            if (cell == PDT_GTAB_MISS) {
                // the exit is not a tabulated boundary state (irregular geometry): keep walking
                off += cnt;
                have_key = false;
                c++;
                continue;
            }
        }
        key = cell & ((1u << D.idx_bits) - 1u);
        off += (long long)(cell >> D.idx_bits);
        have_key = true;
        c++;
    }
    if (threadIdx.x == 0) {
        if (finished || c >= n_chunks) stats[2] = walked;
        GardnerChainState st;
        st.c = c; st.off = off; st.key = key; st.walked = walked; st.have_key = have_key ? 1 : 0;
        st.ns = S.ns; st.prev = S.prev; st.half = S.half; st.q_last = S.q_last; st.i_last = S.i_last;
        *state = st;
    }
}

__device__ __forceinline__ void k_gardner_segfill(const float *__restrict__ in, GardnerParams<float> P, GardnerDomain D,
                                                         long long n_chunks, const unsigned *__restrict__ table, int G,
                                                         const GardnerSegStart *__restrict__ segstart,
                                                         GardnerEntry<float> *__restrict__ entries, long long seg_first)
{
    const long long s = blockIdx.x + seg_first;
    const long long c0 = s * G;
    if (c0 + G > n_chunks - 1) return;
    const GardnerSegStart ss = segstart[s];
    if (!ss.hopped) return;
    // 64 lanes, G <= 64 chunks: the key chain is serial (G dependent lookups), the entry states are not
    const size_t stride = (size_t)(2 * D.n_q);
    __shared__ unsigned s_key[64];
    __shared__ long long s_off[64];
    if (threadIdx.x == 0) {
        unsigned k = ss.key;
        long long off = ss.offset;
        for (int g = 0; g < G; g++) {
            s_key[g] = k;
            s_off[g] = off;
            const unsigned cell = table[(size_t)(c0 + g) * stride + k];
            k = cell & ((1u << D.idx_bits) - 1u);
            off += (long long)(cell >> D.idx_bits);
        }
    }
    __syncthreads();
    const int g = threadIdx.x;
    if (g < G) {
        const long long c = c0 + g;
        GardnerEntry<float> e;
        e.ns = 0; e.prev = 0; e.half = 0;
        if (c >= 1) gardner_entry_from_candidate(in, P, D, c * D.span, (int)s_key[g], e.ns, e.prev, e.half);
        e.offset = s_off[g];
        entries[c] = e;
    }
}

// ------------------------------------------------------------------------------------------
// Manchester decision (reference: common/ManchesterDecode.c:10-100)
//
// With q_i = parity of the global symbol index i, pp = sym[i-2], p = sym[i-1], cur = sym[i]
// (0 before the stream start) and R_i = sign(pp)==sign(p) && |pp|>thr && |p|>thr, the
// reference's clockmod after symbol i is q_j for the last j <= i with R_j (initially 0),
// and a bit is emitted at i iff q_i == clockmod_i.  That is a "last flagged position" scan
// plus an ordered compaction: three passes over 4096-symbol tiles.
// ------------------------------------------------------------------------------------------
#define PDT_TILE 4096
#define PDT_TILE_THREADS 256

template <typename T> __device__ __forceinline__ int sgn(T x) { return (x > 0) - (x < 0); }

template <typename T> __device__ __forceinline__ bool manch_resync(const T *sym, long long i, T thr)
{
    const T pp = (i >= 2) ? sym[i - 2] : (T)0;
    const T p = (i >= 1) ? sym[i - 1] : (T)0;
    return sgn(pp) == sgn(p) && Real<T>::abs(pp) > thr && Real<T>::abs(p) > thr;
}

// ---- helpers of the Manchester tile kernels: the tile's symbols are staged once in LDS (coalesced loads; one pad word
// per 16 so that threads walking 16 consecutive symbols each hit distinct banks), and the two cross-thread
// prefixes (last resync position, emitted count) are block scans instead of one thread's loop over 256 entries.
#define PDT_MANCH_LDS ((PDT_TILE + 2) + (PDT_TILE + 2) / 16 + 1)
__device__ __forceinline__ int manch_at(int k) { return k + (k >> 4); }
// s_sym[manch_at(k)] = sym[t0 - 2 + k], k in [0, PDT_TILE + 2); 0 outside [0, nsym)
template <typename T>
__device__ __forceinline__ void manch_stage(const T *__restrict__ sym, long long t0, long long nsym, T *s_sym)
{
    for (int k = threadIdx.x; k < PDT_TILE + 2; k += PDT_TILE_THREADS) {
        const long long gi = t0 - 2 + k;
        s_sym[manch_at(k)] = (gi >= 0 && gi < nsym) ? sym[gi] : (T)0;
    }
}
// resync test of tile-relative symbol j (manch_resync on the staged copy)
template <typename T> __device__ __forceinline__ bool manch_resync_lds(const T *s_sym, int j, T thr)
{
    const T pp = s_sym[manch_at(j)];
    const T p = s_sym[manch_at(j + 1)];
    return sgn(pp) == sgn(p) && Real<T>::abs(pp) > thr && Real<T>::abs(p) > thr;
}
// exclusive block scans over the 256 threads (4 wavefronts); s_w: 4 ints of LDS
__device__ __forceinline__ int manch_scan_max_excl(int v, int *s_w)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc = (o > inc) ? o : inc;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int ex = __shfl_up(inc, 1);
    if (lane == 0) ex = -1;
    for (int w = 0; w < wave; w++) ex = (s_w[w] > ex) ? s_w[w] : ex;
    __syncthreads();
    return ex;
}
__device__ __forceinline__ unsigned manch_scan_add_excl(unsigned v, unsigned *s_w)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    unsigned ex = inc - v;
    for (int w = 0; w < wave; w++) ex += s_w[w];
    __syncthreads();
    return ex;
}

struct ManchTile {
    int last_r_parity;      // parity of the last resync position in the tile, -1 = none
    unsigned first_r;       // tile-relative index of the first resync position, PDT_TILE = none
    unsigned cnt_before[2]; // bits emitted before first_r when the incoming clockmod is 0 / 1
    unsigned cnt_after;     // bits emitted from first_r on
    unsigned clock_in;      // filled by the scan pass
    unsigned long long out_base;
};

// block-wide inclusive max-scan of (position of last flagged symbol) in LDS
template <typename T>
__device__ __forceinline__ void k_manch_tile(const T *__restrict__ sym,
                                                                  const unsigned long long *__restrict__ nsym_p, T thr,
                                                                  ManchTile *__restrict__ tiles, long long i0)
{
    // (i0: first symbol to decide -- 0 for a capture; in a stream segment the symbols before it are the history the
    // decisions look back on, and their index parity equals that of their position in the whole stream)
    const long long nsym = (long long)*nsym_p;
    const long long t0 = i0 + (long long)blockIdx.x * PDT_TILE;
    if (t0 >= nsym) return;
    __shared__ int s_last[PDT_TILE];     // tile-relative index of last R at or before i, -1 none
    __shared__ T s_sym[PDT_MANCH_LDS];
    __shared__ unsigned s_first;
    __shared__ unsigned s_cnt[3];
    __shared__ int s_w[4];
    if (threadIdx.x == 0) { s_first = PDT_TILE; s_cnt[0] = s_cnt[1] = s_cnt[2] = 0; }
    manch_stage(sym, t0, nsym, s_sym);
    __syncthreads();
    const int per = PDT_TILE / PDT_TILE_THREADS;            // 16 consecutive symbols per thread
    const int lo = threadIdx.x * per;
    int last = -1;
    for (int u = 0; u < per; u++) {
        const long long i = t0 + lo + u;
        if (i < nsym && manch_resync_lds(s_sym, lo + u, thr)) last = lo + u;
        s_last[lo + u] = last;
    }
    // propagate across threads: thread t needs the last R of all previous threads (positions grow with t: a max scan)
    __shared__ int s_tlast[PDT_TILE_THREADS];
    const int carry = manch_scan_max_excl(last, s_w);
    s_tlast[threadIdx.x] = carry;
    __syncthreads();
    unsigned c_b0 = 0, c_b1 = 0, c_a = 0;
    unsigned my_first = PDT_TILE;
    for (int u = 0; u < per; u++) {
        const long long i = t0 + lo + u;
        if (i >= nsym) break;
        int l = s_last[lo + u];
        if (l < 0) l = carry;
        const unsigned q = (unsigned)(i & 1);
        if (l < 0) {                       // clockmod still the incoming one
            c_b0 += (q == 0);
            c_b1 += (q == 1);
        } else {
            if (my_first == PDT_TILE) my_first = (unsigned)l;
            const unsigned cm = (unsigned)((t0 + l) & 1);
            c_a += (q == cm);
        }
    }
    // tile totals: wavefront reductions by shuffles, then one LDS atomic per wavefront (256 threads each adding to the
    // same four words took longer than the rest of the kernel)
    for (int d = 32; d >= 1; d >>= 1) {
        c_b0 += __shfl_xor(c_b0, d);
        c_b1 += __shfl_xor(c_b1, d);
        c_a += __shfl_xor(c_a, d);
        const unsigned of = __shfl_xor(my_first, d);
        my_first = (of < my_first) ? of : my_first;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_cnt[0], c_b0);
        atomicAdd(&s_cnt[1], c_b1);
        atomicAdd(&s_cnt[2], c_a);
        atomicMin(&s_first, my_first);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ManchTile mt;
        const long long t1 = (t0 + PDT_TILE < nsym) ? t0 + PDT_TILE : nsym;
        const int l_end = s_last[(int)(t1 - t0 - 1)] >= 0 ? s_last[(int)(t1 - t0 - 1)] : s_tlast[(int)((t1 - t0 - 1) / per)];
        mt.last_r_parity = (l_end >= 0) ? (int)((t0 + l_end) & 1) : -1;
        mt.first_r = s_first;
        mt.cnt_before[0] = s_cnt[0];
        mt.cnt_before[1] = s_cnt[1];
        mt.cnt_after = s_cnt[2];
        mt.clock_in = 0;
        mt.out_base = 0;
        tiles[blockIdx.x] = mt;
    }
}

// pass over the tile summaries: one wavefront, 64 tiles per round trip; the (clockmod, output
// offset) chain itself is evaluated redundantly by all lanes from shuffled values
// A tile acts on the running (clockmod, bit count) as a function of the incoming clockmod c in {0,1}:
// clockmod' = o[c], count += a[c].  Such maps compose associatively, so the tile chain is a block-wide
// scan (1024 tiles per round, Hillis-Steele in LDS) instead of a walk.
struct ManchMap { unsigned o0, o1; unsigned long long a0, a1; };
__device__ __forceinline__ ManchMap manch_compose(const ManchMap &f, const ManchMap &g)   // f first, then g
{
    ManchMap r;
    r.o0 = f.o0 ? g.o1 : g.o0;
    r.o1 = f.o1 ? g.o1 : g.o0;
    r.a0 = f.a0 + (f.o0 ? g.a1 : g.a0);
    r.a1 = f.a1 + (f.o1 ? g.a1 : g.a0);
    return r;
}

__device__ __forceinline__ void k_manch_scan(ManchTile *__restrict__ tiles, const unsigned long long *__restrict__ nsym_p,
                                                      unsigned long long *__restrict__ nbits_out, long long i0, unsigned clock0,
                                                      unsigned long long bit0, unsigned *__restrict__ clock_out)
{
    __shared__ ManchMap s_map[1024];
    __shared__ unsigned s_clock;
    __shared__ unsigned long long s_base;
    const long long nsym = (long long)*nsym_p;
    const long long nt = (nsym > i0) ? (nsym - i0 + PDT_TILE - 1) / PDT_TILE : 0;
    if (threadIdx.x == 0) { s_clock = clock0; s_base = bit0; }      // bit0: bits carried in front of this segment's
    __syncthreads();
    for (long long t0 = 0; t0 < nt; t0 += 1024) {
        const long long mine = t0 + threadIdx.x;
        ManchMap m;
        m.o0 = 0; m.o1 = 1; m.a0 = 0; m.a1 = 0;                       // identity for the padding lanes
        if (mine < nt) {
            const ManchTile mt = tiles[mine];
            m.a0 = (unsigned long long)mt.cnt_before[0] + mt.cnt_after;
            m.a1 = (unsigned long long)mt.cnt_before[1] + mt.cnt_after;
            if (mt.last_r_parity >= 0) m.o0 = m.o1 = (unsigned)mt.last_r_parity;
        }
        s_map[threadIdx.x] = m;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {                          // inclusive scan
            ManchMap left;
            const bool has = (int)threadIdx.x >= d;
            if (has) left = s_map[threadIdx.x - d];
            __syncthreads();
            if (has) s_map[threadIdx.x] = manch_compose(left, s_map[threadIdx.x]);
            __syncthreads();
        }
        const unsigned clock0 = s_clock;
        const unsigned long long base0 = s_base;
        // state entering my tile = carry applied through the maps of the tiles before it
        unsigned my_clock = clock0;
        unsigned long long my_base = base0;
        if (threadIdx.x > 0) {
            const ManchMap p = s_map[threadIdx.x - 1];
            my_clock = clock0 ? p.o1 : p.o0;
            my_base = base0 + (clock0 ? p.a1 : p.a0);
        }
        if (mine < nt) {
            tiles[mine].clock_in = my_clock;
            tiles[mine].out_base = my_base;
        }
        const ManchMap all = s_map[1023];
        __syncthreads();
        if (threadIdx.x == 0) {
            s_clock = clock0 ? all.o1 : all.o0;
            s_base = base0 + (clock0 ? all.a1 : all.a0);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *nbits_out = s_base;
        if (clock_out) *clock_out = s_clock;
    }
}

template <typename T>
__device__ __forceinline__ void k_manch_emit(const T *__restrict__ sym,
                                                                  const unsigned long long *__restrict__ nsym_p, T thr,
                                                                  const ManchTile *__restrict__ tiles,
                                                                  unsigned char *__restrict__ bits,
                                                                  unsigned *__restrict__ bitsym, long long bit_cap, long long i0)
{
    const long long nsym = (long long)*nsym_p;
    const long long t0 = i0 + (long long)blockIdx.x * PDT_TILE;
    if (t0 >= nsym) return;
    const ManchTile mt = tiles[blockIdx.x];
    __shared__ T s_sym[PDT_MANCH_LDS];
    __shared__ int s_wi[4];
    __shared__ unsigned s_wu[4];
    manch_stage(sym, t0, nsym, s_sym);
    __syncthreads();
    const int per = PDT_TILE / PDT_TILE_THREADS;
    const int lo = threadIdx.x * per;
    // pass 1: per-thread last-R and (given the carry) emitted count -- two sweeps over 16 symbols
    int last = -1;
    unsigned rs_mask = 0;                 // resync test of the thread's 16 symbols, kept for the second sweep
    for (int u = 0; u < per; u++) {
        const long long i = t0 + lo + u;
        if (i < nsym && manch_resync_lds(s_sym, lo + u, thr)) { last = lo + u; rs_mask |= 1u << u; }
    }
    const int carry = manch_scan_max_excl(last, s_wi);
    unsigned clock = (carry >= 0) ? (unsigned)((t0 + carry) & 1) : mt.clock_in;
    unsigned emit_mask = 0;
    unsigned my_cnt = 0;
    for (int u = 0; u < per; u++) {
        const long long i = t0 + lo + u;
        if (i >= nsym) break;
        const unsigned q = (unsigned)(i & 1);
        if (rs_mask & (1u << u)) clock = q;
        if (q == clock) { emit_mask |= 1u << u; my_cnt++; }
    }
    unsigned long long o = mt.out_base + manch_scan_add_excl(my_cnt, s_wu);
    for (int u = 0; u < per; u++) {
        if (!(emit_mask & (1u << u))) continue;
        const long long i = t0 + lo + u;
        const T p = s_sym[manch_at(lo + u + 1)];          // sym[i - 1] (0 before the stream start)
        const T cur = s_sym[manch_at(lo + u + 2)];        // sym[i]
        unsigned char bit;
        if (Real<T>::abs(p) > Real<T>::abs(cur))
            bit = (p > 0) ? '1' : '0';
        else
            bit = (cur > 0) ? '0' : '1';
        if ((long long)o < bit_cap) {
            bits[o] = bit;
            bitsym[o] = (unsigned)i;
        }
        o++;
    }
}

// ------------------------------------------------------------------------------------------
// Sync-word search + frame extraction
// (reference: POESTIPdemod/ByteSync.c:16-150, ARGOSdemod/ByteSync.c:17-150)
//
// Pass 1 marks every bit position whose trailing `len` bits equal the sync word or
// (POES) its complement; history before the first bit is '0' (ByteSync.c:39).
// Pass 2 (one lane, hits are ~1 per 832 bits) applies the "not already inside a frame"
// rule: a frame opened at bit p absorbs `span` following bits, and a new frame may open
// at p + span at the earliest (the in-frame flag is cleared before the sync test of that
// same bit, ByteSync.c:63-67,93).  Pass 3 packs each frame's bytes in parallel.
// ------------------------------------------------------------------------------------------
struct SyncParams {
    unsigned long long pattern;    // sync word, first bit = MSB of the low `len` bits
    unsigned len;
    unsigned allow_inverse;
    unsigned span;                 // bits after the sync bit until the frame closes: 813 POES, 56 ARGOS
    unsigned first_bits;           // bits of the first (partial) byte: 5 POES (bitIdx starts at 3), 8 ARGOS
    unsigned nbytes;               // payload bytes shifted in: 102 POES, 7 ARGOS
    unsigned prefix;               // literal bytes printed first: 2 POES (ED E2), 0 ARGOS
};

__device__ __forceinline__ void k_sync_hits(const unsigned char *__restrict__ bits,
                                                    const unsigned long long *__restrict__ nbits_p, SyncParams P,
                                                    unsigned *__restrict__ hits, unsigned *__restrict__ nhits,
                                                    unsigned hit_cap, const unsigned *__restrict__ only_if, long long min_pos)
{
    if (only_if && *only_if == 0) return;            // generic path: only when the tile path overflowed
    const long long nbits = (long long)*nbits_p;
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbits || b < min_pos) return;           // (min_pos: a stream segment's window begins inside a frame already reported)
    unsigned long long w = 0;
    for (unsigned k = 0; k < P.len; k++) {
        const long long idx = b - (long long)(P.len - 1) + k;
        const unsigned v = (idx >= 0) ? (unsigned)(bits[idx] != '0') : 0u;
        w = (w << 1) | v;
    }
    const unsigned long long mask = (P.len >= 64) ? ~0ull : ((1ull << P.len) - 1ull);
    unsigned kind = 0;
    if (w == P.pattern)
        kind = 1;
    else if (P.allow_inverse && w == (~P.pattern & mask))
        kind = 2;
    if (kind) {
        const unsigned slot = atomicAdd(nhits, 1u);
        if (slot < hit_cap) hits[slot] = ((unsigned)b << 1) | (kind - 1);   // bit index < 2^31
    }
}

// Ordered hit collection: one 128-byte record per 4096-bit tile (hits are ~5 per tile: one per
// 832-bit frame plus the odd chance match), sorted inside the tile, so that the frame filter can
// walk tiles in order without a global sort.  More than 31 hits in a tile raises `overflow` and
// the generic path (atomic append + bitonic sort) takes over.
struct SyncTile {
    unsigned count;
    unsigned hits[31];     // (bit index << 1) | inverse
};

__device__ __forceinline__ void k_sync_hits_tile(const unsigned char *__restrict__ bits,
                                                         const unsigned long long *__restrict__ nbits_p, SyncParams P,
                                                         SyncTile *__restrict__ tiles, unsigned *__restrict__ overflow, long long min_pos)
{
    const long long nbits = (long long)*nbits_p;
    const long long t0 = (long long)blockIdx.x * 4096;
    if (t0 >= nbits) return;
    __shared__ unsigned s_hits[64];
    __shared__ unsigned s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const unsigned long long mask = (P.len >= 64) ? ~0ull : ((1ull << P.len) - 1ull);
    const long long b0 = t0 + (long long)threadIdx.x * 16;
    // window ending at b0-1
    unsigned long long w = 0;
    for (unsigned k = 0; k + 1 < P.len; k++) {
        const long long idx = b0 - (long long)(P.len - 1) + k;
        const unsigned v = (idx >= 0 && idx < nbits) ? (unsigned)(bits[idx] != '0') : 0u;
        w = (w << 1) | v;
    }
    for (int u = 0; u < 16; u++) {
        const long long b = b0 + u;
        if (b >= nbits) break;
        w = ((w << 1) | (unsigned long long)(bits[b] != '0')) & mask;
        unsigned kind = 0;
        if (w == P.pattern) kind = 1;
        else if (P.allow_inverse && w == (~P.pattern & mask)) kind = 2;
        if (b < min_pos) kind = 0;
        if (kind) {
            const unsigned slot = atomicAdd(&s_n, 1u);
            if (slot < 64) s_hits[slot] = ((unsigned)b << 1) | (kind - 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned nh = s_n;
        if (nh > 31) { atomicAdd(overflow, 1u); nh = (nh > 64) ? 64 : nh; }
        // insertion sort (a handful of entries)
        for (unsigned i = 1; i < nh; i++) {
            const unsigned v = s_hits[i];
            int j = (int)i - 1;
            while (j >= 0 && s_hits[j] > v) { s_hits[j + 1] = s_hits[j]; j--; }
            s_hits[j + 1] = v;
        }
        SyncTile tl;
        tl.count = (nh > 31) ? 31 : nh;
        for (unsigned i = 0; i < 31; i++) tl.hits[i] = (i < nh) ? s_hits[i] : 0u;
        tiles[blockIdx.x] = tl;
    }
}

struct FrameRec {
    long long bit_index;
    long long time_src;
    unsigned char inverted, nbytes, complete, pad;
    unsigned char bytes[104];
};

// single workgroup: sort the (sparse, unordered) hit list, then walk it sequentially
__device__ __forceinline__ void k_sync_frames(unsigned *__restrict__ hits, const unsigned *__restrict__ nhits_p,
                                                      unsigned hit_cap, SyncParams P, FrameRec *__restrict__ frames,
                                                      unsigned *__restrict__ nframes, unsigned frame_cap,
                                                      const unsigned *__restrict__ only_if)
{
    if (only_if && *only_if == 0) return;
    unsigned nh = *nhits_p;
    if (nh > hit_cap) nh = hit_cap;
    // odd-even transposition would be O(n^2); hits arrive nearly sorted (atomic order follows
    // block order closely), so use a parallel rank sort in chunks: each thread ranks its
    // elements by counting smaller ones in a window, falling back to a full count.
    // For simplicity and determinism: bitonic sort in global memory over the next power of two.
    unsigned np2 = 1;
    while (np2 < nh) np2 <<= 1;
    for (unsigned i = nh + threadIdx.x; i < np2 && i < hit_cap; i += blockDim.x) hits[i] = 0xffffffffu;
    __syncthreads();
    if (np2 <= hit_cap) {
        for (unsigned k = 2; k <= np2; k <<= 1) {
            for (unsigned j = k >> 1; j > 0; j >>= 1) {
                for (unsigned i = threadIdx.x; i < np2; i += blockDim.x) {
                    const unsigned ixj = i ^ j;
                    if (ixj > i) {
                        const unsigned a = hits[i], b = hits[ixj];
                        const bool up = ((i & k) == 0);
                        if ((a > b) == up) { hits[i] = b; hits[ixj] = a; }
                    }
                }
                __threadfence_block();
                __syncthreads();
            }
        }
    }
    if (threadIdx.x == 0) {
        unsigned nf = 0;
        long long next_free = 0;
        for (unsigned h = 0; h < nh; h++) {
            const unsigned v = hits[h];
            const long long pos = (long long)(v >> 1);
            if (pos < next_free) continue;
            if (nf < frame_cap) {
                frames[nf].bit_index = pos;
                frames[nf].inverted = (unsigned char)(v & 1u);
            }
            nf++;
            next_free = pos + P.span;
        }
        *nframes = nf;
    }
}

// frame filter over the ordered tiles.  One workgroup of 1024:
//  (A) the tile hit lists are compacted, in order, into one dense list (block-wide scan of the counts);
//  (B) the "a hit inside an open frame is ignored" rule is a walk along successor links --
//      succ(a) = first hit at least `span` bits after hit a -- starting at hit 0.  With up to 8192
//      hits the links are found by binary search in LDS and the set reachable from hit 0 by pointer
//      doubling (marks spread through J = succ^(2^k) while J is squared), ~13 rounds instead of a
//      serial pass over every hit; longer lists go through the serial pass in LDS batches.
//  (C) the surviving hits are compacted, in order, into the frame records.
#define PDT_SYNC_BATCH 8192
#define PDT_SYNC_THREADS 1024
__device__ __forceinline__ unsigned sync_block_scan(unsigned v, unsigned *s_scan)   // inclusive, 1024 threads
{
    s_scan[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < PDT_SYNC_THREADS; d <<= 1) {
        const unsigned add = ((int)threadIdx.x >= d) ? s_scan[threadIdx.x - d] : 0u;
        __syncthreads();
        s_scan[threadIdx.x] += add;
        __syncthreads();
    }
    return s_scan[threadIdx.x];
}

__device__ __forceinline__ void k_sync_frames_tiles(const SyncTile *__restrict__ tiles,
                                                            const unsigned long long *__restrict__ nbits_p, SyncParams P,
                                                            unsigned *__restrict__ dense, unsigned dense_cap,
                                                            FrameRec *__restrict__ frames, unsigned *__restrict__ nframes,
                                                            unsigned frame_cap, const unsigned *__restrict__ overflow,
                                                            unsigned *__restrict__ gscr /* 2 (dense_cap + 1) + dense_cap / 32 + 2 words */)
{
    if (*overflow) return;                       // the generic path handles this capture
    const long long nbits = (long long)*nbits_p;
    const long long nt = (nbits + 4095) / 4096;
    __shared__ unsigned s_scan[PDT_SYNC_THREADS];
    __shared__ unsigned s_base;
    __shared__ unsigned s_hits[PDT_SYNC_BATCH];            // positions, later the two link tables (2 x u16)
    __shared__ unsigned s_mark[PDT_SYNC_BATCH / 32];
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    // ---- (A) ordered compaction
    for (long long t0 = 0; t0 < nt; t0 += PDT_SYNC_THREADS) {
        const long long mine = t0 + threadIdx.x;
        const unsigned cnt = (mine < nt) ? tiles[mine].count : 0u;
        const unsigned incl = sync_block_scan(cnt, s_scan);
        const unsigned base = s_base;
        const unsigned off = base + incl - cnt;
        const unsigned total = s_scan[PDT_SYNC_THREADS - 1];
        for (unsigned q = 0; q < cnt; q++)
            if (off + q < dense_cap) dense[off + q] = tiles[mine].hits[q];
        __syncthreads();
        if (threadIdx.x == 0) s_base = base + total;
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    const unsigned nh_all = (s_base < dense_cap) ? s_base : dense_cap;
    // ---- (B), (C) in batches of hits that fit the LDS tables (one batch up to ~14 minutes of POES; an hour takes five): a
    // batch is entered at its first hit at or after the end of the last frame accepted so far -- the one fact it needs
    // from its predecessors -- and leaves the end of its own last frame behind.
    __shared__ unsigned s_last, s_nf;
    __shared__ long long s_next_free;
    if (threadIdx.x == 0) { s_nf = 0; s_next_free = 0; }
    __syncthreads();
    constexpr unsigned BATCH = PDT_SYNC_BATCH - 2;       // both link tables (nh + 1 u16 entries each) fit the staging array
    for (unsigned b0 = 0; b0 < nh_all; b0 += BATCH) {
        const unsigned nh = (nh_all - b0 < BATCH) ? (nh_all - b0) : BATCH;
        const unsigned *dn = dense + b0;
        constexpr int PER = PDT_SYNC_BATCH / PDT_SYNC_THREADS;       // 8 hits per thread
        __syncthreads();
        for (unsigned i = threadIdx.x; i < nh; i += PDT_SYNC_THREADS) s_hits[i] = dn[i] >> 1;
        for (unsigned i = threadIdx.x; i < PDT_SYNC_BATCH / 32; i += PDT_SYNC_THREADS) s_mark[i] = 0u;
        if (threadIdx.x == 0) s_last = 0xffffffffu;
        __syncthreads();
        // entry: first hit of the batch that does not fall into the frame still open
        const long long nfree = s_next_free;
        unsigned entry;
        {
            unsigned lo = 0, hi = nh;
            while (lo < hi) {
                const unsigned mid = (lo + hi) >> 1;
                if ((long long)s_hits[mid] >= nfree) hi = mid; else lo = mid + 1;
            }
            entry = lo;
        }
        if (entry >= nh) continue;                        // (uniform) the whole batch lies inside an open frame
        if (threadIdx.x == 0) s_mark[entry >> 5] = 1u << (entry & 31);
        __syncthreads();
        unsigned succ[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const unsigned i = threadIdx.x + (unsigned)k * PDT_SYNC_THREADS;
            succ[k] = nh;
            if (i < nh) {
                const unsigned want = s_hits[i] + P.span;
                unsigned lo = i + 1, hi = nh;                          // first index in (i, nh) with pos >= want
                while (lo < hi) {
                    const unsigned mid = (lo + hi) >> 1;
                    if (s_hits[mid] >= want) hi = mid; else lo = mid + 1;
                }
                succ[k] = lo;
            }
        }
        __syncthreads();
        unsigned short *J0 = reinterpret_cast<unsigned short *>(s_hits);
        unsigned short *J1 = J0 + PDT_SYNC_BATCH;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const unsigned i = threadIdx.x + (unsigned)k * PDT_SYNC_THREADS;
            if (i < nh) J0[i] = (unsigned short)succ[k];
        }
        if (threadIdx.x == 0) J0[nh] = (unsigned short)nh;              // the end is its own successor
        __syncthreads();
        unsigned short *Jc = J0, *Jn = J1;
        for (unsigned reach = 1; reach < nh; reach <<= 1) {
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const unsigned i = threadIdx.x + (unsigned)k * PDT_SYNC_THREADS;
                if (i <= nh) {
                    const unsigned jm = Jc[i];
                    if (i < nh && jm < nh && ((s_mark[i >> 5] >> (i & 31)) & 1u)) atomicOr(&s_mark[jm >> 5], 1u << (jm & 31));
                    Jn[i] = Jc[jm];
                }
            }
            __syncthreads();
            unsigned short *tmp = Jc; Jc = Jn; Jn = tmp;
        }
        // ---- (C) ordered compaction of the marked hits (thread t owns hits [8t, 8t+8))
        const unsigned i0 = threadIdx.x * PER;
        unsigned mine = 0, my_last = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const unsigned i = i0 + k;
            if (i < nh && ((s_mark[i >> 5] >> (i & 31)) & 1u)) { mine++; my_last = i; }
        }
        const unsigned incl = sync_block_scan(mine, s_scan);
        const unsigned fbase = s_nf;
        unsigned at = fbase + incl - mine;
        for (int k = 0; k < PER; k++) {
            const unsigned i = i0 + k;
            if (i < nh && ((s_mark[i >> 5] >> (i & 31)) & 1u)) {
                if (at < frame_cap) {
                    const unsigned v = dn[i];
                    frames[at].bit_index = (long long)(v >> 1);
                    frames[at].inverted = (unsigned char)(v & 1u);
                }
                at++;
            }
        }
        // the thread that owns the last marked hit publishes where its frame ends
        if (mine > 0 && incl == s_scan[PDT_SYNC_THREADS - 1]) s_last = my_last;      // (several threads may tie only with mine == 0)
        __syncthreads();
        if (threadIdx.x == 0) {
            s_nf = fbase + s_scan[PDT_SYNC_THREADS - 1];
            if (s_last != 0xffffffffu) s_next_free = (long long)(dn[s_last] >> 1) + (long long)P.span;
        }
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0) *nframes = s_nf;
    (void)gscr;
}

__device__ __forceinline__ void k_frame_pack(const unsigned char *__restrict__ bits,
                                                     const unsigned long long *__restrict__ nbits_p,
                                                     const unsigned *__restrict__ bitsym,
                                                     const long long *__restrict__ symidx, SyncParams P,
                                                     FrameRec *__restrict__ frames, const unsigned *__restrict__ nframes_p,
                                                     unsigned frame_cap)
{
    unsigned nf = *nframes_p;
    if (nf > frame_cap) nf = frame_cap;
    const unsigned f = blockIdx.x;
    if (f >= nf) return;
    const long long nbits = (long long)*nbits_p;
    const long long pos = frames[f].bit_index;
    const unsigned inv = frames[f].inverted;
    const unsigned t = threadIdx.x;
    if (t < P.nbytes) {
        // payload byte t: its bits start at pos + 1 + (t ? first_bits + 8*(t-1) : 0)
        const unsigned width = t ? 8u : P.first_bits;
        const long long b0 = pos + 1 + (t ? (long long)P.first_bits + 8ll * (t - 1) : 0ll);
        if (b0 + width <= nbits) {
            unsigned v = 0;
            for (unsigned k = 0; k < width; k++) {
                unsigned bit = (unsigned)(bits[b0 + k] != '0');
                if (inv) bit ^= 1u;
                v = (v << 1) | bit;
            }
            frames[f].bytes[P.prefix + t] = (unsigned char)v;
        }
    }
    if (t == 0) {
        if (P.prefix == 2) {
            frames[f].bytes[0] = 0xED;
            frames[f].bytes[1] = 0xE2;
        }
        // bytes completed before the stream ended
        const long long avail = nbits - (pos + 1);
        unsigned done = 0;
        if (avail >= (long long)P.first_bits) done = 1 + (unsigned)((avail - P.first_bits) / 8);
        if (done > P.nbytes) done = P.nbytes;
        frames[f].nbytes = (unsigned char)(P.prefix + done);
        frames[f].complete = (unsigned char)(done == P.nbytes);
        frames[f].pad = 0;
        frames[f].time_src = symidx[bitsym[pos]];
    }
}

// What a stream segment hands to the next one, gathered at the end of the segment's kernels into one record (one copy back).
#define PDT_SEG_KEEP 1024
template <typename T> struct SegTail {
    SamplerCarry<T> sampler;          // written by the sampler kernel
    T pll_phase, pll_freq;            // PLL state after the window's last sample (seam record of its last block)
    T locksig;                        // lock-detector value of the last sample (when that stream is kept)
    T agc_gain;                       // AGC gain after the window's last output
    T sym_m2, sym_m1;                 // the last two symbols
    unsigned clock;                   // Manchester clockmod after the last symbol (written by k_manch_scan)
    unsigned nkeep;                   // valid entries below = min(bits, PDT_SEG_KEEP), the LAST bits of the segment
    long long src[PDT_SEG_KEEP];      // local interpolated-sample index each came from
    unsigned char bits[PDT_SEG_KEEP];
};

template <typename T, typename SeamP, typename SeamA>
__device__ __forceinline__ void k_seg_tail(const SeamP *__restrict__ seams_pll, long long last_pll, const T *__restrict__ lock,
                                           long long n, const SeamA *__restrict__ seams_agc, long long last_agc,
                                           const T *__restrict__ sym, const unsigned long long *__restrict__ nsym_p,
                                           const unsigned char *__restrict__ bits, const unsigned *__restrict__ bitsym,
                                           const long long *__restrict__ symidx, const unsigned long long *__restrict__ nbits_p,
                                           long long first_bit_with_source, SegTail<T> *__restrict__ out)
{
    const long long nsym = (long long)*nsym_p, nbits = (long long)*nbits_p;
    if (threadIdx.x == 0) {
        if (last_pll >= 0) { out->pll_phase = seams_pll[last_pll].phase1; out->pll_freq = seams_pll[last_pll].freq1; }
        out->locksig = (lock && n > 0) ? lock[n - 1] : (T)0;
        out->agc_gain = (last_agc >= 0) ? seams_agc[last_agc].g1 : (T)0;
        out->sym_m2 = (nsym >= 2) ? sym[nsym - 2] : (T)0;
        out->sym_m1 = (nsym >= 1) ? sym[nsym - 1] : (T)0;
        out->nkeep = (unsigned)((nbits < PDT_SEG_KEEP) ? nbits : PDT_SEG_KEEP);
    }
    const long long keep = (nbits < PDT_SEG_KEEP) ? nbits : PDT_SEG_KEEP;
    const long long b0 = nbits - keep;
    for (long long k = threadIdx.x; k < keep; k += blockDim.x) {
        out->bits[k] = bits[b0 + k];
        // (bits before first_bit_with_source were carried in from the previous segment: the host knows their sources)
        out->src[k] = (b0 + k >= first_bit_with_source) ? symidx[bitsym[b0 + k]] : -1;
    }
}

// identity index maps for the stage-level byte-sync entry (time stamp of bit k = k)
__device__ __forceinline__ void k_iota(unsigned *__restrict__ bitsym, long long *__restrict__ symidx, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { bitsym[i] = (unsigned)i; symidx[i] = i; }
}

// ------------------------------------------------------------------------------------------
// Frame validation (reference: standalone_matlab/Functionized/checkParity.m:1-92,
// daytimeDecode.m:1-40; MATLAB's minorFrames(frame, w) is bytes[w-1]).  One thread per frame;
// the summary counters and the two histograms MATLAB's mode() needs are accumulated with atomics.
// ------------------------------------------------------------------------------------------
struct TipFrame {            // == pdt_tip_frame
    unsigned short minor_id;
    unsigned char spacecraft, parity, checked, has_time;
    unsigned short day;
    int day_ms;
};
struct TipCounters {
    unsigned long long frames_checked, good_frames, bad_chunks, time_frames;
    unsigned hist_sc[256];
    unsigned hist_day[512];
};

__device__ __forceinline__ void k_tip_check(const FrameRec *__restrict__ frames, unsigned nframes,
                                                    TipFrame *__restrict__ out, TipCounters *__restrict__ cnt)
{
    const unsigned f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nframes) return;
    const FrameRec &fr = frames[f];
    TipFrame o;
    o.minor_id = 0; o.spacecraft = 0; o.parity = 0; o.checked = 0; o.has_time = 0; o.day = 0; o.day_ms = 0;
    if (fr.complete && fr.nbytes == 104) {
        const unsigned char *b = fr.bytes;
        o.checked = 1;
        // five groups of 17 bytes starting at byte 2; parity bits 5..1 of byte 103 (checkParity.m:20-86)
        unsigned bad = 0;
#pragma unroll
        for (int g = 0; g < 5; g++) {
            unsigned ones = 0;
            for (int w = 0; w < 17; w++) ones += (unsigned)__popc((unsigned)b[2 + 17 * g + w]);
            const unsigned bit = ((unsigned)b[103] >> (5 - g)) & 1u;
            bad |= ((ones & 1u) != bit) ? (1u << g) : 0u;
        }
        o.parity = (unsigned char)bad;
        o.minor_id = (unsigned short)(((b[4] & 1u) << 8) | b[5]);                  // daytimeDecode.m:4
        o.spacecraft = b[2];                                                       // :16
        atomicAdd(&cnt->frames_checked, 1ull);
        if (bad == 0) atomicAdd(&cnt->good_frames, 1ull);
        atomicAdd(&cnt->bad_chunks, (unsigned long long)__popc(bad));
        atomicAdd(&cnt->hist_sc[o.spacecraft], 1u);
        if (o.minor_id == 0) {                                                     // :18-31
            o.has_time = 1;
            o.day = (unsigned short)(((unsigned)b[8] << 1) + (((unsigned)b[9] | 128u) >> 7));
            const int ms = (int)(((unsigned)(b[9] & 7u) << 24) + ((unsigned)b[10] << 16) + ((unsigned)b[11] << 8) + (unsigned)b[12]);
            o.day_ms = (ms < 86400000) ? ms : -1;
            atomicAdd(&cnt->time_frames, 1ull);
            atomicAdd(&cnt->hist_day[o.day], 1u);
        }
    }
    out[f] = o;
}

// ------------------------------------------------------------------------------------------
// What the reference's chunk loop knows after every chunk (pdt_keep_quality; POESTIPdemod/main.c:413-481,
// ARGOSdemod/main.c:265-294): CarrierTrackPLL's return value = averagePhase after the chunk's last sample
// (CarrierTrackingPLL.c:277), the symbols and bits decided so far (the sums of GardenerClockRecovery's and ManchesterDecode's
// return values) and where element 0 of the time array the progress line prints comes from.  One thread per chunk: the symbol
// pick indices and the bits' symbol indices are ascending, so the counts are two binary searches.
// ------------------------------------------------------------------------------------------
struct ChunkInfo {
    double avg_phase;
    unsigned long long sym_upto, bits_upto;    // symbols / bits decided in chunks 0 .. c
    long long t0_src;                          // ARGOS: global sample index behind waveDataTime[0] after the in-place compactions
};

template <typename T>
__device__ __forceinline__ void k_chunk_info(const T *__restrict__ avg_stream, long long n, long long chunk, long long n_chunks,
                                             int interp, const long long *__restrict__ symidx,
                                             const unsigned long long *__restrict__ nsym_p, const unsigned *__restrict__ bitsym,
                                             const unsigned long long *__restrict__ nbits_p, ChunkInfo *__restrict__ out)
{
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const long long nsym = (long long)*nsym_p, nbits = (long long)*nbits_p;
    auto syms_before = [&](long long g) {              // symbols picked at an interpolated-sample index below g
        long long lo = 0, hi = nsym;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (symidx[mid] < g) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    auto bits_before = [&](long long sidx) {           // bits whose time stamp comes from a symbol below sidx
        long long lo = 0, hi = nbits;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if ((long long)bitsym[mid] < sidx) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const long long begin = c * chunk;
    const long long end = ((c + 1) * chunk < n) ? (c + 1) * chunk : n;
    const long long s0 = syms_before(begin * interp), s1 = syms_before(end * interp);
    const long long b0 = bits_before(s0), b1 = bits_before(s1);
    ChunkInfo o;
    o.avg_phase = avg_stream ? (double)avg_stream[end - 1] : 0.0;
    o.sym_upto = (unsigned long long)s1;
    o.bits_upto = (unsigned long long)b1;
    // Gardner compacts time[k] = time[pick k], Manchester time[j] = time[symbol of bit j] (GardenerClockRecovery.c:31,
    // ManchesterDecode.c:86): element 0 ends up as the first bit's symbol's pick, else the first symbol's, else stays put
    o.t0_src = (b1 > b0) ? symidx[bitsym[b0]] : (s1 > s0) ? symidx[s0] : begin * interp;
    out[c] = o;
}

}  // namespace pdt
