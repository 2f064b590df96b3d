// Lane-per-block AGC walk: where do the ~89 cycles per sample go?  (experiment; hipcc --offload-arch=gfx950 -O3 -ffp-contract=off)
//   variant 0: compiler-scheduled look-ahead ring (what agc_range does), PF vectors
//   variant 1: hand-issued loads + explicit waits, no stores
//   variant 2: no memory at all (inputs synthesised in registers): the arithmetic floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct alignas(16) V4 { float v[4]; };
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float step(float x, float &g, float att, float dec) {
    x = x * g;
    float err = __builtin_fabsf(x) - 1.0f;
    float ga = g - err * att, gd = g - err * dec;
    float n = (__builtin_fabsf(err) > g) ? ga : gd;
    n = (n < 0.f) ? 1e-4f : n;
    n = (n > 5000.f) ? 5000.f : n;
    g = n;
    return x;
}
// speculated steps: `odd` collects whether the exact step would have differed
__device__ __forceinline__ float step_open(float x, float &g, bool &odd, float att, float dec) {     // no clamps
    x = x * g;
    float err = __builtin_fabsf(x) - 1.0f;
    float ga = g - err * att, gd = g - err * dec;
    float n = (__builtin_fabsf(err) > g) ? ga : gd;
    odd = odd | (n < 0.f) | (n > 5000.f);
    g = n;
    return x;
}
__device__ __forceinline__ float step_decay(float x, float &g, bool &odd, float att, float dec) {    // decay branch, no clamps
    x = x * g;
    float err = __builtin_fabsf(x) - 1.0f;
    float n = g - err * dec;
    odd = odd | (__builtin_fabsf(err) > g) | (n < 0.f) | (n > 5000.f);
    g = n;
    return x;
}
template <int MODE, int BATCH>
__global__ void __launch_bounds__(64) spec(long long B, float att, float dec, float *gout)
{
    long long j = blockIdx.x * 64ll + threadIdx.x;
    float g = 1.0f;
    float x0 = 0.3f + 1e-6f * (float)j;
    for (long long i = 0; i + BATCH <= B; i += BATCH) {
        const float g_in = g;
        bool odd = false;
        float acc = 0;
#pragma unroll
        for (int w = 0; w < BATCH; w++) {
            const float x = (w & 1) ? -x0 * (1.0f + 0.1f * (float)(w & 3)) : x0;
            acc += (MODE == 0) ? step(x, g, att, dec) : (MODE == 1) ? step_open(x, g, odd, att, dec) : step_decay(x, g, odd, att, dec);
        }
        if (odd) {
            g = g_in;
            acc = 0;
#pragma unroll
            for (int w = 0; w < BATCH; w++) {
                const float x = (w & 1) ? -x0 * (1.0f + 0.1f * (float)(w & 3)) : x0;
                acc += step(x, g, att, dec);
            }
        }
        asm volatile("" :: "v"(acc));
    }
    gout[j] = g;
}
template <int MODE, int BATCH> float run_spec(long long B, float *g, int lanes)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((spec<MODE, BATCH>), dim3((lanes + 63) / 64), dim3(64), 0, 0, B, 0.0033f, 0.0067f, g);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}
template <int VARIANT, int PF, bool STORE>
__global__ void __launch_bounds__(64) walk(const float *__restrict__ in, float *__restrict__ out, long long B, long long n, float att, float dec, float *gout)
{
    long long j = blockIdx.x * 64ll + threadIdx.x;
    long long i = j * B, i1 = i + B;
    if (i1 > n) return;
    float g = 1.0f;
    if (VARIANT == 0) {
        V4 buf[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) buf[u] = *reinterpret_cast<const V4 *>(in + i + u * 4);
        for (; i + PF * 4 <= i1; i += PF * 4) {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                V4 y;
#pragma unroll
                for (int w = 0; w < 4; w++) y.v[w] = step(buf[u].v[w], g, att, dec);
                if (STORE) *reinterpret_cast<V4 *>(out + i + u * 4) = y;
                long long q = i + (PF + u) * 4;
                asm volatile("" : "+v"(q));
                buf[u] = *reinterpret_cast<const V4 *>(in + q);
            }
        }
    } else if (VARIANT == 1) {
        f4 buf[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const float *p = in + i + u * 4;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(buf[u]) : "v"(p) : "memory");
        }
        for (; i + PF * 4 <= i1; i += PF * 4) {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(buf[u]) : "n"(PF - 1));
                f4 x = buf[u];
                f4 y;
                y.x = step(x.x, g, att, dec); y.y = step(x.y, g, att, dec); y.z = step(x.z, g, att, dec); y.w = step(x.w, g, att, dec);
                if (STORE) {
                    float *po = out + i + u * 4;
                    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(po), "v"(y) : "memory");
                }
                const float *p = in + i + (PF + u) * 4;
                asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(buf[u]) : "v"(p) : "memory");
            }
        }
#pragma unroll
        for (int u = 0; u < PF; u++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(buf[u]));
    } else {
        float x0 = 0.3f + 1e-6f * (float)j;
        for (; i + 4 <= i1; i += 4) {
            float a = step(x0, g, att, dec), b = step(-x0, g, att, dec), c = step(x0 * 0.5f, g, att, dec), d = step(-x0 * 0.7f, g, att, dec);
            asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d));
        }
    }
    gout[j] = g;
}
template <int VARIANT, int PF, bool STORE> float run(const float *in, float *out, long long B, long long n, float *g, int lanes)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((walk<VARIANT, PF, STORE>), dim3((lanes + 63) / 64), dim3(64), 0, 0, in, out, B, n, 0.0033f, 0.0067f, g);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}
int main()
{
    const long long n = 90000000;
    float *in, *out, *g;
    hipMalloc(&in, (n + 4096) * 4); hipMalloc(&out, (n + 4096) * 4); hipMalloc(&g, 1 << 22);
    std::vector<float> h(n);
    unsigned s = 1;
    for (long long i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) % 2000 - 1000) * 3e-4f; }
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    {
        const long long B = 19968; const int lanes = 9000; const double ns = 1e6 / (double)B;
        printf("arithmetic only, ns per sample: exact %.2f/%.2f  no clamps %.2f/%.2f/%.2f  decay-speculated %.2f/%.2f/%.2f (check every 4/16[/32])\n",
               run_spec<0, 4>(B, g, lanes) * ns, run_spec<0, 16>(B, g, lanes) * ns,
               run_spec<1, 4>(B, g, lanes) * ns, run_spec<1, 16>(B, g, lanes) * ns, run_spec<1, 32>(B, g, lanes) * ns,
               run_spec<2, 4>(B, g, lanes) * ns, run_spec<2, 16>(B, g, lanes) * ns, run_spec<2, 32>(B, g, lanes) * ns);
    }
    for (long long B : {9984ll}) {
        const int lanes = (int)(n / B);
        const double ns = 1e6 / (double)B;
        printf("B %lld lanes %d (ns per sample per lane)\n", B, lanes);
        printf("  compiler ring  PF 8  no store %.2f  store %.2f\n", run<0, 8, false>(in, out, B, n, g, lanes) * ns, run<0, 8, true>(in, out, B, n, g, lanes) * ns);
        printf("  compiler ring  PF 32 no store %.2f  store %.2f\n", run<0, 32, false>(in, out, B, n, g, lanes) * ns, run<0, 32, true>(in, out, B, n, g, lanes) * ns);
        printf("  hand-issued    PF 8  no store %.2f  store %.2f\n", run<1, 8, false>(in, out, B, n, g, lanes) * ns, run<1, 8, true>(in, out, B, n, g, lanes) * ns);
        printf("  hand-issued    PF 16 no store %.2f  store %.2f\n", run<1, 16, false>(in, out, B, n, g, lanes) * ns, run<1, 16, true>(in, out, B, n, g, lanes) * ns);
        printf("  hand-issued    PF 32 no store %.2f  store %.2f\n", run<1, 32, false>(in, out, B, n, g, lanes) * ns, run<1, 32, true>(in, out, B, n, g, lanes) * ns);
        printf("  arithmetic only            %.2f\n", run<2, 8, false>(in, out, B, n, g, lanes) * ns);
    }
    return 0;
}
