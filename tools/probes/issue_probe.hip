// What paces a lone wavefront per SIMD: dependent latency or issue rate?  (experiment)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(64) k(long long n, float a, float *out)
{
    float g = 1.0f + threadIdx.x * 1e-6f, h = 2.0f, p = 3.0f, q = 4.0f;
    for (long long i = 0; i < n; i++) {
        if (MODE == 0) { g = g * a; }                                                    // 1 dependent
        if (MODE == 1) { g = g * a; g = g + a; g = g * a; g = g + a; }                    // 4 dependent
        if (MODE == 2) { g = g * a; g = g + a; g = g * a; g = g + a; g = g * a; g = g + a; g = g * a; g = g + a; }   // 8 dependent
        if (MODE == 3) { g = g * a; h = h * a; p = p * a; q = q * a; }                    // 4 independent chains, 1 op each
        if (MODE == 4) { g = g * a; h = h * a; p = p * a; q = q * a; g = g + a; h = h + a; p = p + a; q = q + a; }   // 4 chains x 2
        if (MODE == 5) { g = g * a; g = (g > 2.0f) ? 1.0f : g; }                          // mul, cmp, cndmask
        asm volatile("" : "+v"(g), "+v"(h), "+v"(p), "+v"(q));
    }
    out[blockIdx.x * 64 + threadIdx.x] = g + h + p + q;
}
template <int MODE> void run(const char *what, int waves)
{
    float *o; hipMalloc(&o, 1 << 22);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const long long n = 200000;
    float best = 1e9f;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, n, 0.999f, o);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("%-40s waves %5d: %.2f ns per iteration\n", what, waves, best * 1e6 / n);
    hipFree(o);
}
int main()
{
    for (int waves : {140, 1024, 2048, 4096}) {
        run<0>("1 dependent op", waves);
        run<1>("4 dependent ops", waves);
        run<2>("8 dependent ops", waves);
        run<3>("4 independent ops", waves);
        run<4>("8 ops in 4 chains", waves);
        run<5>("mul, cmp, select", waves);
    }
    return 0;
}
