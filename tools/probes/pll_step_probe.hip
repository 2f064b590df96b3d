// Arithmetic pace of the PLL loop-filter step on a lone wavefront per SIMD (experiment): ns per step with the inputs
// in registers, for the select-based step the kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float unwrap(float x)
{
    const float hi = __builtin_copysignf(6.2831854820251465f, x);
    const float d = __builtin_copysignf(1.7484555314695172e-07f, x);
    float r = (x - hi) + d;
    asm volatile("" : "+v"(r));
    return r;
}
__device__ __forceinline__ void step(float th, float &phase, float &freq, float alpha, float beta, float maxf, float minf)
{
    const float diff = th - phase;
    const float wrapped = unwrap(diff);
    const float err = (__builtin_fabsf(diff) >= 3.14159274f) ? wrapped : diff;
    const float f1 = freq + beta * err;
    float ph = phase + f1 + alpha * err;
    const float phw = unwrap(ph);
    ph = (__builtin_fabsf(ph) >= 6.28318548f) ? phw : ph;
    phase = ph;
    freq = __builtin_amdgcn_fmed3f(f1, minf, maxf);
}
template <int UNROLL>
__global__ void __launch_bounds__(64) k(long long n, float alpha, float beta, float *out)
{
    float ph = 0.1f + threadIdx.x * 1e-3f, fr = 0.01f;
    float t0 = 0.3f, t1 = -1.2f, t2 = 2.2f, t3 = -2.9f;
    for (long long i = 0; i < n; i += 4 * UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            step(t0, ph, fr, alpha, beta, 0.56f, -0.56f);
            step(t1, ph, fr, alpha, beta, 0.56f, -0.56f);
            step(t2, ph, fr, alpha, beta, 0.56f, -0.56f);
            step(t3, ph, fr, alpha, beta, 0.56f, -0.56f);
        }
        asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
    }
    out[blockIdx.x * 64 + threadIdx.x] = ph + fr;
}
int main()
{
    float *o; hipMalloc(&o, 1 << 22);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const long long n = 400000;
    for (int waves : {188, 1024}) {
        float best = 1e9f;
        for (int r = 0; r < 3; r++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k<8>, dim3(waves), dim3(64), 0, 0, n, 0.0026f, 3.4e-6f, o);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("waves %d: %.2f ns per PLL step\n", waves, best * 1e6 / n);
    }
    return 0;
}
