// clock_probe2.hip -- single-wave VALU issue/latency model: dependent chain vs independent chains
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int ILP, int UNROLL>
__global__ void chain(float *out, long long n, long long *cyc)
{
    float x[ILP];
    for (int j = 0; j < ILP; j++) x[j] = out[j];
    long long c0 = clock64();
    for (long long i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
            for (int j = 0; j < ILP; j++) x[j] = __builtin_fmaf(x[j], 1.0000001f, 1e-7f);
    }
    long long c1 = clock64();
    float s = 0;
    for (int j = 0; j < ILP; j++) s += x[j];
    out[0] = s;
    *cyc = c1 - c0;
}
template <int ILP, int UNROLL> void run(float *d, long long *dc, int threads, const char *what)
{
    const long long n = 400000;
    hipLaunchKernelGGL((chain<ILP, UNROLL>), dim3(1), dim3(threads), 0, 0, d, n, dc);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("%-28s ILP=%d UNROLL=%2d threads=%3d : %.2f cycles per fma instruction, %.2f cycles per dependent step\n", what, ILP, UNROLL,
           threads, (double)c / ((double)n * ILP * UNROLL), (double)c / ((double)n * UNROLL));
}
__global__ void mixed(float *out, long long n, long long *cyc)
{   // dependent chain of different op kinds: mul, add, cmp+cndmask, rndne, cvt
    float x = out[0], g = out[1];
    long long c0 = clock64();
    for (long long i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float r = __builtin_rintf(x);                // 1
            float d = x - r;                             // 2
            float e = d * g;                             // 3
            e = (e > 0.1f) ? 0.1f : e;                   // 4,5 (cmp + cndmask)
            x = x - e;                                   // 6
            x = x + 9.01f;                               // 7
            x = (x > 30000.f) ? x - 30000.f : x;         // 8,9,10
        }
    }
    long long c1 = clock64();
    out[0] = x;
    *cyc = c1 - c0;
}
int main()
{
    float *d; long long *dc;
    hipMalloc(&d, 256); hipMalloc(&dc, 8); hipMemset(d, 0, 256);
    run<1, 4>(d, dc, 64, "dependent");
    run<1, 16>(d, dc, 64, "dependent");
    run<1, 64>(d, dc, 64, "dependent");
    run<2, 16>(d, dc, 64, "2 independent chains");
    run<4, 16>(d, dc, 64, "4 independent chains");
    run<8, 8>(d, dc, 64, "8 independent chains");
    run<1, 64>(d, dc, 1, "dependent, 1 thread");
    run<1, 64>(d, dc, 256, "dependent, 4 waves");
    const long long n = 200000;
    hipLaunchKernelGGL(mixed, dim3(1), dim3(64), 0, 0, d, n, dc);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("mixed 10-op dependent body: %.1f cycles per body (%.2f per op)\n", (double)c / (n * 4.0), (double)c / (n * 40.0));
    return 0;
}
