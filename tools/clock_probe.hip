// clock_probe.hip -- what shader clock does a single-wavefront serial kernel actually get?
// Runs a dependent v_fma chain on one lane (a) alone and (b) while a "heater" kernel keeps
// every CU busy on another stream, and reports cycles/op and effective MHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

__global__ void chain(float *out, long long n, long long *cyc, long long *wall)
{
    float x = out[0];
    long long c0 = clock64(), w0 = wall_clock64();
    for (long long i = 0; i < n; i++) {
        x = __builtin_fmaf(x, 1.0000001f, 1e-7f);
        x = __builtin_fmaf(x, 0.9999999f, 1e-7f);
        x = __builtin_fmaf(x, 1.0000001f, 1e-7f);
        x = __builtin_fmaf(x, 0.9999999f, 1e-7f);
    }
    long long c1 = clock64(), w1 = wall_clock64();
    out[0] = x;
    *cyc = c1 - c0;
    *wall = w1 - w0;
}

__global__ void heater(float *buf, volatile int *stop, int iters)
{
    float x = threadIdx.x;
    for (int k = 0; k < iters && !*stop; k++)
        for (int i = 0; i < 4096; i++) x = __builtin_fmaf(x, 1.0000001f, 1e-7f);
    buf[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main()
{
    float *d; long long *dc, *dw; int *dstop; float *hb;
    hipMalloc(&d, 4); hipMalloc(&dc, 8); hipMalloc(&dw, 8); hipMalloc(&dstop, 4); hipMalloc(&hb, 4 * 1024 * 256 * 8);
    hipMemset(d, 0, 4); hipMemset(dstop, 0, 4);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    int wcr = 0; hipDeviceGetAttribute(&wcr, hipDeviceAttributeWallClockRate, 0);
    const long long n = 2000000;   // 8M dependent fmas
    for (int mode = 0; mode < 4; mode++) {
        int heat_blocks = (mode == 0) ? 0 : (mode == 1 ? 256 : (mode == 2 ? 2048 : 64));
        if (heat_blocks) hipLaunchKernelGGL(heater, dim3(heat_blocks), dim3(256), 0, s2, hb, dstop, 4000);
        auto t0 = std::chrono::high_resolution_clock::now();
        hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, d, n, dc, dw);
        hipStreamSynchronize(s1);
        auto t1 = std::chrono::high_resolution_clock::now();
        int one = 1; hipMemcpy(dstop, &one, 4, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        int zero = 0; hipMemcpy(dstop, &zero, 4, hipMemcpyHostToDevice);
        long long c, w; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); hipMemcpy(&w, dw, 8, hipMemcpyDeviceToHost);
        double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        double wall_s = (double)w / (wcr * 1e3);
        printf("heater_blocks=%4d host_ms=%8.3f wall_clock_ms=%8.3f clock64_cycles=%lld -> %.1f MHz, %.2f ns/fma, %.2f cycles/fma\n",
               heat_blocks, ms, wall_s * 1e3, c, c / wall_s / 1e6, wall_s * 1e9 / (4.0 * n), (double)c / (4.0 * n));
    }
    return 0;
}
