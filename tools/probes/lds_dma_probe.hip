#include <hip/hip_runtime.h>
#include <cstdio>
extern "C" __global__ void __launch_bounds__(64) k(const float *in, float *out)
{
    __shared__ __attribute__((aligned(16))) float lds[2][256];
    const float *p = in + threadIdx.x * 100;       // every lane its own stream
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(p), "s"((unsigned)(size_t)&lds[1][0]) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int w = 0; w < 4; w++) out[threadIdx.x * 4 + w] = lds[1][threadIdx.x * 4 + w];
    for (int w = 0; w < 256; w++) if (threadIdx.x == 0) out[256 + w] = lds[1][w];
}
int main()
{
    float *in, *out; hipMalloc(&in, 64 * 100 * 4 + 64); hipMalloc(&out, 2048 * 4);
    float h[6400 + 16]; for (int i = 0; i < 6416; i++) h[i] = (float)i;
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, out);
    float o[512]; hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost);
    printf("lane 0: %g %g %g %g  lane 1: %g %g %g %g  lane 63: %g %g %g %g\n", o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[252], o[253], o[254], o[255]);
    printf("raw lds[0..11]: "); for (int i = 0; i < 12; i++) printf("%g ", o[256 + i]); printf("\n");
    return 0;
}
