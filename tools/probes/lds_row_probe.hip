// ds_read_b128 / ds_write_b128 pace when every lane walks its own LDS row (experiment): which row strides are conflict-free?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) k(int stride16, int mode, int iters, float *out)
{
    extern __shared__ __align__(16) unsigned char lds[];
    const int lane = threadIdx.x;
    int rowoff;
    if (mode == 0) rowoff = lane * stride16 * 16;                                   // plain stride
    else if (mode == 1) rowoff = lane * 1024 + 16 * ((lane + (lane >> 3)) & 7);     // rotate by the octet index
    else if (mode == 2) rowoff = lane * 1024 + 16 * ((lane ^ (lane >> 3)) & 7);     // xor with the octet index
    else rowoff = lane * 1024 + 16 * (((lane & 3) * 2 + ((lane >> 2) & 1) + (lane >> 5)) & 7);
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < 70000 / 4; i += 64) ((float *)lds)[i] = (float)i;
    __syncthreads();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            const f4 v = *(const f4 *)(lds + rowoff + 16 * ((kk + it) & 63));
            acc += v;
        }
    }
    out[blockIdx.x * 64 + lane] = acc.x + acc.y + acc.z + acc.w;
}
int main()
{
    float *o; hipMalloc(&o, 1 << 20);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    struct { int s16, mode; const char *what; } cases[] = {{1, 0, "stride 16 B (lanes adjacent)"}, {64, 0, "stride 1024"}, {65, 0, "stride 1040"},
        {66, 0, "stride 1056"}, {68, 0, "stride 1088"}, {67, 0, "stride 1072"}, {69, 0, "stride 1104"}, {0, 1, "1024 + 16 ((l + l/8) & 7)"},
        {0, 2, "1024 + 16 ((l ^ l/8) & 7)"}, {0, 3, "1024 + 16 (perm)"}};
    for (auto &c : cases) {
        float best = 1e9f;
        for (int r = 0; r < 3; r++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 70 * 1024, 0, c.s16, c.mode, iters, o);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("%-32s %.2f ns per ds_read_b128\n", c.what, best * 1e6 / (iters * 16.0));
    }
    return 0;
}
