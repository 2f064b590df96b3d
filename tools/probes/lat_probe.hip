// Round 4 probe: what a lone wavefront on gfx950 pays per DEPENDENT instruction, by kind -- and whether a wave64 vector
// operation with 16 (or 32) lanes enabled issues faster than one with all 64.  (experiment; s_memtime around unrolled chains)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP64(x) x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x

__device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }

template <int MODE>
__global__ void __launch_bounds__(64) k(float a, float *out, uint64_t *clk, int lanes)
{
    float g = 1.0f + threadIdx.x * 1e-6f, h = 2.0f;
    __shared__ float lds[4096];
    lds[threadIdx.x] = (float)((threadIdx.x * 4) & 255);
    __syncthreads();
    uint64_t best = ~0ull;
    for (int r = 0; r < 5; r++) {
        uint64_t t0, t1;
        if ((int)threadIdx.x < lanes) {
            t0 = now();
            for (int it = 0; it < 16; it++) {
                if (MODE == 0) { REP64(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(g) : "v"(a));) }
                if (MODE == 1) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(g) : "v"(a));) }
                if (MODE == 2) { REP64(asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2" : "+v"(g), "+v"(h) : "v"(a));) }   // 2 independent
                if (MODE == 3) { REP64(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(g) : "v"(a) : "vcc");) }
                if (MODE == 4) { REP64(asm volatile("v_mul_f32 %0, %0, %1\n s_nop 0" : "+v"(g) : "v"(a));) }
                if (MODE == 5) { REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*(double *)&g) : "v"((double)a));) }
                if (MODE == 6) { REP64(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(*(int *)&h));) }
                if (MODE == 7) { REP64(asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(g));) }
                if (MODE == 8) { REP64(asm volatile("v_readfirstlane_b32 s20, %0\n v_mov_b32 %0, s20" : "+v"(g) : : "s20");) }
                if (MODE == 9) { REP64(asm volatile("v_rcp_f32 %0, %0" : "+v"(g));) }
                if (MODE == 10) { REP64(asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1" : "+v"(g) : "v"(a));) }
                if (MODE == 11) { REP64(asm volatile("s_mul_i32 s20, s20, 3" : : : "s20");) }
                if (MODE == 12) { REP64(asm volatile("v_trunc_f32 %0, %0" : "+v"(g));) }
                if (MODE == 13) { REP64(asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[0,1]" : "+v"(*(double *)&g) : "v"((double)a));) }
                if (MODE == 14) { REP64(asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(g) : "v"(a));) }
                if (MODE == 15) { REP64(asm volatile("v_bfi_b32 %0, %1, 1.0, %0" : "+v"(g) : "v"(a));) }
                if (MODE == 16) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1" : "+v"(g) : "v"(a));) }
                if (MODE == 17) { REP64(asm volatile("v_mul_f32 %0, 0x3e22f983, %0\n v_trunc_f32 %0, %0\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %0" : "+v"(g) : "v"(a));) }
                if (MODE == 18) { REP64(asm volatile("v_cmp_ge_f32_e64 vcc, |%0|, %1\n v_bfi_b32 %2, %1, 1.0, %0\n v_fma_f32 %2, %2, %1, %0\n v_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(g), "+v"(h) : "v"(a) : "vcc");) }
                if (MODE == 19) { REP64(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(g) : "v"(a));) }
            }
            t1 = now();
            if (t1 - t0 < best) best = t1 - t0;
        }
    }
    if (threadIdx.x == 0) clk[blockIdx.x] = best;
    out[blockIdx.x * 64 + threadIdx.x] = g + h;
}

template <int MODE> void run(const char *what, int per, int lanes, int waves = 1)
{
    float *o; uint64_t *c; hipMalloc(&o, 1 << 20); hipMalloc(&c, 8 * 4096);
    hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, 0.999f, o, c, lanes);
    hipDeviceSynchronize();
    uint64_t hc[4096]; hipMemcpy(hc, c, 8 * waves, hipMemcpyDeviceToHost);
    uint64_t mn = ~0ull, mx = 0; for (int i = 0; i < waves; i++) { if (hc[i] < mn) mn = hc[i]; if (hc[i] > mx) mx = hc[i]; }
    printf("%-44s lanes %2d waves %4d: %.2f .. %.2f counter ticks per instruction\n", what, lanes, waves, (double)mn / (16.0 * 64 * per), (double)mx / (16.0 * 64 * per));
    hipFree(o); hipFree(c);
}

// ticks of the counter per ns: time a long spin with events
__global__ void spin(uint64_t *c) { uint64_t t0 = now(); float g = 1.f; for (int i = 0; i < 2000000; i++) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(g)); c[0] = now() - t0; c[1] = (uint64_t)g; }

int main()
{
    uint64_t *c; hipMalloc(&c, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, c); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, c); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    uint64_t hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
    printf("counter: %.4f ticks per ns; spin of 2M dependent v_mul: %.3f ms = %.2f ns each\n", hc[0] / (ms * 1e6), ms, ms * 1e6 / 2e6);
    for (int lanes : {64, 32, 16, 1}) {
        run<0>("dependent v_mul_f32", 1, lanes);
        run<1>("dependent v_fma_f32", 1, lanes);
        run<2>("2 independent v_mul_f32 (per pair /2)", 2, lanes);
        run<3>("v_cmp + v_cndmask (per pair /2)", 2, lanes);
        run<10>("mul add mul sub dependent", 4, lanes);
    }
    run<4>("v_mul + s_nop 0 (per pair /2)", 2, 64);
    run<5>("dependent v_mul_f64", 1, 64);
    run<5>("dependent v_mul_f64", 1, 16);
    run<6>("dependent ds_read_b32 + wait", 1, 64);
    run<6>("dependent ds_read_b32 + wait", 1, 1);
    run<7>("dependent v_mov_dpp row_shr", 1, 64);
    run<8>("readfirstlane + v_mov (per pair /2)", 2, 64);
    run<9>("dependent v_rcp_f32", 1, 64);
    run<11>("dependent s_mul_i32", 1, 64);
    run<12>("dependent v_trunc_f32 (+nop: -4.1)", 1, 64);
    run<13>("dependent v_pk_mul_f32 (+nop)", 1, 64);
    run<14>("dependent v_med3_f32 (+nop)", 1, 64);
    run<15>("dependent v_bfi_b32 (+nop)", 1, 64);
    run<16>("4 dependent v_fma_f32 per block (5 slots with nop)", 4, 64);
    run<17>("mul-literal, trunc, fma, fma (phase wrap)", 4, 64);
    run<18>("cmp, bfi, fma, cndmask", 4, 64);
    run<19>("4 dependent v_add_f32", 4, 64);
    // one wavefront per SIMD on every CU, and two / four per SIMD
    for (int waves : {1024, 2048, 4096}) { run<0>("dependent v_mul_f32", 1, 64, waves); run<10>("mul add mul sub dependent", 4, 64, waves); }
    return 0;
}
