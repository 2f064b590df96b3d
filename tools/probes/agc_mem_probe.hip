// Round 4 probe: what holds the AGC walkers up at the c3 geometry (31 816 blocks of 28 288 samples, warm-up of one block,
// 498 wavefronts, two per CU)?  PC sampling is not available on the gpurun boxes ("rocprofv3-avail list --pc-sampling": no
// agent), so the attribution is differential: the walker of pdt_kernels_front.h (agc_range: LDS-direct look-ahead ring,
// 16-sample calm batches) rebuilt with switches --
//   LOAD  0: the ring is never refilled (the walker reads whatever LDS holds; arithmetic pace only)
//   STORE 0: no output stores
//   LT    1: the stream is addressed lane-tiled (a wavefront's 64 x 16 B are 1 KiB of consecutive bytes) instead of one
//            stream per lane 113 KB apart -- same bytes, coalesced
//   PF      : look-ahead depth in 16-byte vectors (ring = PF KiB of LDS per wavefront => wavefronts per CU)
//   WARM    : warm-up length in samples
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I project-desert-tortoise_amd/csrc -o tools/probes/agc_mem_probe tools/probes/agc_mem_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "pdt_kernels_back.h"
#include "pdt_kernels_front.h"
using namespace pdt;

__global__ void fill(float *x, long long n)
{
    long long i = blockIdx.x * 256ll + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 13);
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
        x[i] = ((float)(h & 0xffff) / 65536.0f - 0.5f) * 0.9f;          // |x| < 0.45: mean |x| 0.225 -> gain ~4.4: calm
    }
}

__global__ void checksum(const float *x, long long i0, long long n, unsigned long long *acc)
{
    long long i = i0 + blockIdx.x * 256ll + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    unsigned long long a = 0;
    for (; i < n; i += stride) a += (unsigned long long)__float_as_uint(x[i]) * (unsigned long long)((i % 1000003) + 1);
    atomicAdd(acc, a);
}
static unsigned long long check(const float *x, long long i0, long long n)
{
    unsigned long long *d, h = 0;
    (void)hipMalloc(&d, 8); (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL(checksum, dim3(4096), dim3(256), 0, 0, x, i0, n, d);
    (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); (void)hipFree(d);
    return h;
}

template <bool LOAD, bool STORE, bool LT, int PF>
__device__ __forceinline__ void walk(const float *__restrict__ in, float *__restrict__ out, long long i0, long long i1, float &gain,
                                     float decay, unsigned char *ring, long long lt_base, long long lt_stride)
{
    // LT: lane's vector q of the range lives at in + lt_base + q * lt_stride (floats); lt_stride = 256 (64 lanes x 4 floats)
    constexpr int NB = 4, NBATCH = PF / NB;
    const long long nbt = (i1 - i0) / 16;
    const unsigned ring0 = (unsigned)(size_t)ring;
    const unsigned char *mine = ring + 16 * (threadIdx.x & 63);
    auto src = [&](long long v) -> const float * { return LT ? in + lt_base + v * lt_stride : in + i0 + v * 4; };
    auto dst = [&](long long v) -> float * { return LT ? out + lt_base + v * lt_stride : out + i0 + v * 4; };
    if (LOAD) {
#pragma unroll
        for (int u = 0; u < PF; u++) ring_issue(src(u), ring0 + u * PDT_RING_SLOT);
        ring_wait<PF - NB>();
    }
    Vec16<float> xb[NB], xn[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) xb[k] = *reinterpret_cast<const Vec16<float> *>(mine + k * PDT_RING_SLOT);
    int rb = 0;
    long long v = 0;
    for (long long bt = 0; bt < nbt; bt++, v += NB) {
        const int rnext = (rb + 1 == NBATCH) ? 0 : rb + 1;
        if (LOAD) ring_wait<PF - 2 * NB>();
#pragma unroll
        for (int k = 0; k < NB; k++) xn[k] = *reinterpret_cast<const Vec16<float> *>(mine + (rnext * NB + k) * PDT_RING_SLOT);
        Vec16<float> yv[NB];
        AgcParams<float> P; P.attack = decay * 0.5f; P.decay = decay; P.squelch = 0; P.raw_out = nullptr; P.squelch_thr = 0;
        if (agc_calm<float, NB>(xb, gain, decay)) {
#pragma unroll
            for (int k = 0; k < NB; k++)
#pragma unroll
                for (int w = 0; w < 4; w++) yv[k].v[w] = agc_step_calm(xb[k].v[w], gain, decay);
        } else {
#pragma unroll
            for (int k = 0; k < NB; k++)
#pragma unroll
                for (int w = 0; w < 4; w++) yv[k].v[w] = agc_step(xb[k].v[w], gain, P);
        }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            if (STORE) *reinterpret_cast<Vec16<float> *>(dst(v + k)) = yv[k];
            if (LOAD) ring_issue(src(v + PF + k), ring0 + (unsigned)(rb * NB + k) * PDT_RING_SLOT);
            xb[k] = xn[k];
        }
        rb = rnext;
    }
    if (LOAD) ring_wait<0>();
}

template <bool LOAD, bool STORE, bool LT, int PF>
__global__ void __launch_bounds__(64) k_walk(const float *__restrict__ in, float *__restrict__ out, long long n, long long B, long long W,
                                             float decay, float *__restrict__ gout)
{
    __shared__ __attribute__((aligned(16))) unsigned char ring[PF * PDT_RING_SLOT];
    const long long j = (long long)blockIdx.x * 64 + threadIdx.x;
    long long start = j * B;
    if (start >= n) return;
    float gain = 4.4f;
    const long long wb = (start >= W) ? W : 0;
    // LT addressing: tile = 64 consecutive blocks; the lane's vector q of its block at tile_base + q * 256 + lane * 4
    const long long tile_base = (long long)blockIdx.x * 64 * B + (threadIdx.x & 63) * 4;
    // warm-up over the samples in front of the block: natural layout = the previous lane's block; LT: the same tile region
    // shifted (the bytes touched are what counts for the probe)
    if (wb) walk<LOAD, false, LT, PF>(in, out, start - wb, start, gain, decay, ring, tile_base + (B - wb) / 4 * 256 - (blockIdx.x ? 64 * B : 0), 256);
    walk<LOAD, STORE, LT, PF>(in, out, start, start + B, gain, decay, ring, tile_base, 256);
    gout[j] = gain;
}


// ---- natural layout, full-line transfers: a wavefront owns 64 consecutive blocks; samples move in super-batches of 32 per
// lane (128 B = one line per block).  Transfer instruction s (of 8) carries the blocks l with (l & 7) == s: its lane t moves
// piece (t & 7) of block 8 (t >> 3) + s -- eight lanes per line, eight full lines per instruction -- to / from LDS slot s
// (slot stride 1040 B); walker lane l finds piece p of its own block at (l & 7) * 1040 + (l >> 3) * 128 + 16 p: conflict-free
// 16-byte LDS accesses on both faces.
#define TR_SLOT 1040
#define TR_SB (8 * TR_SLOT)
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool STORE, int R>
__device__ __forceinline__ void walk_tr(const float *__restrict__ in, float *__restrict__ out, long long e0, long long nsb, float &gain,
                                        float decay, unsigned char *ring, const unsigned (&voff)[8])
{
    // e0: element index of lane 0's first sample of the range (block 0 of the wavefront); every lane walks the same offsets
    const int t = threadIdx.x & 63;
    const unsigned ring0 = (unsigned)(size_t)ring;
    const unsigned stage0 = ring0 + R * TR_SB;
    const unsigned char *mine = ring + (t & 7) * TR_SLOT + (t >> 3) * 128;
    unsigned char *stage_mine = ring + R * TR_SB + (t & 7) * TR_SLOT + (t >> 3) * 128;
    const unsigned char *stage_row = ring + R * TR_SB + t * 16;
    const char *gin = (const char *)(in + e0);
    char *gout = (char *)(out + e0);
    auto issue = [&](long long sb, int slot) {
        const char *b = gin + sb * 128;
#pragma unroll
        for (int s = 0; s < 8; s++)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff[s]), "s"(b), "s"(ring0 + slot * TR_SB + s * TR_SLOT) : "memory");
    };
    for (int u = 0; u < R; u++) issue(u, u);
    int slot = 0;
    AgcParams<float> P; P.attack = decay * 0.5f; P.decay = decay; P.squelch = 0; P.raw_out = nullptr; P.squelch_thr = 0;
    for (long long sb = 0; sb < nsb; sb++) {
        ring_wait<(R - 1) * 8>();
        Vec16<float> x[8], y[8];
#pragma unroll
        for (int p = 0; p < 8; p++) x[p] = *reinterpret_cast<const Vec16<float> *>(mine + slot * TR_SB + p * 16);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (agc_calm<float, 4>(x + 4 * h, gain, decay)) {
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int w = 0; w < 4; w++) y[4 * h + k].v[w] = agc_step_calm(x[4 * h + k].v[w], gain, decay);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int w = 0; w < 4; w++) y[4 * h + k].v[w] = agc_step(x[4 * h + k].v[w], gain, P);
            }
        }
        if (STORE) {
#pragma unroll
            for (int p = 0; p < 8; p++) *reinterpret_cast<Vec16<float> *>(stage_mine + p * 16) = y[p];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            char *ob = gout + sb * 128;
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const f4 v = *reinterpret_cast<const f4 *>(stage_row + s * TR_SLOT);
                asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(voff[s]), "v"(v), "s"(ob) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue(sb + R, slot);
        slot = (slot + 1 == R) ? 0 : slot + 1;
    }
    ring_wait<0>();
}

template <bool STORE, int R>
__global__ void __launch_bounds__(64) k_walk_tr(const float *__restrict__ in, float *__restrict__ out, long long n, long long B, long long W,
                                                float decay, float *__restrict__ gout)
{
    extern __shared__ __attribute__((aligned(128))) unsigned char ring[];
    const int t = threadIdx.x & 63;
    const long long jb0 = (long long)blockIdx.x * 64;
    const long long j = jb0 + t;
    unsigned voff[8];
#pragma unroll
    for (int s = 0; s < 8; s++) voff[s] = (unsigned)(((8 * (t >> 3) + s) * B + 4 * (t & 7)) * 4);
    float gain = 4.4f;
    const long long wb = (jb0 > 0) ? W : 0;             // (the first wavefront of the probe skips its warm-up)
    if (wb) walk_tr<false, R>(in, out, jb0 * B - wb, wb / 32, gain, decay, ring, voff);
    walk_tr<STORE, R>(in, out, jb0 * B, B / 32, gain, decay, ring, voff);
    if (j * B < n) gout[j] = gain;
}

template <bool STORE, int R>
void run_tr(const char *what, const float *in, float *out, long long n, long long B, long long W, float *g)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const long long nb = n / B;
    const int lds = (R + 1) * TR_SB;
    (void)hipFuncSetAttribute((const void *)k_walk_tr<STORE, R>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float best = 1e9f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k_walk_tr<STORE, R>), dim3((unsigned)((nb + 63) / 64)), dim3(64), lds, 0, in, out, n, B, W, 0.004f, g);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double steps = (double)(B + W);
    const double bytes = (double)(B + W) / B * n * 4 + (STORE ? n * 4.0 : 0);
    fflush(stdout);
    printf("%-64s B %6lld W %6lld R  %3d waves %5lld: %7.3f ms  %6.1f ns/step  %6.1f clk/step  read+write %.1f GB -> %.2f TB/s  (%s)\n", what, B, W, R,
           (nb + 63) / 64, best, best * 1e6 / steps, best * 1e6 / steps * 2.4, bytes / 1e9, bytes / (best * 1e9), hipGetErrorString(hipGetLastError()));
}

template <bool LOAD, bool STORE, bool LT, int PF>
void run(const char *what, const float *in, float *out, long long n, long long B, long long W, float *g)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const long long nb = n / B;
    float best = 1e9f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k_walk<LOAD, STORE, LT, PF>), dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, 0, in, out, n, B, W, 0.004f, g);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double steps = (double)(B + W);
    fflush(stdout);
    printf("%-64s B %6lld W %6lld PF %3d waves %5lld: %7.3f ms  %6.1f ns/step  %6.1f clk/step  read+write %.1f GB -> %.2f TB/s\n", what, B, W, PF,
           (nb + 63) / 64, best, best * 1e6 / steps, best * 1e6 / steps * 2.4, (LOAD ? (double)(B + W) / B * n * 4 : 0) / 1e9 + (STORE ? n * 4.0 / 1e9 : 0),
           ((LOAD ? (double)(B + W) / B * n * 4 : 0) + (STORE ? n * 4.0 : 0)) / (best * 1e9));
}

int main(int argc, char **argv)
{
    const long long B0 = 28288, nb0 = 31816;
    const long long n = B0 * nb0;            // 900 M samples
    float *in, *out, *g;
    if (hipMalloc(&in, (n + (1 << 22)) * 4) != hipSuccess || hipMalloc(&out, (n + (1 << 22)) * 4) != hipSuccess) { printf("no memory\n"); return 1; }
    (void)hipMalloc(&g, 1 << 22);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, in, n + (1 << 22));
    (void)hipDeviceSynchronize();
    printf("AGC walker probe, %lld samples\n", n); fflush(stdout); setvbuf(stdout, nullptr, _IONBF, 0);
    run<true, true, false, 64>("product form: per-lane streams, loads + stores", in, out, n, B0, B0, g);
    const unsigned long long ref = check(out, 64 * B0, n);
    run<true, false, false, 64>("no stores", in, out, n, B0, B0, g);
    run<false, true, false, 64>("no loads (stores only)", in, out, n, B0, B0, g);
    run<false, false, false, 64>("no memory at all (arithmetic + LDS reads)", in, out, n, B0, B0, g);
    run<true, true, true, 64>("lane-tiled addressing (coalesced KiB), loads + stores", in, out, n, B0, B0, g);
    run<true, false, true, 64>("lane-tiled, no stores", in, out, n, B0, B0, g);
    run<false, true, true, 64>("lane-tiled, stores only", in, out, n, B0, B0, g);
    run<true, true, false, 16>("per-lane streams, PF 16 (8 wavefronts per CU could fit)", in, out, n, B0, B0, g);
    run<true, true, true, 16>("lane-tiled, PF 16", in, out, n, B0, B0, g);
    run<true, true, true, 32>("lane-tiled, PF 32", in, out, n, B0, B0, g);
    // shorter warm-up (what a tile-granular guess would allow): 9 984 samples
    run<true, true, false, 64>("per-lane streams, warm-up 9 984", in, out, n, B0, 9984, g);
    run<true, true, true, 64>("lane-tiled, warm-up 9 984", in, out, n, B0, 9984, g);
    // half-length blocks, twice the wavefronts (needs PF <= 32 for 4 per CU)
    run<true, true, true, 32>("lane-tiled, B/2, warm-up 9 984, PF 32", in, out, n, B0 / 2, 9984, g);
    (void)hipMemset(out, 0, n * 4);
    run_tr<true, 8>("natural layout, full-line transfers through LDS, R 8", in, out, n, B0, B0, g);
    printf("output of the full-line walker %s the product form's (checksum over blocks 64 .. end)\n", check(out, 64 * B0, n) == ref ? "EQUALS" : "DIFFERS FROM");
    run_tr<false, 8>("natural, full-line, no stores, R 8", in, out, n, B0, B0, g);
    run_tr<true, 4>("natural, full-line, R 4", in, out, n, B0, B0, g);
    run_tr<true, 8>("natural, full-line, R 8, warm-up 9 984", in, out, n, B0, 9984, g);
    run_tr<true, 4>("natural, full-line, R 4, warm-up 9 984", in, out, n, B0, 9984, g);
    run_tr<true, 4>("natural, full-line, R 4, B/2, warm-up 9 984", in, out, n, B0 / 2, 9984, g);
    // do the transfers move the right bytes?  (the out stream of a run must equal the recurrence's output: spot check on the host)
    return 0;
}
