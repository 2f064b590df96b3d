// Round 4 probe: what paces the block-parallel PLL walkers at the c3 geometry (45 056 blocks of 19 968 samples, 110 000 of
// tracking warm-up, 176 workgroups of four wavefronts)?  The library's own pll_phase_range over a synthetic LT theta stream:
//   ring 0 / 1 : look-ahead in registers (the compiler's waits) / the LDS ring with hand-placed waits
//   mem  0     : the same number of steps with theta in registers (no memory at all): the arithmetic pace
// and the shader clock actually running while the kernel is at work (s_memtime against the 100 MHz constant clock).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I project-desert-tortoise_amd/csrc -o tools/probes/pll_mem_probe tools/probes/pll_mem_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstdlib>
#include <thread>
#include <vector>
#include <algorithm>
#include "../../project-desert-tortoise_amd/synth/pdt_synth.h"
#include <cstring>
#include <type_traits>
#include "pdt_kernels_back.h"
#include "pdt_kernels_front.h"
using namespace pdt;

__global__ void fill(float *x, long long n)
{
    long long i = blockIdx.x * 256ll + threadIdx.x;
    for (; i < n; i += (long long)gridDim.x * 256) {
        unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 13);
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
        x[i] = ((float)(h & 0xffff) / 65536.0f - 0.5f) * 6.2f;
    }
}

// the library's whole kernel body (guess, wide-band and acquisition-gain stages, tracking warm-up, block) on the synthetic streams
__global__ void __launch_bounds__(256) k_full(IqSrc pcm, const float *__restrict__ theta, float *__restrict__ phi, long long n, long long B,
                                              long long Wacq, long long Wtrk, PllSeam<float> *seams, unsigned *done, int skip_guess,
                                              unsigned long long *__restrict__ clk, int short_group = -1)
{
    PllParams<float> P;
    P.alpha_trk = 1.0385e-3f; P.beta_trk = 5.4e-7f; P.alpha_acq = 1.3e-2f; P.beta_acq = 8.5e-5f; P.alpha_wide = 0.1f; P.beta_wide = 5e-3f;
    P.max_freq = 0.12f; P.min_freq = -0.12f;
    const unsigned long long c0 = wall_clock64();
    if (skip_guess) pcm.p = nullptr;
    k_pll_phase<float, false>(pcm, theta, skip_guess ? 0 : n, P, B, Wacq, Wtrk, 16, phi, seams, (PllPhaseHint *)done, short_group);
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = wall_clock64() - c0;
}

// the library's way in: the body's arguments come from a pack in device memory (pdt_api.hip, k_run), the gains are run-time values
template <typename... A> struct Pack {};
template <typename H, typename... R> struct Pack<H, R...> { H h; Pack<R...> r; };
template <typename H> __device__ __forceinline__ H as_global(H v)
{
    if constexpr (std::is_pointer<H>::value) {
        using E = typename std::remove_pointer<H>::type;
        __attribute__((address_space(1))) E *g = (__attribute__((address_space(1))) E *)v;
        asm("" : "+s"(g));
        return (H)g;
    } else {
        return v;
    }
}
__device__ __forceinline__ IqSrc as_global(IqSrc v) { v.p = as_global(v.p); return v; }
typedef Pack<IqSrc, const float *, long long, PllParams<float>, long long, long long, long long, int, float *, PllSeam<float> *, unsigned *> PhasePack;
__global__ void __launch_bounds__(256) k_packed(const PhasePack *__restrict__ packs)
{
    const PhasePack &p = packs[blockIdx.z];
    k_pll_phase<float, false>(as_global(p.h), as_global(p.r.h), p.r.r.h, p.r.r.r.h, p.r.r.r.r.h, p.r.r.r.r.r.h, p.r.r.r.r.r.r.h, p.r.r.r.r.r.r.r.h,
                              as_global(p.r.r.r.r.r.r.r.r.h), as_global(p.r.r.r.r.r.r.r.r.r.h), (PllPhaseHint *)as_global(p.r.r.r.r.r.r.r.r.r.r.h));
}

// a load like the chain's streaming kernels: every CU full, HBM at full rate and a few dozen FMAs per element
__global__ void __launch_bounds__(256) burn(const float4 *__restrict__ x, float4 *__restrict__ y, long long n4, float a)
{
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 v = x[i];
#pragma unroll
        for (int k = 0; k < 24; k++) { v.x = v.x * a + v.y; v.y = v.y * a + v.z; v.z = v.z * a + v.w; v.w = v.w * a + v.x; }
        y[i] = v;
    }
}

__global__ void spin_long(uint64_t *c)
{
    asm volatile("" ::: "v255", "a255");
    float g = 1.f;
    for (int i = 0; i < 3000000; i++) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(g));     // ~11 ms
    c[0] = (uint64_t)g;
}

// a theta stream like a locked PM signal's: carrier at w rad/sample, +-0.6 rad of Manchester-like modulation, a little noise
__global__ void fill_pm(float *x, long long n, long long B, float w)
{
    long long i = blockIdx.x * 256ll + threadIdx.x;
    for (; i < n; i += (long long)gridDim.x * 256) {
        // natural index of LT element i: tile, row, lane, element
        const long long tile = i / (64 * B), r = i % (64 * B), row = r / 256, lane = (r % 256) / 4, e = r % 4;
        const long long nat = (tile * 64 + lane) * B + row * 4 + e;
        unsigned h = (unsigned)(nat * 2654435761u) ^ (unsigned)(nat >> 13);
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
        double ph = (double)w * (double)nat + (((nat / 15) & 1) ? 0.6 : -0.6) + ((double)(h & 0xffff) / 65536.0 - 0.5) * 0.2;
        ph = ph - 6.283185307179586 * floor(ph / 6.283185307179586 + 0.5);
        x[i] = (float)ph;
    }
}

template <bool RING, bool MEM, bool STORE>
__global__ void __launch_bounds__(256) k_probe(const float *__restrict__ theta, float *__restrict__ phi, long long n, long long B, long long W,
                                               float *__restrict__ gout, unsigned long long *__restrict__ clk)
{
    __shared__ __attribute__((aligned(16))) unsigned char ring_all[4 * PDT_PLL_RING_PF * PDT_RING_SLOT];
    unsigned char *ring = RING ? ring_all + (threadIdx.x >> 6) * (PDT_PLL_RING_PF * PDT_RING_SLOT) : nullptr;
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long start = j * B;
    if (start >= n) return;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    float phase = 0.1f + 1e-4f * (float)(j & 1023), freq = 0.01f;
    const float alpha = 9.4e-4f, beta = 4.4e-7f, maxf = 0.12f, minf = -0.12f;
    const long long ws = start >= W ? start - W : 0;
    if (MEM) {
        pll_phase_range<float, false, false, true>(theta, phi, B, ws, start, phase, freq, alpha, beta, maxf, minf, nullptr, nullptr, ring);
        pll_phase_range<float, STORE, false, true>(theta, phi, B, start, start + B, phase, freq, alpha, beta, maxf, minf, nullptr, nullptr, ring);
    } else {
        float t0 = 0.3f + phase, t1 = -1.2f, t2 = 2.2f, t3 = -2.9f;
        for (long long i = ws; i < start + B; i += 4) {
            pll_phase_step<float, false>(t0, phase, freq, alpha, beta, maxf, minf);
            pll_phase_step<float, false>(t1, phase, freq, alpha, beta, maxf, minf);
            pll_phase_step<float, false>(t2, phase, freq, alpha, beta, maxf, minf);
            pll_phase_step<float, false>(t3, phase, freq, alpha, beta, maxf, minf);
            asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
        }
    }
    gout[j] = phase + freq;
    if ((threadIdx.x & 63) == 0) {
        clk[2 * (j >> 6)] = __builtin_readcyclecounter() - c0;
        clk[2 * (j >> 6) + 1] = wall_clock64() - r0;
    }
}

template <bool RING, bool MEM, bool STORE>
void run(const char *what, const float *theta, float *phi, long long n, long long B, long long W, int groups, float *g, unsigned long long *clk)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k_probe<RING, MEM, STORE>), dim3(groups), dim3(256), 0, 0, theta, phi, n, B, W, g, clk);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    static unsigned long long h[8192];
    (void)hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * groups * 4, hipMemcpyDeviceToHost);
    double ghz = 0; int cnt = 0;
    for (int w = 4; w < groups * 4; w++) if (h[2 * w + 1]) { ghz += (double)h[2 * w] / ((double)h[2 * w + 1] * 10.0); cnt++; }   // 100 MHz constant clock: 10 ns a tick
    const double steps = (double)(B + W);
    const double bytes = MEM ? ((double)(B + W) / B * n * 4 + (STORE ? n * 4.0 : 0)) : 0;
    printf("%-56s B %6lld W %6lld groups %4d: %7.3f ms  %5.1f ns/step  shader clock %.2f GHz -> %5.1f clk/step   %.1f GB -> %.2f TB/s  (%s)\n", what, B, W, groups, best,
           best * 1e6 / steps, cnt ? ghz / cnt : 0.0, best * 1e6 / steps * (cnt ? ghz / cnt : 0.0), bytes / 1e9, bytes / (best * 1e9), hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

__global__ void __launch_bounds__(256) k_theta(IqSrc pcm, long long n, long long B, float *theta) { k_pll_theta<float>(pcm, n, B, theta); }

// `real`: the benched capture itself (libpdtsynth's c3 stream, an hour at 250 ksps) through the library's k_pll_theta -- the walkers
// on exactly the product's input
static int real_capture()
{
    const long long B = 19968, n = 900000000ll, slack = 64 * B + (1 << 20);
    pdt_synth_params sp; pdt_synth_default_params(&sp, 0, 250000, 1000.0, 1234); (void)pdt_synth_sine_table();
    int16_t *h = (int16_t *)malloc((size_t)n * 4);
    std::vector<std::thread> th;
    for (int t = 0; t < 8; t++) th.emplace_back([&, t] { const long long c = (n + 7) / 8, s0 = t * c; if (s0 < n) pdt_synth_fill(&sp, s0, std::min(c, n - s0), h + 2 * s0); });
    for (auto &t : th) t.join();
    void *pcm; float *theta, *phi; unsigned *done; PllSeam<float> *seams; unsigned long long *clk;
    (void)hipMalloc(&pcm, (size_t)(n + slack) * 4); (void)hipMemset(pcm, 0, (size_t)(n + slack) * 4);
    (void)hipMemcpy(pcm, h, (size_t)n * 4, hipMemcpyHostToDevice); free(h);
    (void)hipMalloc(&theta, (size_t)(n + slack) * 4); (void)hipMalloc(&phi, (size_t)(n + slack) * 4); (void)hipMalloc(&clk, 1 << 20);
    (void)hipMemset(theta, 0, (size_t)(n + slack) * 4);
    (void)hipMalloc(&done, 64); (void)hipMalloc(&seams, sizeof(PllSeam<float>) * (n / B + 4096));
    IqSrc src; src.p = pcm; src.fmt = 0;
    const long long tiles = (n / B + 1 + 63) / 64;
    hipLaunchKernelGGL(k_theta, dim3((unsigned)(tiles * (B / 64))), dim3(256), 0, 0, src, n, B, theta);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int zero = 0; zero < 3; zero++) {
        if (zero == 2) (void)hipMemset(theta, 0, (size_t)n * 4);
        const int sg = zero == 1 ? 177 : -1;
        (void)hipMemset(clk, 0, 8 * 178 * 4);
        float best = 1e9f;
        for (int r = 0; r < 3; r++) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k_full, dim3(sg >= 0 ? 178 : 177), dim3(256), 0, 0, src, theta, phi, n, B, 5000ll, 102032ll, seams, done, 0, clk, sg);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        static unsigned long long hc[1024];
        (void)hipMemcpy(hc, clk, 8 * 178 * 4, hipMemcpyDeviceToHost);
        unsigned long long mn = ~0ull, mx = 0; int slowest = 0;
        for (int i = 0; i < 178 * 4; i++) { if (hc[i] > 100 && hc[i] < mn) mn = hc[i]; if (hc[i] > mx) { mx = hc[i]; slowest = i; } }
        printf("the library's k_pll_phase body on %s, warm-up 102032: %7.3f ms; wavefronts take %.3f .. %.3f ms (the slowest: %d)  (%s)\n",
               zero == 2 ? "a theta stream of zeros" : zero == 1 ? "the same, the walkers in front of Wtrk in a wavefront of their own" : "the benched capture's theta", best, mn * 1e-5, mx * 1e-5, slowest, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}

// `lost`: where do walkers lose the carrier?  The benched weak capture (noise x6) or the strong one, the library's stages one by
// one with the loop frequency of every walker recorded behind each: guess, wide-band, acquisition gain (+ vote), tracking warm-up.
__global__ void __launch_bounds__(256) k_diag(IqSrc pcm, const float *__restrict__ theta, float *__restrict__ phi, long long n, long long B,
                                              long long Wacq, long long Wtrk, PllParams<float> P, float *__restrict__ rec /* 8 per walker */,
                                              int consensus)
{
    asm volatile("" ::: "v255", "a255");
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long start = j * B;
    if (start >= n) return;
    const long long Wwide = (Wacq / 4 + 3) & ~3ll;
    if (start < Wwide + Wacq + Wtrk) return;
    const long long ws = start - Wtrk - Wacq - Wwide;
    float phase, freq;
    pll_guess(pcm, ws, n, 16, 0.0f, phase, freq);
    if (freq > P.max_freq) freq = P.max_freq;
    if (freq < P.min_freq) freq = P.min_freq;
    float *r = rec + j * 8;
    r[0] = freq;
    __shared__ __attribute__((aligned(16))) unsigned char ring_all[4 * PDT_PLL_RING_PF * PDT_RING_SLOT];
    unsigned char *ring = ring_all + (threadIdx.x >> 6) * (PDT_PLL_RING_PF * PDT_RING_SLOT);
    pll_phase_range<float, false, false, true>(theta, phi, B, ws, ws + Wwide, phase, freq, P.alpha_wide, P.beta_wide, P.max_freq, P.min_freq);
    r[1] = freq;
    if (consensus & 1) {                     // the wavefront's median frequency replaces an outlier's
        float med;
        { int below; med = freq; for (int k = 0; k < 64; k++) { const float c = __shfl(freq, k); below = __popcll(__ballot(freq < c)); if (below >= 28 && below <= 36) med = c; } }
        if (__builtin_fabsf(freq - med) > 0.02f) freq = med;
    }
    const long long a0 = ws + Wwide, a1 = a0 + Wacq, vote0 = a1 - 128;
    pll_phase_range<float, false, false, true>(theta, phi, B, a0, vote0, phase, freq, P.alpha_acq, P.beta_acq, P.max_freq, P.min_freq);
    int far = 0, seen = 0;
    pll_phase_range<float, false, false, true, PDT_PLL_PF, true>(theta, phi, B, vote0, a1, phase, freq, P.alpha_acq, P.beta_acq, P.max_freq, P.min_freq, &far, &seen);
    r[2] = freq;
    if (2 * far > seen) {
        phase = (phase > 0) ? phase - (float)PDT_PI : phase + (float)PDT_PI;
        if (freq >= 0 && phase < 0) phase += (float)(2 * PDT_PI);
        if (freq < 0 && phase > 0) phase -= (float)(2 * PDT_PI);
    }
    r[5] = (float)far / (float)(seen ? seen : 1);
    if (consensus & 2) {
        float med;
        { int below; med = freq; for (int k = 0; k < 64; k++) { const float c = __shfl(freq, k); below = __popcll(__ballot(freq < c)); if (below >= 28 && below <= 36) med = c; } }
        if (__builtin_fabsf(freq - med) > 0.02f) freq = med;
    }
    pll_phase_range<float, false, false, true>(theta, phi, B, a1, start, phase, freq, P.alpha_trk, P.beta_trk, P.max_freq, P.min_freq, nullptr, nullptr, ring);
    r[3] = freq;
    r[4] = phase;
}

static int lost_walkers(int weak)
{
    const long long fs = 250000, n = weak ? 150000000ll : 900000000ll, B = weak ? 13312 : 19968, slack = 64 * B + (1 << 20);
    const long long Wtrk = weak ? 91900 : 102032;
    pdt_synth_params sp; pdt_synth_default_params(&sp, 0, (uint32_t)fs, 1000.0, 1234); (void)pdt_synth_sine_table();
    if (weak) sp.noise_gain *= 6;
    int16_t *h = (int16_t *)malloc((size_t)n * 4);
    std::vector<std::thread> th;
    for (int t = 0; t < 8; t++) th.emplace_back([&, t] { const long long c = (n + 7) / 8, s0 = t * c; if (s0 < n) pdt_synth_fill(&sp, s0, std::min(c, n - s0), h + 2 * s0); });
    for (auto &t : th) t.join();
    void *pcm; float *theta, *phi, *rec;
    (void)hipMalloc(&pcm, (size_t)(n + slack) * 4); (void)hipMemset(pcm, 0, (size_t)(n + slack) * 4);
    (void)hipMemcpy(pcm, h, (size_t)n * 4, hipMemcpyHostToDevice); free(h);
    (void)hipMalloc(&theta, (size_t)(n + slack) * 4); (void)hipMalloc(&phi, (size_t)(n + slack) * 4);
    (void)hipMemset(theta, 0, (size_t)(n + slack) * 4);
    const long long nb = n / B + 1;
    (void)hipMalloc(&rec, (size_t)(nb + 512) * 8 * 4);
    IqSrc src; src.p = pcm; src.fmt = 0;
    const long long tiles = (n / B + 1 + 63) / 64;
    hipLaunchKernelGGL(k_theta, dim3((unsigned)(tiles * (B / 64))), dim3(256), 0, 0, src, n, B, theta);
    PllParams<float> P; memset(&P, 0, sizeof P);
    {
        const double w = 2.0 * M_PI / (double)fs;
        const float bw_acq = (float)(127.3240 * w), bw_trk = (float)(10.3451 * w), damp = 0.999f, bw_w = bw_acq * 8.0f;
        P.alpha_acq = (4.0f * damp * bw_acq) / (1.0f + 2.0f * damp * bw_acq + bw_acq * bw_acq);
        P.beta_acq = (4.0f * bw_acq * bw_acq) / (1.0f + 2.0f * damp * bw_acq + bw_acq * bw_acq);
        P.alpha_trk = (float)((4.0 * 0.999 * (double)bw_trk) / (1.0 + 2.0 * 0.999 * (double)bw_trk + (double)(bw_trk * bw_trk)));
        P.beta_trk = (float)((4.0 * (double)bw_trk * (double)bw_trk) / (1.0 + 2.0 * 0.999 * (double)bw_trk + (double)(bw_trk * bw_trk)));
        P.alpha_wide = (4.0f * damp * bw_w) / (1.0f + 2.0f * damp * bw_w + bw_w * bw_w);
        P.beta_wide = (4.0f * bw_w * bw_w) / (1.0f + 2.0f * damp * bw_w + bw_w * bw_w);
        P.max_freq = (float)(2.0 * M_PI * 4500.0 / (double)fs); P.min_freq = -P.max_freq;
    }
    const float f_true = (float)(2.0 * M_PI * 1000.0 / (double)fs);
    std::vector<float> hr((size_t)nb * 8);
    for (int consensus = 0; consensus < 4; consensus++) {
        (void)hipMemset(rec, 0, (size_t)nb * 8 * 4);
        hipLaunchKernelGGL(k_diag, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, 0, src, theta, phi, n, B, 5000ll, Wtrk, P, rec, consensus);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(hr.data(), rec, (size_t)nb * 8 * 4, hipMemcpyDeviceToHost);
        long long cnt[4] = {0, 0, 0, 0}, walkers = 0, votes = 0;
        for (long long j = 0; j < nb; j++) {
            const float *r = &hr[(size_t)j * 8];
            if (r[0] == 0 && r[3] == 0) continue;
            walkers++;
            for (int k = 0; k < 4; k++) if (fabsf(r[k] - f_true) > 0.01f) cnt[k]++;
            if (r[5] > 0.5f) votes++;
        }
        printf("%s capture, %lld walkers (B %lld, W %lld), consensus %d (1: behind the wide stage, 2: behind the acquisition stage): frequency more than 0.01 rad/sample (400 Hz) "
               "from the carrier behind the guess %lld, the wide-band stage %lld, the acquisition-gain stage %lld, the tracking warm-up %lld; votes for the far point %lld  (%s)\n",
               weak ? "weak" : "strong", walkers, B, Wtrk, consensus, cnt[0], cnt[1], cnt[2], cnt[3], votes, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "lost")) return lost_walkers(argc > 2 && !strcmp(argv[2], "weak"));
    if (argc > 1 && !strcmp(argv[1], "real")) return real_capture();
    const long long B = 19968, W = 110000 / 4 * 4;
    const int groups = 176;
    const long long n = B * 256 * groups;            // 899.7 M samples
    float *theta, *phi, *g; unsigned long long *clk;
    const long long slack = 64 * B + (1 << 20);
    if (hipMalloc(&theta, (n + slack) * 4) != hipSuccess || hipMalloc(&phi, (n + slack) * 4) != hipSuccess) { printf("no memory\n"); return 1; }
    (void)hipMalloc(&g, 1 << 22); (void)hipMalloc(&clk, 1 << 20);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, theta, n + slack);
    (void)hipDeviceSynchronize();
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("PLL walker probe, %lld samples\n", n);
    run<false, false, false>("no memory: theta in registers", theta, phi, n, B, W, groups, g, clk);
    run<false, true, true>("look-ahead in registers (48 vectors), loads + stores", theta, phi, n, B, W, groups, g, clk);
    run<true, true, true>("LDS ring (32 vectors), loads + stores", theta, phi, n, B, W, groups, g, clk);
    run<true, true, false>("LDS ring, no stores", theta, phi, n, B, W, groups, g, clk);
    run<false, true, false>("registers, no stores", theta, phi, n, B, W, groups, g, clk);
    run<true, true, true>("LDS ring, warm-up 55 000", theta, phi, n, B, 55000, groups, g, clk);
    {
        void *pcm; unsigned *done; PllSeam<float> *seams;
        (void)hipMalloc(&pcm, (n + slack) * 4); (void)hipMemset(pcm, 0x11, (n + slack) * 4);
        (void)hipMalloc(&done, 64); (void)hipMalloc(&seams, sizeof(PllSeam<float>) * (n / B + 4096));
        IqSrc src; src.p = pcm; src.fmt = 0;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        for (int wt : {99328, 49664, 24832}) {
            float best = 1e9f;
            for (int r = 0; r < 3; r++) {
                (void)hipEventRecord(a);
                hipLaunchKernelGGL(k_full, dim3(177), dim3(256), 0, 0, src, theta, phi, n, B, 5000ll, (long long)wt, seams, done, 0, clk);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            static unsigned long long h[1024];
            (void)hipMemcpy(h, clk, 8 * 177 * 4, hipMemcpyDeviceToHost);
            unsigned long long mn = ~0ull, mx = 0; for (int i = 0; i < 176 * 4; i++) { if (h[i] < mn) mn = h[i]; if (h[i] > mx) mx = h[i]; }
            printf("the library's k_pll_phase body, tracking warm-up %6d: %7.3f ms; wavefronts take %.3f .. %.3f ms  (%s)\n", wt, best, mn * 1e-5, mx * 1e-5, hipGetErrorString(hipGetLastError()));
        }
    }
    {
        void *pcm; unsigned *done; PllSeam<float> *seams; PhasePack *dp;
        (void)hipMalloc(&pcm, (n + slack) * 4); (void)hipMemset(pcm, 0x11, (n + slack) * 4);
        (void)hipMalloc(&done, 64); (void)hipMalloc(&seams, sizeof(PllSeam<float>) * (n / B + 4096)); (void)hipMalloc(&dp, sizeof(PhasePack));
        PhasePack hp; memset(&hp, 0, sizeof hp);
        PllParams<float> P; memset(&P, 0, sizeof P);
        P.alpha_trk = 1.0385e-3f; P.beta_trk = 5.4e-7f; P.alpha_acq = 1.3e-2f; P.beta_acq = 8.5e-5f; P.alpha_wide = 0.1f; P.beta_wide = 5e-3f;
        P.max_freq = 0.12f; P.min_freq = -0.12f;
        hp.h.p = pcm; hp.h.fmt = 0; hp.r.h = theta; hp.r.r.h = n; hp.r.r.r.h = P; hp.r.r.r.r.h = B; hp.r.r.r.r.r.h = 5000; hp.r.r.r.r.r.r.h = 99328;
        hp.r.r.r.r.r.r.r.h = 16; hp.r.r.r.r.r.r.r.r.h = phi; hp.r.r.r.r.r.r.r.r.r.h = seams; hp.r.r.r.r.r.r.r.r.r.r.h = done;
        (void)hipMemcpy(dp, &hp, sizeof hp, hipMemcpyHostToDevice);
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        float best = 1e9f;
        for (int r = 0; r < 3; r++) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k_packed, dim3(177), dim3(256), 0, 0, (const PhasePack *)dp);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("the same body entered the library's way (arguments from a pack in device memory, run-time gains): %7.3f ms  (%s)\n", best, hipGetErrorString(hipGetLastError()));
        (void)hipFree(pcm); (void)hipFree(seams);
    }
    // does the PLACEMENT of the two streams matter?  (the walkers read theta and write phi at the same offsets at the same time;
    // the kernel moves 25 GB -- theta six times over, phi once -- in 5 ms: as close to the HBM roof as to the issue roof)
    {
        unsigned char *arena; void *pcm; unsigned *done; PllSeam<float> *seams;
        const size_t span = (size_t)(n + slack) * 4;
        (void)hipMalloc(&pcm, span); (void)hipMemset(pcm, 0x11, span);
        (void)hipMalloc(&done, 64); (void)hipMalloc(&seams, sizeof(PllSeam<float>) * (n / B + 4096));
        if (hipMalloc(&arena, 2 * span + (256u << 20)) == hipSuccess) {
            IqSrc src; src.p = pcm; src.fmt = 0;
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            const size_t base = ((span + (2u << 20) - 1) >> 21) << 21;          // phi = theta + a multiple of 2 MiB + delta
            printf("theta at %p (from hipMalloc: %p, phi %p)\n", (void *)arena, (void *)theta, (void *)phi);
            for (size_t delta : {(size_t)0, (size_t)256, (size_t)1024, (size_t)4096, (size_t)(64u << 10), (size_t)(1u << 20), (size_t)((1u << 20) + 4096 + 1024), (size_t)(32u << 20), (size_t)(128u << 20)}) {
                float *th = (float *)arena, *ph = (float *)(arena + base + delta);
                (void)hipMemcpy(th, theta, span, hipMemcpyDeviceToDevice);
                float best = 1e9f;
                for (int r = 0; r < 3; r++) {
                    (void)hipEventRecord(a);
                    hipLaunchKernelGGL(k_full, dim3(177), dim3(256), 0, 0, src, th, ph, n, B, 5000ll, 99328ll, seams, done, 0, clk);
                    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                    float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
                }
                printf("  phi - theta = %zu x 2 MiB + %9zu B: %7.3f ms\n", base >> 21, delta, best);
            }
            (void)hipFree(arena);
        }
        (void)hipFree(pcm); (void)hipFree(seams);
    }
    {
        hipLaunchKernelGGL(fill_pm, dim3(4096), dim3(256), 0, 0, theta, n + slack, B, 0.02f);
        (void)hipDeviceSynchronize();
        void *pcm; unsigned *done; PllSeam<float> *seams;
        (void)hipMalloc(&pcm, (n + slack) * 4); (void)hipMemset(pcm, 0x11, (n + slack) * 4);
        (void)hipMalloc(&done, 64); (void)hipMalloc(&seams, sizeof(PllSeam<float>) * (n / B + 4096));
        IqSrc src; src.p = pcm; src.fmt = 0;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        float best = 1e9f;
        for (int r = 0; r < 3; r++) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k_full, dim3(177), dim3(256), 0, 0, src, theta, phi, n, B, 5000ll, 99328ll, seams, done, 0, clk);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("the library's k_pll_phase body on a PM-like theta stream (the loop locks), warm-up 99328: %7.3f ms\n", best);
        // ... and right behind a kernel that has just written the whole theta stream (as k_pll_theta has in the chain), and
        // after a run of back-to-back launches without a pause (sustained clocks)
        best = 1e9f;
        float worst = 0;
        for (int r = 0; r < 6; r++) {
            hipLaunchKernelGGL(fill_pm, dim3(4096), dim3(256), 0, 0, theta, n + slack, B, 0.02f);
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k_full, dim3(177), dim3(256), 0, 0, src, theta, phi, n, B, 5000ll, 99328ll, seams, done, 0, clk);
            (void)hipEventRecord(b);
            hipLaunchKernelGGL(fill_pm, dim3(4096), dim3(256), 0, 0, phi, n + slack, B, 0.01f);     // (keep the chip busy behind it too)
            (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; if (ms > worst) worst = ms;
        }
        (void)hipDeviceSynchronize();
        printf("the same, launched right behind a kernel that rewrites theta, six times without a pause: %7.3f .. %7.3f ms\n", best, worst);
        // ... on a stream of the highest priority (the library's side stream), alone and beside a lone spinning wavefront on another
        // stream (as the head walker is in the chain)
        {
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            hipStream_t hi, lo;
            (void)hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, greatest);
            (void)hipStreamCreateWithFlags(&lo, hipStreamNonBlocking);
            printf("stream priorities: least %d, greatest %d\n", least, greatest);
            uint64_t *sc; (void)hipMalloc(&sc, 64);
            {
                hipStream_t low;
                (void)hipStreamCreateWithPriority(&low, hipStreamNonBlocking, least);
                for (int which = 0; which < 2; which++) {
                    best = 1e9f;
                    for (int r = 0; r < 3; r++) {
                        (void)hipDeviceSynchronize();
                        (void)hipEventRecord(a, which ? low : lo);
                        hipLaunchKernelGGL(k_full, dim3(177), dim3(256), 0, which ? low : lo, src, theta, phi, n, B, 5000ll, 99328ll, seams, done, 0, clk);
                        (void)hipEventRecord(b, which ? low : lo); (void)hipEventSynchronize(b);
                        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
                    }
                    printf("the same on a stream %s: %7.3f ms\n", which ? "of the lowest priority" : "created without a priority (non-blocking)", best);
                }
            }
            for (int beside = 0; beside < 2; beside++) {
                best = 1e9f;
                for (int r = 0; r < 3; r++) {
                    (void)hipDeviceSynchronize();
                    if (beside) hipLaunchKernelGGL(spin_long, dim3(1), dim3(64), 0, lo, sc);
                    (void)hipEventRecord(a, hi);
                    hipLaunchKernelGGL(k_full, dim3(177), dim3(256), 0, hi, src, theta, phi, n, B, 5000ll, 99328ll, seams, done, 0, clk);
                    (void)hipEventRecord(b, hi); (void)hipEventSynchronize(b);
                    float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
                }
                (void)hipDeviceSynchronize();
                printf("the same on a stream of priority %d%s: %7.3f ms\n", greatest, beside ? ", a lone wavefront spinning on another stream" : "", best);
            }
        }
        // ... and under a sustained load like the chain's: ~8 ms of full-rate streaming in front of every launch, 40 times over
        // (does the power management lower the shader clock for an issue-bound kernel that follows HBM-bound ones?)
        {
            float first = 0, lo5 = 1e9f, hi5 = 0;
            for (int r = 0; r < 40; r++) {
                for (int c = 0; c < 6; c++) (void)hipMemcpyAsync(phi, theta, (size_t)n * 4, hipMemcpyDeviceToDevice, 0);
                (void)hipEventRecord(a, 0);
                hipLaunchKernelGGL(k_full, dim3(177), dim3(256), 0, 0, src, theta, phi, n, B, 5000ll, 99328ll, seams, done, 0, clk);
                (void)hipEventRecord(b, 0);
                if (r == 0 || r >= 20) {
                    (void)hipEventSynchronize(b);
                    float ms; (void)hipEventElapsedTime(&ms, a, b);
                    if (r == 0) first = ms; else { if (ms < lo5) lo5 = ms; if (ms > hi5) hi5 = ms; }
                }
            }
            (void)hipDeviceSynchronize();
            printf("the same behind 6 device copies of theta each time, 40 rounds: first %7.3f ms, rounds 20..39 %7.3f .. %7.3f ms\n", first, lo5, hi5);
        }
        // ... and behind ~12 ms of a compute + HBM load on every CU, 40 rounds without a pause: the power management's answer
        {
            float *scratch; (void)hipMalloc(&scratch, (size_t)n * 4);
            float first = 0, lo5 = 1e9f, hi5 = 0, burn_ms = 0;
            hipEvent_t c; (void)hipEventCreate(&c);
            for (int r = 0; r < 40; r++) {
                (void)hipEventRecord(c, 0);
                for (int k = 0; k < 5; k++) hipLaunchKernelGGL(burn, dim3(256 * 8), dim3(256), 0, 0, (const float4 *)theta, (float4 *)scratch, n / 4, 0.999f);
                (void)hipEventRecord(a, 0);
                hipLaunchKernelGGL(k_full, dim3(177), dim3(256), 0, 0, src, theta, phi, n, B, 5000ll, 99328ll, seams, done, 0, clk);
                (void)hipEventRecord(b, 0);
                if (r == 0 || r >= 20) {
                    (void)hipEventSynchronize(b);
                    float ms; (void)hipEventElapsedTime(&ms, a, b);
                    if (r == 0) first = ms; else { if (ms < lo5) lo5 = ms; if (ms > hi5) hi5 = ms; }
                    (void)hipEventElapsedTime(&burn_ms, c, a);
                }
            }
            (void)hipDeviceSynchronize();
            printf("the same behind %.1f ms of FMA + HBM load on every CU each time, 40 rounds: first %7.3f ms, rounds 20..39 %7.3f .. %7.3f ms\n", burn_ms, first, lo5, hi5);
            (void)hipFree(scratch);
        }
        run<true, true, true>("LDS ring on the PM-like stream", theta, phi, n, B, W, groups, g, clk);
    }
    run<false, false, false>("no memory, 64 groups", theta, phi, n, B, W, 64, g, clk);
    run<true, true, true>("LDS ring, 64 groups (a third of the lanes)", theta, phi, n, B, W, 64, g, clk);
    run<true, true, true>("LDS ring, 256 groups", theta, phi, n + 80 * 256 * B > n + slack ? n : n, B, W, 176, g, clk);
    return 0;
}
