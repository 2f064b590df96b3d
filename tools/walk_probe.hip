// walk_probe.hip -- single-trajectory PLL walker microbenchmark (uses the library's device code).
// Synthetic theta: carrier 1 kHz at 50 ksps, PM +-1.06 rad with 3-sample symbols, small noise.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../project-desert-tortoise_amd/csrc/pdt_kernels_front.h"
using namespace pdt;

__device__ unsigned long long g_redo, g_vecs;

template <int MODE>
__global__ void __launch_bounds__(64) k_walk(const float *theta, float *phi, long long n, float alpha, float beta, float maxf,
                                              long long *cyc, float *state)
{
    __shared__ __attribute__((aligned(16))) float s_th[PDT_WALK_TILE + 8];
    __shared__ __attribute__((aligned(16))) float s_ph[PDT_WALK_TILE];
    float phase = 0.5f, freq = 0.125f;
    long long c0 = clock64();
    if (MODE == 0) pll_walk_wave<float, false>(theta, phi, 0, 0, n, phase, freq, alpha, beta, maxf, -maxf, s_th, s_ph);
    if (MODE == 1) {   // slow steps only, same staging
        if (threadIdx.x == 0) pll_phase_range<float, true, false, false, 32>(theta, phi, 0, n, phase, freq, alpha, beta, maxf, -maxf);
    }
    if (MODE == 2) {   // count speculation failures
        unsigned long long redo = 0, vecs = 0;
        for (long long i = 0; i + 4 <= n; i += 4) {
            Vec16<float> tv = *reinterpret_cast<const Vec16<float> *>(theta + i), pv;
            float p0 = phase, f0 = freq;
            pll_phase_vec<float, false>(tv, pv, phase, freq, alpha, beta, maxf, -maxf, true);
            float p1 = p0, f1 = f0;
            // re-derive whether the fast path stood: run the fast conditions again
            const float d0 = tv.v[0] - p1;
            bool ok = true;
            const bool wr = PiAbs<float>::ge_pi(d0);
            float ph = p1, fr = f1;
            for (int w = 0; w < 4; w++) {
                const float diff = tv.v[w] - ph;
                if (!wr) ok = ok && !PiAbs<float>::ge_pi(diff);
                else ok = ok && PiAbs<float>::ge_pi(diff) && ((diff < 0) == (d0 < 0));
                const float err = wr ? WrapConst<float>::apply(diff, (d0 < 0) ? -1.f : 1.f) : diff;
                const float ff = fr + beta * err;
                ph = ph + ff + alpha * err;
                ok = ok && !PiAbs<float>::ge_2pi(ph);
                fr = PiAbs<float>::clamp(ff, -maxf, maxf);
            }
            vecs++;
            redo += ok ? 0 : 1;
        }
        if (threadIdx.x == 0) { g_redo = redo; g_vecs = vecs; }
    }
    long long c1 = clock64();
    if (threadIdx.x == 0) { *cyc = c1 - c0; state[0] = phase; state[1] = freq; }
}

int main()
{
    const long long n = 200000;
    std::vector<float> th(n + 1024);
    unsigned long long s = 12345;
    double car = 0.3;
    for (long long i = 0; i < n + 1024; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const double noise = ((double)(s >> 40) / 16777216.0 - 0.5) * 0.3;
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const int sym = ((i / 3) * 2654435761u >> 7) & 1;
        car += 2 * M_PI * 1000.0 / 50000.0;
        double a = car + (sym ? 1.06 : -1.06) + noise;
        a = fmod(a + M_PI, 2 * M_PI);
        if (a < 0) a += 2 * M_PI;
        th[i] = (float)(a - M_PI);
    }
    float *d_th, *d_ph, *d_st; long long *d_c;
    hipMalloc(&d_th, th.size() * 4); hipMalloc(&d_ph, th.size() * 4); hipMalloc(&d_c, 8); hipMalloc(&d_st, 8);
    hipMemcpy(d_th, th.data(), th.size() * 4, hipMemcpyHostToDevice);
    const float alpha = 0.005181347585199856f, beta = 6.742513059966792e-06f, maxf = 0.5654867f;
    for (int rep = 0; rep < 2; rep++) {
        long long c; float st[2];
        hipLaunchKernelGGL(k_walk<0>, dim3(1), dim3(64), 0, 0, d_th, d_ph, n, alpha, beta, maxf, d_c, d_st);
        hipDeviceSynchronize(); hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost); hipMemcpy(st, d_st, 8, hipMemcpyDeviceToHost);
        printf("wave walker (LDS, speculative): %.1f cycles/sample  state %.6f %.6f\n", (double)c / n, st[0], st[1]);
        hipLaunchKernelGGL(k_walk<1>, dim3(1), dim3(64), 0, 0, d_th, d_ph, n, alpha, beta, maxf, d_c, d_st);
        hipDeviceSynchronize(); hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost); hipMemcpy(st, d_st, 8, hipMemcpyDeviceToHost);
        printf("lane walker (global, full steps): %.1f cycles/sample  state %.6f %.6f\n", (double)c / n, st[0], st[1]);
        hipLaunchKernelGGL(k_walk<2>, dim3(1), dim3(1), 0, 0, d_th, d_ph, n, alpha, beta, maxf, d_c, d_st);
        hipDeviceSynchronize();
        unsigned long long redo, vecs;
        hipMemcpyFromSymbol(&redo, HIP_SYMBOL(g_redo), 8); hipMemcpyFromSymbol(&vecs, HIP_SYMBOL(g_vecs), 8);
        printf("speculation: %llu of %llu vectors redone (%.1f%%)\n", redo, vecs, 100.0 * redo / vecs);
    }
    return 0;
}
