// Probe (round 5): what a one-shot process pays before its first capture is in HBM.
// Times every step of a cold start the way libpdt takes them: runtime start, streams (default and with priorities), events,
// pinned staging, the big device buffers, the FIRST pinned -> device copies on a fresh stream, a first kernel launch.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_touch(float *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = 1.0f; }
#define STEP(name, ...) do { const double t0_ = now(); __VA_ARGS__; printf("  %-58s %8.2f ms\n", name, now() - t0_); } while (0)
int main()
{
    const double t_all = now();
    int ndev = 0;
    STEP("hipGetDeviceCount (runtime start)", (void)hipGetDeviceCount(&ndev));
    STEP("hipSetDevice(0)", (void)hipSetDevice(0));
    hipStream_t s0, s1, cs[4];
    STEP("hipStreamCreate (first stream)", (void)hipStreamCreate(&s0));
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    STEP("hipStreamCreateWithPriority (highest)", (void)hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, greatest));
    for (int q = 0; q < 4; q++) { char nm[64]; snprintf(nm, sizeof nm, "hipStreamCreateWithPriority (lowest) #%d", q); STEP(nm, (void)hipStreamCreateWithPriority(&cs[q], hipStreamNonBlocking, least)); }
    std::vector<hipEvent_t> ev(32);
    STEP("32 x hipEventCreateWithFlags", for (auto &e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming));
    void *pin = nullptr;
    STEP("hipHostMalloc 128 MiB", (void)hipHostMalloc(&pin, 128u << 20, hipHostMallocDefault));
    STEP("memset of the pinned 128 MiB (first touch)", memset(pin, 1, 128u << 20));
    void *d[8];
    const size_t big = (size_t)3600 << 20;
    for (int k = 0; k < 4; k++) { char nm[64]; snprintf(nm, sizeof nm, "hipMalloc 3.6 GB #%d", k); STEP(nm, (void)hipMalloc(&d[k], big)); }
    STEP("first kernel launch + sync (code object load)", hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s0, (float *)d[0], 64); (void)hipStreamSynchronize(s0));
    STEP("kernel that writes all of buffer 0 (3.6 GB) + sync", hipLaunchKernelGGL(k_touch, dim3((unsigned)(big / 4 / 256)), dim3(256), 0, s0, (float *)d[0], big / 4); (void)hipStreamSynchronize(s0));
    STEP("the same again", hipLaunchKernelGGL(k_touch, dim3((unsigned)(big / 4 / 256)), dim3(256), 0, s0, (float *)d[0], big / 4); (void)hipStreamSynchronize(s0));
    auto copies = [&](void *dst, int n, const char *nm) {
        const double t0 = now();
        for (int k = 0; k < n; k++) {
            (void)hipMemcpyAsync((char *)dst + (size_t)k * (8u << 20), (char *)pin + (size_t)(k % 16) * (8u << 20), 8u << 20, hipMemcpyHostToDevice, cs[k % 4]);
            (void)hipEventRecord(ev[k % 16], cs[k % 4]);
        }
        for (int q = 0; q < 4; q++) (void)hipStreamSynchronize(cs[q]);
        const double dt = now() - t0;
        printf("  %-58s %8.2f ms  (%.1f GB/s)\n", nm, dt, n * 8.0 / 1024 / (dt * 1e-3));
    };
    copies(d[1], 8, "first 8 copies of 8 MiB into a FRESH device buffer");
    copies(d[1], 442, "the other 442 copies into it (3.6 GB in all)");
    copies(d[1], 450, "3.6 GB into the same buffer again");
    copies(d[2], 450, "3.6 GB into another fresh buffer");
    copies(d[0], 450, "3.6 GB into the buffer a kernel has written");
    const double t_free = now();
    for (int k = 0; k < 4; k++) (void)hipFree(d[k]);
    printf("  %-58s %8.2f ms\n", "hipFree of the four buffers", now() - t_free);
    printf("  %-58s %8.2f ms\n", "total", now() - t_all);
    return 0;
}
