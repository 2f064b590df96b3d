// pdt_gather.hip -- libpdtgather.so: RCCL gather of decoded frame records (include/pdt_gather.h).  gfx950 / ROCm only.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/pdt_gather.h"

#define G_TRY(expr, code)                                                                        \
    do {                                                                                         \
        if ((expr) != 0) {                                                                       \
            fprintf(stderr, "libpdtgather: %s failed (%s:%d)\n", #expr, __FILE__, __LINE__);     \
            rc = (code);                                                                         \
            goto done;                                                                           \
        }                                                                                        \
    } while (0)

extern "C" int pdt_gather_frames(pdt_ctx *const *ctxs, int n, int root, pdt_frame **out, uint64_t *counts)
{
    if (n <= 0 || !ctxs || !out || !counts || root < 0 || root >= n) return PDT_ERR_ARG;
    int rc = PDT_OK;
    std::vector<int> dev((size_t)n);
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return PDT_ERR_ARG;
        dev[(size_t)i] = pdt_get_device(ctxs[i]);
        for (int j = 0; j < i; j++)
            if (dev[(size_t)j] == dev[(size_t)i]) return PDT_ERR_ARG;            // one rank per GPU
    }
    std::vector<ncclComm_t> comm((size_t)n, nullptr);
    std::vector<hipStream_t> st((size_t)n, nullptr);
    std::vector<unsigned long long *> d_cnt((size_t)n, nullptr), d_all((size_t)n, nullptr);
    std::vector<unsigned char *> d_rec((size_t)n, nullptr), d_gath((size_t)n, nullptr);
    std::vector<std::vector<pdt_frame>> mine((size_t)n);
    std::vector<unsigned long long> all((size_t)n, 0);
    size_t nmax = 1;
    *out = nullptr;
    G_TRY(ncclCommInitAll(comm.data(), n, dev.data()), PDT_ERR_NOGPU);
    // ---- counts: one all-gather of a 64-bit word per rank
    for (int i = 0; i < n; i++) {
        G_TRY(hipSetDevice(dev[(size_t)i]), PDT_ERR_NOGPU);
        G_TRY(hipStreamCreate(&st[(size_t)i]), PDT_ERR_NOGPU);
        const uint64_t nf = pdt_num_frames(ctxs[i]);
        mine[(size_t)i].resize((size_t)nf);
        if (nf) pdt_frames(ctxs[i], mine[(size_t)i].data(), nf);
        G_TRY(hipMalloc(&d_cnt[(size_t)i], sizeof(unsigned long long)), PDT_ERR_NOMEM);
        G_TRY(hipMalloc(&d_all[(size_t)i], sizeof(unsigned long long) * (size_t)n), PDT_ERR_NOMEM);
        const unsigned long long v = nf;
        G_TRY(hipMemcpyAsync(d_cnt[(size_t)i], &v, sizeof v, hipMemcpyHostToDevice, st[(size_t)i]), PDT_ERR_NOGPU);
        G_TRY(hipStreamSynchronize(st[(size_t)i]), PDT_ERR_NOGPU);
    }
    G_TRY(ncclGroupStart(), PDT_ERR_NOGPU);
    for (int i = 0; i < n; i++)
        G_TRY(ncclAllGather(d_cnt[(size_t)i], d_all[(size_t)i], 1, ncclUint64, comm[(size_t)i], st[(size_t)i]), PDT_ERR_NOGPU);
    G_TRY(ncclGroupEnd(), PDT_ERR_NOGPU);
    G_TRY(hipSetDevice(dev[(size_t)root]), PDT_ERR_NOGPU);
    G_TRY(hipMemcpyAsync(all.data(), d_all[(size_t)root], sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost, st[(size_t)root]),
          PDT_ERR_NOGPU);
    G_TRY(hipStreamSynchronize(st[(size_t)root]), PDT_ERR_NOGPU);
    for (int i = 0; i < n; i++) {
        counts[i] = all[(size_t)i];
        nmax = std::max(nmax, (size_t)all[(size_t)i]);
    }
    // ---- records, padded to the largest count
    for (int i = 0; i < n; i++) {
        G_TRY(hipSetDevice(dev[(size_t)i]), PDT_ERR_NOGPU);
        G_TRY(hipMalloc(&d_rec[(size_t)i], nmax * sizeof(pdt_frame)), PDT_ERR_NOMEM);
        G_TRY(hipMalloc(&d_gath[(size_t)i], nmax * sizeof(pdt_frame) * (size_t)n), PDT_ERR_NOMEM);
        G_TRY(hipMemsetAsync(d_rec[(size_t)i], 0, nmax * sizeof(pdt_frame), st[(size_t)i]), PDT_ERR_NOGPU);
        if (!mine[(size_t)i].empty())
            G_TRY(hipMemcpyAsync(d_rec[(size_t)i], mine[(size_t)i].data(), mine[(size_t)i].size() * sizeof(pdt_frame), hipMemcpyHostToDevice,
                                 st[(size_t)i]), PDT_ERR_NOGPU);
        G_TRY(hipStreamSynchronize(st[(size_t)i]), PDT_ERR_NOGPU);
    }
    G_TRY(ncclGroupStart(), PDT_ERR_NOGPU);
    for (int i = 0; i < n; i++)
        G_TRY(ncclAllGather(d_rec[(size_t)i], d_gath[(size_t)i], nmax * sizeof(pdt_frame), ncclUint8, comm[(size_t)i], st[(size_t)i]),
              PDT_ERR_NOGPU);
    G_TRY(ncclGroupEnd(), PDT_ERR_NOGPU);
    {
        size_t total = 0;
        for (int i = 0; i < n; i++) total += (size_t)counts[i];
        pdt_frame *res = (pdt_frame *)malloc(std::max<size_t>(total, 1) * sizeof(pdt_frame));
        if (!res) { rc = PDT_ERR_NOMEM; goto done; }
        std::vector<unsigned char> host(nmax * sizeof(pdt_frame) * (size_t)n);
        if (hipSetDevice(dev[(size_t)root]) != hipSuccess ||
            hipMemcpyAsync(host.data(), d_gath[(size_t)root], host.size(), hipMemcpyDeviceToHost, st[(size_t)root]) != hipSuccess ||
            hipStreamSynchronize(st[(size_t)root]) != hipSuccess) {
            free(res);
            rc = PDT_ERR_NOGPU;
            goto done;
        }
        size_t at = 0;
        for (int i = 0; i < n; i++) {
            memcpy(res + at, host.data() + (size_t)i * nmax * sizeof(pdt_frame), (size_t)counts[i] * sizeof(pdt_frame));
            at += (size_t)counts[i];
        }
        *out = res;
    }
done:
    for (int i = 0; i < n; i++) {
        (void)hipSetDevice(dev[(size_t)i]);
        if (d_cnt[(size_t)i]) (void)hipFree(d_cnt[(size_t)i]);
        if (d_all[(size_t)i]) (void)hipFree(d_all[(size_t)i]);
        if (d_rec[(size_t)i]) (void)hipFree(d_rec[(size_t)i]);
        if (d_gath[(size_t)i]) (void)hipFree(d_gath[(size_t)i]);
        if (st[(size_t)i]) (void)hipStreamDestroy(st[(size_t)i]);
        if (comm[(size_t)i]) (void)ncclCommDestroy(comm[(size_t)i]);
    }
    return rc;
}
