// pdt_gather.hip -- libpdtgather.so: RCCL gather of decoded frame records (include/pdt_gather.h).  gfx950 / ROCm only.
//
// Round 4: the communicators, streams and device buffers belong to a GATHERER that lives as long as the process needs it (one
// ncclCommInitAll per set of GPUs, not one per call: communicator set-up is tens of milliseconds per rank); the ragged <-> padded
// bookkeeping is in two host-only functions (pdt_gather_plan / pdt_gather_unpad) that bench.py's multi-process gather uses as
// well, so that there is ONE statement of the exchange format, exercised without a GPU by tests/test_gather_gloo.py.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
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

// ---------------------------------------------------------------- the exchange format (host only)
extern "C" int pdt_gather_plan(const uint64_t *counts, int n, uint64_t *nmax_out, uint64_t *offsets /* n + 1 */)
{
    if (!counts || n <= 0 || !nmax_out) return PDT_ERR_ARG;
    uint64_t nmax = 1, at = 0;                               // (at least one record per rank: a collective of zero bytes is no collective)
    for (int i = 0; i < n; i++) {
        nmax = std::max(nmax, counts[i]);
        if (offsets) offsets[i] = at;
        at += counts[i];
    }
    if (offsets) offsets[n] = at;
    *nmax_out = nmax;
    return PDT_OK;
}

extern "C" int pdt_gather_unpad(const void *padded, const uint64_t *counts, int n, uint64_t nmax, uint64_t record_bytes, void *out)
{
    if (!padded || !counts || n <= 0 || !out || !record_bytes) return PDT_ERR_ARG;
    size_t at = 0;
    for (int i = 0; i < n; i++) {
        if (counts[i] > nmax) return PDT_ERR_ARG;
        memcpy((unsigned char *)out + at, (const unsigned char *)padded + (size_t)i * nmax * record_bytes, (size_t)counts[i] * record_bytes);
        at += (size_t)counts[i] * record_bytes;
    }
    return PDT_OK;
}

// ---------------------------------------------------------------- the gatherer
struct pdt_gatherer {
    int n = 0;
    std::vector<int> dev;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> st;
    std::vector<unsigned long long *> d_cnt, d_all;
    std::vector<unsigned char *> d_rec, d_gath;
    std::vector<unsigned char *> pin;                        // pinned staging per rank (records up)
    unsigned char *pin_root = nullptr;                       // ... and for the root's copy down
    size_t cap = 0, pin_root_cap = 0;                        // records the per-rank buffers hold
    bool dead = false;                                       // a collective failed: the communicators may be unusable (aborted on close)
    std::mutex mu;
};

static void gatherer_free_buffers(pdt_gatherer *g)
{
    for (int i = 0; i < g->n; i++) {
        (void)hipSetDevice(g->dev[(size_t)i]);
        if (g->d_rec[(size_t)i]) (void)hipFree(g->d_rec[(size_t)i]);
        if (g->d_gath[(size_t)i]) (void)hipFree(g->d_gath[(size_t)i]);
        if (g->pin[(size_t)i]) (void)hipHostFree(g->pin[(size_t)i]);
        g->d_rec[(size_t)i] = g->d_gath[(size_t)i] = g->pin[(size_t)i] = nullptr;
    }
    if (g->pin_root) (void)hipHostFree(g->pin_root);
    g->pin_root = nullptr;
    g->cap = g->pin_root_cap = 0;
}

extern "C" int pdt_gatherer_open(const int *devices, int n, pdt_gatherer **out)
{
    if (!devices || n <= 0 || !out) return PDT_ERR_ARG;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++)
            if (devices[j] == devices[i]) return PDT_ERR_ARG;                    // one rank per GPU
    int rc = PDT_OK;
    pdt_gatherer *g = new pdt_gatherer;
    g->n = n;
    g->dev.assign(devices, devices + n);
    g->comm.assign((size_t)n, nullptr);
    g->st.assign((size_t)n, nullptr);
    g->d_cnt.assign((size_t)n, nullptr);
    g->d_all.assign((size_t)n, nullptr);
    g->d_rec.assign((size_t)n, nullptr);
    g->d_gath.assign((size_t)n, nullptr);
    g->pin.assign((size_t)n, nullptr);
    G_TRY(ncclCommInitAll(g->comm.data(), n, g->dev.data()), PDT_ERR_NOGPU);
    for (int i = 0; i < n; i++) {
        G_TRY(hipSetDevice(g->dev[(size_t)i]), PDT_ERR_NOGPU);
        G_TRY(hipStreamCreateWithFlags(&g->st[(size_t)i], hipStreamNonBlocking), PDT_ERR_NOGPU);
        G_TRY(hipMalloc(&g->d_cnt[(size_t)i], sizeof(unsigned long long)), PDT_ERR_NOMEM);
        G_TRY(hipMalloc(&g->d_all[(size_t)i], sizeof(unsigned long long) * (size_t)n), PDT_ERR_NOMEM);
    }
done:
    if (rc != PDT_OK) {
        pdt_gatherer_close(g);
        return rc;
    }
    *out = g;
    return PDT_OK;
}

extern "C" void pdt_gatherer_close(pdt_gatherer *g)
{
    if (!g) return;
    gatherer_free_buffers(g);
    for (int i = 0; i < g->n; i++) {
        (void)hipSetDevice(g->dev[(size_t)i]);
        if (g->d_cnt[(size_t)i]) (void)hipFree(g->d_cnt[(size_t)i]);
        if (g->d_all[(size_t)i]) (void)hipFree(g->d_all[(size_t)i]);
        if (g->st[(size_t)i]) (void)hipStreamDestroy(g->st[(size_t)i]);
        if (g->comm[(size_t)i]) (void)(g->dead ? ncclCommAbort(g->comm[(size_t)i]) : ncclCommDestroy(g->comm[(size_t)i]));
    }
    delete g;
}

// records[i] = counts_in[i] records of rank i in host memory (any memory; staged through the gatherer's pinned buffers)
extern "C" int pdt_gatherer_gather(pdt_gatherer *g, const pdt_frame *const *records, const uint64_t *counts_in, int root, pdt_frame **out,
                                   uint64_t *counts)
{
    if (!g || !records || !counts_in || !out || !counts || root < 0 || root >= g->n) return PDT_ERR_ARG;
    std::lock_guard<std::mutex> lock(g->mu);
    if (g->dead) return PDT_ERR_STATE;                       // an earlier collective failed: open a new gatherer
    const int n = g->n;
    int rc = PDT_OK;
    bool in_group = false;                                   // between ncclGroupStart and ncclGroupEnd
    std::vector<unsigned long long> all((size_t)n, 0);
    uint64_t nmax = 1;
    *out = nullptr;
    // ---- counts: one all-gather of a 64-bit word per rank (the ranks of a real deployment do not know each other's counts)
    for (int i = 0; i < n; i++) {
        G_TRY(hipSetDevice(g->dev[(size_t)i]), PDT_ERR_NOGPU);
        const unsigned long long v = counts_in[i];
        G_TRY(hipMemcpyAsync(g->d_cnt[(size_t)i], &v, sizeof v, hipMemcpyHostToDevice, g->st[(size_t)i]), PDT_ERR_NOGPU);
        G_TRY(hipStreamSynchronize(g->st[(size_t)i]), PDT_ERR_NOGPU);            // (v leaves scope)
    }
    G_TRY(ncclGroupStart(), PDT_ERR_NOGPU);
    in_group = true;
    for (int i = 0; i < n; i++)
        G_TRY(ncclAllGather(g->d_cnt[(size_t)i], g->d_all[(size_t)i], 1, ncclUint64, g->comm[(size_t)i], g->st[(size_t)i]), PDT_ERR_NOGPU);
    in_group = false;
    G_TRY(ncclGroupEnd(), PDT_ERR_NOGPU);
    G_TRY(hipSetDevice(g->dev[(size_t)root]), PDT_ERR_NOGPU);
    G_TRY(hipMemcpyAsync(all.data(), g->d_all[(size_t)root], sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost, g->st[(size_t)root]),
          PDT_ERR_NOGPU);
    G_TRY(hipStreamSynchronize(g->st[(size_t)root]), PDT_ERR_NOGPU);
    for (int i = 0; i < n; i++) counts[i] = all[(size_t)i];
    (void)pdt_gather_plan(counts, n, &nmax, nullptr);
    // ---- buffers grow with the largest count seen (they are kept between calls)
    if ((size_t)nmax > g->cap) {
        gatherer_free_buffers(g);
        const size_t want = (size_t)nmax + (size_t)nmax / 4 + 64;
        for (int i = 0; i < n; i++) {
            G_TRY(hipSetDevice(g->dev[(size_t)i]), PDT_ERR_NOGPU);
            G_TRY(hipMalloc(&g->d_rec[(size_t)i], want * sizeof(pdt_frame)), PDT_ERR_NOMEM);
            G_TRY(hipMalloc(&g->d_gath[(size_t)i], want * sizeof(pdt_frame) * (size_t)n), PDT_ERR_NOMEM);
            G_TRY(hipHostMalloc(&g->pin[(size_t)i], want * sizeof(pdt_frame), hipHostMallocDefault), PDT_ERR_NOMEM);
        }
        G_TRY(hipSetDevice(g->dev[(size_t)root]), PDT_ERR_NOGPU);
        G_TRY(hipHostMalloc(&g->pin_root, want * sizeof(pdt_frame) * (size_t)n, hipHostMallocDefault), PDT_ERR_NOMEM);
        g->pin_root_cap = want * (size_t)n;
        g->cap = want;
    }
    // ---- records, padded to the largest count (the padding is never looked at: no memset)
    for (int i = 0; i < n; i++) {
        if (!counts_in[i]) continue;
        G_TRY(hipSetDevice(g->dev[(size_t)i]), PDT_ERR_NOGPU);
        memcpy(g->pin[(size_t)i], records[i], (size_t)counts_in[i] * sizeof(pdt_frame));
        G_TRY(hipMemcpyAsync(g->d_rec[(size_t)i], g->pin[(size_t)i], (size_t)counts_in[i] * sizeof(pdt_frame), hipMemcpyHostToDevice,
                             g->st[(size_t)i]), PDT_ERR_NOGPU);
    }
    G_TRY(ncclGroupStart(), PDT_ERR_NOGPU);
    in_group = true;
    for (int i = 0; i < n; i++)
        G_TRY(ncclAllGather(g->d_rec[(size_t)i], g->d_gath[(size_t)i], (size_t)nmax * sizeof(pdt_frame), ncclUint8, g->comm[(size_t)i],
                            g->st[(size_t)i]), PDT_ERR_NOGPU);
    in_group = false;
    G_TRY(ncclGroupEnd(), PDT_ERR_NOGPU);
    {
        uint64_t total = 0;
        for (int i = 0; i < n; i++) total += counts[i];
        pdt_frame *res = (pdt_frame *)malloc(std::max<size_t>((size_t)total, 1) * sizeof(pdt_frame));
        if (!res) { rc = PDT_ERR_NOMEM; goto done; }
        if (hipSetDevice(g->dev[(size_t)root]) != hipSuccess ||
            hipMemcpyAsync(g->pin_root, g->d_gath[(size_t)root], (size_t)nmax * sizeof(pdt_frame) * (size_t)n, hipMemcpyDeviceToHost,
                           g->st[(size_t)root]) != hipSuccess ||
            hipStreamSynchronize(g->st[(size_t)root]) != hipSuccess) {
            free(res);
            rc = PDT_ERR_NOGPU;
            goto done;
        }
        (void)pdt_gather_unpad(g->pin_root, counts, n, nmax, sizeof(pdt_frame), res);
        *out = res;
    }
done:
    // a failure inside a group leaves it open, and a failed collective leaves the communicators in an unknown state: close the
    // group, and never use this gatherer again (pdt_gather_frames evicts it from its cache; pdt_gatherer_close aborts it)
    if (in_group) (void)ncclGroupEnd();
    if (rc == PDT_ERR_NOGPU) g->dead = true;
    for (int i = 0; i < n; i++) {                            // (every rank's stream is idle when this returns)
        (void)hipSetDevice(g->dev[(size_t)i]);
        (void)hipStreamSynchronize(g->st[(size_t)i]);
    }
    return rc;
}

// ---------------------------------------------------------------- the one-call form: a gatherer per set of GPUs, kept for the process
static std::mutex g_cache_mu;
static std::vector<pdt_gatherer *> g_cache;

extern "C" void pdt_gather_shutdown(void)
{
    std::lock_guard<std::mutex> lock(g_cache_mu);
    for (pdt_gatherer *g : g_cache) pdt_gatherer_close(g);
    g_cache.clear();
}

extern "C" int pdt_gather_frames(pdt_ctx *const *ctxs, int n, int root, pdt_frame **out, uint64_t *counts)
{
    if (n <= 0 || !ctxs || !out || !counts || root < 0 || root >= n) return PDT_ERR_ARG;
    std::vector<int> dev((size_t)n);
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return PDT_ERR_ARG;
        dev[(size_t)i] = pdt_get_device(ctxs[i]);
    }
    pdt_gatherer *g = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        for (pdt_gatherer *c : g_cache)
            if (c->dev == dev) g = c;
        if (!g) {
            const int rc = pdt_gatherer_open(dev.data(), n, &g);
            if (rc != PDT_OK) return rc;
            g_cache.push_back(g);
        }
    }
    std::vector<std::vector<pdt_frame>> mine((size_t)n);
    std::vector<const pdt_frame *> ptr((size_t)n, nullptr);
    std::vector<uint64_t> cnt((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        cnt[(size_t)i] = pdt_num_frames(ctxs[i]);
        mine[(size_t)i].resize((size_t)cnt[(size_t)i]);
        if (cnt[(size_t)i]) pdt_frames(ctxs[i], mine[(size_t)i].data(), cnt[(size_t)i]);
        ptr[(size_t)i] = mine[(size_t)i].data();
    }
    const int rc = pdt_gatherer_gather(g, ptr.data(), cnt.data(), root, out, counts);
    if (g->dead) {                                           // the next call opens a fresh set of communicators
        std::lock_guard<std::mutex> lock(g_cache_mu);
        g_cache.erase(std::remove(g_cache.begin(), g_cache.end(), g), g_cache.end());
        pdt_gatherer_close(g);
    }
    return rc;
}
