// pdt_api.hip -- libpdt.so: context management, kernel orchestration and the C ABI of
// include/pdt.h.  Compiled for gfx950 only, with -ffp-contract=off (see pdt_device_math.h).
#include "pdt_rt.h"

#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>

namespace pdtrt {
PDT_CHAIN_INSTANCES(extern, float)
PDT_CHAIN_INSTANCES(extern, double)
std::atomic<long long> g_alloc_ns{0};
}  // namespace pdtrt

static std::atomic<int> g_open_contexts{0};          // contexts alive in this process (pdt_open / pdt_close)
// One ingest per GPU at a time.  Two contexts on one GPU (bin/demodMulti's two lanes) that read their captures at once would
// share the PCIe link and both arrive late; taking turns, the second one's capture arrives while the first one's chain runs.
static std::mutex g_link_mu[64];

static std::mutex g_dev_mu;
static std::map<std::string, std::string> g_dev;
void Tuning::load()
{
    std::lock_guard<std::mutex> lock(g_dev_mu);
    if (g_dev.empty()) return;
    auto get = [&](const char *n) -> const char * { auto it = g_dev.find(n); return it == g_dev.end() ? nullptr : it->second.c_str(); };
    if (const char *e = get("PDT_PLL_WARM_SCALE")) pll_warm_scale = atof(e);
    if (const char *e = get("PDT_HEAD_TAUS")) head_taus = atof(e);
    if (const char *e = get("PDT_BAND_PAD")) band_pad = atof(e);
    if (const char *e = get("PDT_PLL_WARM_S")) pll_warm_s = atof(e);
    if (const char *e = get("PDT_AGC_WARM_S")) agc_warm_s = atof(e);
    if (const char *e = get("PDT_AGC_K")) agc_k = atof(e);
    if (const char *e = get("PDT_AGC_TPB")) agc_tpb = std::max(1, atoi(e));
    if (const char *e = get("PDT_PLL_BLOCK")) pll_block = atoi(e);
    if (const char *e = get("PDT_HBM_LIMIT_MB")) hbm_limit_mb = std::max(1ll, atoll(e));        // (tests: pretend the device has this much free memory)
    if (const char *e = get("PDT_WINDOW_PIECE")) window_piece = std::max(1ll, atoll(e));          // (tests: samples per piece of the bounded window)
    if (const char *e = get("PDT_SCOUT_SYMS")) scout_syms = std::max(16, atoi(e));
    if (const char *e = get("PDT_FIR_WG_PER_CU")) fir_wg_per_cu = std::min(4096, std::max(1, atoi(e)));
    if (const char *e = get("PDT_INGEST_THREADS")) ingest_threads = std::min(128, std::max(1, atoi(e)));
    if (const char *e = get("PDT_INGEST_SPAN_MB")) ingest_span_mb = std::min(256, std::max(1, atoi(e)));
    if (const char *e = get("PDT_INGEST_STREAMS")) ingest_streams = std::min(4, std::max(1, atoi(e)));
    if (const char *e = get("PDT_INGEST_DIRECT")) ingest_direct = atoi(e) ? 1 : 0;
    if (const char *e = get("PDT_INGEST_NUMA")) ingest_numa = atoi(e) ? 1 : 0;
    if (const char *e = get("PDT_FIX_PASSES")) fix_passes = atoi(e);
    if (const char *e = get("PDT_GSEG")) gseg = std::min(64, std::max(2, atoi(e)));
    if (const char *e = get("PDT_GSPAN")) gspan = std::min(256, std::max(1, atoi(e)));
    if (const char *e = get("PDT_GSPAN_CAP")) gspan_cap = std::max(64, atoi(e));          // (tests: rows that do not fit the key list)
    fir_generic = get("PDT_FIR_GENERIC") != nullptr;
    mix_unfused = get("PDT_MIX_UNFUSED") != nullptr;
    quality_inline = get("PDT_QUALITY_INLINE") != nullptr;
    gardner_nostride = get("PDT_GARDNER_NOSTRIDE") != nullptr;
    gemit_groups = get("PDT_GEMIT_GROUPS") != nullptr;
    agc_unfused = get("PDT_AGC_UNFUSED") != nullptr;
    agc_lanes = get("PDT_AGC_LANES") != nullptr;           // the per-lane walkers of rounds 1 - 3 (k_agc_block)
    no_excl = get("PDT_NO_EXCL") != nullptr;
    gardner_onebuf = get("PDT_GARDNER_ONEBUF") != nullptr;
    gardner_noring = get("PDT_GARDNER_NORING") != nullptr;
    ema_noguess = get("PDT_EMA_NOGUESS") != nullptr;
    gardner_sequential = get("PDT_GARDNER_SEQUENTIAL") != nullptr;
    seg_sequential = get("PDT_SEG_SEQUENTIAL") != nullptr;
    overlap = get("PDT_NO_OVERLAP") == nullptr;            // (round 4: off unless PDT_OVERLAP; round 5: on)
    if (const char *e = get("PDT_OVERLAP_SPLIT")) {          // "0.55,0.28,0.17": the segments' fractions of the capture
        int k = 0;
        for (const char *q = e; *q && k < 8; k++) {
            char *end = nullptr;
            overlap_split[k] = strtod(q, &end);
            if (end == q) break;
            q = (*end == ',') ? end + 1 : end;
        }
    }
    chain_one_range = get("PDT_CHAIN_ONE_RANGE") != nullptr;
    debug_overlap = get("PDT_DEBUG_OVERLAP") != nullptr;
    if (const char *e = get("PDT_OVERLAP_SEGMENTS")) overlap_segments = std::min(64, std::max(1, atoi(e)));
    if (const char *e = get("PDT_OVERLAP_MIN_MB")) overlap_min_mb = std::min(1 << 20, std::max(1, atoi(e)));
    debug_sync = get("PDT_DEBUG_SYNC") != nullptr;
    pll_noshort = get("PDT_PLL_NOSHORT") != nullptr;
    pll_nockpt = get("PDT_PLL_NOCKPT") != nullptr;
    pll_noconsensus = get("PDT_PLL_NOCONSENSUS") != nullptr;
    pll_notail = get("PDT_PLL_NOTAIL") != nullptr;
    seg_plain = get("PDT_SEG_PLAIN") != nullptr;
    sync_block = get("PDT_SYNC_BLOCK") != nullptr;
}

namespace pdtrt {

// Run the plans of M contexts (all of one shape, see Plan::same_shape) as one sequence of operations on the streams of
// ctxs[0]: each launch covers the M captures through grid.z, memsets and copies are issued per capture.  No host wait.
int execute_plans(pdt_ctx *const *ctxs, int M)
{
    pdt_ctx *L0 = ctxs[0];
    HIP_TRY(hipSetDevice(L0->cfg.device));
    const Plan &P0 = L0->plan;
    for (int m = 1; m < M; m++)
        if (!P0.same_shape(ctxs[m]->plan)) return PDT_ERR_STATE;
    const size_t per = P0.packs.size();
    const size_t total = per * (size_t)M;
    int rc;
    if ((rc = L0->packs_dev.ensure(total + 64))) return rc;
    if (total + 64 > L0->packs_pin_cap) {
        if (L0->packs_pin) (void)hipHostFree(L0->packs_pin);
        L0->packs_pin = nullptr;
        L0->packs_pin_cap = 0;
        const size_t want = 2 * total + 4096;
        if (timed_host_malloc((void **)&L0->packs_pin, want) != hipSuccess) { (void)hipGetLastError(); return PDT_ERR_NOMEM; }
        L0->packs_pin_cap = want;
    }
    // pack of capture m for the launch at offset `off`: M * off + m * size
    unsigned char *pin = (unsigned char *)L0->packs_pin;
    for (const PlanOp &o : P0.ops) {
        if (o.op != OP_LAUNCH) continue;
        for (int m = 0; m < M; m++)
            memcpy(pin + (size_t)M * o.pack_off + (size_t)m * o.pack_size, ctxs[m]->plan.packs.data() + o.pack_off, o.pack_size);
    }
    hipStream_t st[2] = { L0->stream, L0->stream2 };
    if (total) HIP_TRY(hipMemcpyAsync(L0->packs_dev.p, pin, total, hipMemcpyHostToDevice, st[0]));
    Launcher T(L0);
    const unsigned char *dev = (const unsigned char *)L0->packs_dev.p;
    for (size_t i = 0; i < P0.ops.size(); i++) {
        const PlanOp &o = P0.ops[i];
        switch (o.op) {
        case OP_LAUNCH: {
            dim3 g = o.grid;
            for (int m = 1; m < M; m++) {
                const dim3 gm = ctxs[m]->plan.ops[i].grid;
                g.x = std::max(g.x, gm.x);
                g.y = std::max(g.y, gm.y);
            }
            g.z = (unsigned)M;
            o.go(g, o.block, o.shmem, st[o.side], dev + (size_t)M * o.pack_off);
            if (L0->tune.debug_sync) {                                  // developer switch: find the launch that faults
                const hipError_t e = hipStreamSynchronize(st[o.side]);
                fprintf(stderr, "libpdt: [%zu] %s grid (%u,%u,%u) block %u lds %zu: %s\n", i, o.name, g.x, g.y, g.z, o.block.x, o.shmem,
                        hipGetErrorString(e));
            }
            break;
        }
        case OP_MEMSET:
            for (int m = 0; m < M; m++) {
                const PlanOp &om = ctxs[m]->plan.ops[i];
                if (om.bytes) HIP_TRY(hipMemsetAsync(om.dst, om.value, om.bytes, st[om.side]));
            }
            T.gap();
            break;
        case OP_D2H:
        case OP_H2D:
            for (int m = 0; m < M; m++) {
                const PlanOp &om = ctxs[m]->plan.ops[i];
                if (om.bytes)
                    HIP_TRY(hipMemcpyAsync(om.dst, om.src, om.bytes, o.op == OP_D2H ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, st[0]));
            }
            T.gap();
            break;
        case OP_FORK:
            HIP_TRY(hipEventRecord(L0->ev_fork, st[0]));
            HIP_TRY(hipStreamWaitEvent(st[1], L0->ev_fork, 0));
            break;
        case OP_JOIN_RECORD: HIP_TRY(hipEventRecord(L0->ev_join, st[1])); break;
        case OP_JOIN_WAIT: HIP_TRY(hipStreamWaitEvent(st[0], L0->ev_join, 0)); break;
        case OP_TBEGIN: T.begin(o.name, st[o.side]); break;
        case OP_TEND: T.end(); break;
        case OP_TGAP: T.gap(); break;
        case OP_EV0: HIP_TRY(hipEventRecord(L0->ev0, st[0])); break;
        case OP_EV1: HIP_TRY(hipEventRecord(L0->ev1, st[0])); break;
        }
    }
    HIP_TRY(hipGetLastError());
    for (int m = 0; m < M; m++) ctxs[m]->leader = L0;
    return PDT_OK;
}

void launch_bytesync(pdt_ctx *ctx, Plan &PL, hipStream_t st, const SyncParams &SP, DevScalars *d_sc, long long bit_cap, uint32_t hit_cap,
                     uint32_t frame_cap, long long min_pos)
{
    unsigned char *d_bits = (unsigned char *)ctx->bits.p;
    unsigned *d_bitsym = (unsigned *)ctx->bitsym.p;
    long long *d_symidx = (long long *)ctx->symidx.p;
    unsigned *d_hits = (unsigned *)ctx->hits.p;
    FrameRec *d_frames = (FrameRec *)ctx->frames.p;
    const long long n_stiles = (bit_cap + 4095) / 4096;
    SyncTile *d_stiles = (SyncTile *)ctx->stiles.p;
    PDT_LAUNCH(256, k_sync_hits_tile, dim3((unsigned)n_stiles), dim3(256), 0, st, d_bits, &d_sc->nbits, SP, d_stiles,
                       &d_sc->sync_overflow, min_pos);
    PDT_LAUNCH(PDT_SYNC_THREADS, k_sync_frames_tiles, dim3(1), dim3(PDT_SYNC_THREADS), 0, st, (const SyncTile *)d_stiles, &d_sc->nbits, SP, d_hits, hit_cap,
                       d_frames, &d_sc->nframes, frame_cap, &d_sc->sync_overflow, (unsigned *)ctx->sync_scr.p);
    // generic path (atomic append + sort), only when a tile overflowed
    const long long grid = (bit_cap + 255) / 256;
    PDT_LAUNCH(256, k_sync_hits, dim3((unsigned)grid), dim3(256), 0, st, d_bits, &d_sc->nbits, SP, d_hits, &d_sc->nhits, hit_cap,
                       &d_sc->sync_overflow, min_pos);
    PDT_LAUNCH(256, k_sync_frames, dim3(1), dim3(256), 0, st, d_hits, &d_sc->nhits, hit_cap, SP, d_frames, &d_sc->nframes,
                       frame_cap, &d_sc->sync_overflow);
    PDT_LAUNCH(128, k_frame_pack, dim3(frame_cap), dim3(128), 0, st, d_bits, &d_sc->nbits, d_bitsym, d_symidx, SP, d_frames,
                       &d_sc->nframes, frame_cap);
}

// reports of chunks [c0, c1) of the capture (c1 <= chunk_host.size()); `total`: the capture's length as far as it is known
// `open_frame`: a frame whose sync word has been seen and whose last byte has not (a segment's end) -- counted where ByteSync
// counts it, at the sync word
void chunk_reports_range(const pdt_ctx *ctx, uint64_t c0, uint64_t c1, uint64_t total, pdt_chunk_report *out, const pdt_frame *open_frame)
{
    const bool argos = ctx->cfg.mode == PDT_MODE_ARGOS;
    const uint64_t chunk = ctx->cfg.chunk;
    uint64_t sym_prev = c0 ? ctx->chunk_host[(size_t)c0 - 1].sym_upto : 0, bits_prev = c0 ? ctx->chunk_host[(size_t)c0 - 1].bits_upto : 0;
    size_t f = 0;
    if (c0) {                                            // first frame whose sync word was completed behind chunk c0 - 1
        size_t lo = 0, hi = ctx->frames_host.size();
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if ((uint64_t)ctx->frames_host[mid].bit_index < bits_prev) lo = mid + 1; else hi = mid;
        }
        f = lo;
    }
    for (uint64_t c = c0; c < c1; c++) {
        const pdt::ChunkInfo &ci = ctx->chunk_host[(size_t)c];
        pdt_chunk_report &o = out[c - c0];
        memset(&o, 0, sizeof o);
        o.samples = std::min<uint64_t>(chunk, total - c * chunk);
        o.avg_phase = ci.avg_phase;
        o.symbols = ci.sym_upto - sym_prev;
        o.bits = ci.bits_upto - bits_prev;
        // ByteSyncOnSyncword / FindSyncWords count a frame at the bit that completes its sync word (ByteSync.c:105,139)
        while (f < ctx->frames_host.size() && (uint64_t)ctx->frames_host[f].bit_index < ci.bits_upto) { o.frames++; f++; }
        if (open_frame && (uint64_t)open_frame->bit_index >= bits_prev && (uint64_t)open_frame->bit_index < ci.bits_upto) o.frames++;
        // waveDataTime[0] as the progress line prints it: POES keeps the input time axis apart (main.c:424,438,445), ARGOS
        // compacts the symbol and bit times into it (ARGOSdemod/main.c:278,282)
        o.time0 = (argos && ctx->elem == 8) ? const_cast<pdt_ctx *>(ctx)->axis_d.at((uint64_t)ci.t0_src + 1)
                  : argos ? (double)const_cast<pdt_ctx *>(ctx)->axis_f.at((uint64_t)ci.t0_src + 1)
                          : (double)const_cast<pdt_ctx *>(ctx)->axis_f.at(c * chunk + 1);
        sym_prev = ci.sym_upto;
        bits_prev = ci.bits_upto;
    }
}


}  // namespace pdtrt

namespace {

// ---------------------------------------------------------------- host -> HBM ingest
// The capture (a file the caller opened, or host memory) is cut into 2 MiB spans.  A few host threads bring the spans into
// pinned slots -- pread from the page cache resp. memcpy, ~5 GB/s per thread -- and queue one asynchronous copy per span on a
// copy stream; a slot is refilled when its previous copy has completed.  The demodulation stream then waits for the last
// copy.  (One pageable hipMemcpy of the whole capture runs at a fraction of the link rate and cannot start before the file
// has been read.)  Small captures take the plain copy.
struct IngestSrc {
    const unsigned char *mem = nullptr;   // host memory, or
    int fd = -1;                          // an open file ...
    uint64_t off = 0;                     // ... and the byte offset of the first sample
};
constexpr size_t PDT_INGEST_SPAN_DEFAULT = 2u << 20;
constexpr int PDT_INGEST_SLOTS = 2;      // per thread

// An ingest that goes on in the background: every span has a flag that says its copy has been queued; the caller makes its
// stream wait for the spans of a prefix (ingest_wait_mark) and starts work on it while the rest still arrives, and joins the
// threads at the end (ingest_join).
struct IngestJob {
    std::vector<std::thread> pool;
    std::unique_ptr<std::atomic<int>[]> submitted;
    std::unique_ptr<std::atomic<int>[]> slot_state;                  // 0 free, 1 filled (to be copied), 2 copy in flight
    std::unique_ptr<size_t[]> slot_span;                             // the span a filled slot holds (written before state 1)
    std::unique_ptr<size_t[]> slot_delta;                            // ... and where in the slot its first byte lies (O_DIRECT reads start on a 4 KiB boundary of the file)
    int fd_direct = -1;                                              // the capture file once more, opened with O_DIRECT (or -1)
    ~IngestJob() { if (fd_direct >= 0) close(fd_direct); }
    std::atomic<int> failed{0};
    size_t nspans = 0, span = 0, waited = 0;
    hipStream_t cs[4] = { nullptr, nullptr, nullptr, nullptr };      // the copy streams the spans go round robin over
    int ncs = 0;
    // prefixes the caller will wait for (bytes, ascending; set before ingest_capture): when the last span of prefix m has been
    // queued the submitter records one event per copy stream -- behind that prefix's copies and in front of everything later --
    // and raises mark_ready[m]
    std::vector<size_t> mark_bytes;
    std::vector<size_t> mark_spans;
    std::unique_ptr<std::atomic<int>[]> mark_ready;
};

// ---- where a capture file's bytes are, and where they should be read (round 6)
// Every byte of a capture crosses host memory three times on its way -- out of the page cache, into a pinned slot, out again by
// the DMA engine -- unless the file is read with O_DIRECT straight into the pinned slots: two.  At N = 8 GPUs the first is
// ~1.3 TB/s of host traffic against the ~1.0 - 1.1 the two sockets deliver.  O_DIRECT is right for a file that is NOT
// in the page cache (it then comes from the device either way; the copy through the cache is pure overhead) and wrong for one
// that is (tmpfs does not offer it at all): a probe of the file's pages decides.  The same probe says on which NUMA node the
// cached pages lie (move_pages); what to do with that is ingest_capture's decision (measured: see there).
struct FilePlacement {
    double resident = 1.0;      // fraction of the sampled pages that are in the page cache (1 = nothing to gain from O_DIRECT)
    int node = -1;              // the NUMA node that holds most of the resident ones (-1 = unknown / mixed)
};
static FilePlacement probe_file(int fd, uint64_t off, size_t bytes)
{
    FilePlacement fp;
    const long page = sysconf(_SC_PAGESIZE);
    if (page <= 0 || bytes < (size_t)(64 * page)) return fp;
    const uint64_t a0 = off / (uint64_t)page * (uint64_t)page;
    const size_t len = (size_t)(off + bytes - a0);
    void *m = mmap(nullptr, len, PROT_READ, MAP_SHARED, fd, (off_t)a0);
    if (m == MAP_FAILED) return fp;
    constexpr int NS = 64;
    const size_t npages = (len + (size_t)page - 1) / (size_t)page;
    unsigned char vec = 0;
    void *where[NS];
    int got = 0, res = 0;
    for (int k = 0; k < NS; k++) {
        const size_t pg = (size_t)((double)k + 0.5) * npages / NS;
        void *addr = (unsigned char *)m + pg * (size_t)page;
        if (mincore(addr, (size_t)page, &vec) != 0) continue;
        got++;
        if (vec & 1) where[res++] = addr;
    }
    if (got) fp.resident = (double)res / (double)got;
    if (res >= NS / 2) {
        // the node of every resident sample page: touch it (maps the page-cache page into this process), then ask
        int status[NS];
        volatile unsigned char sink = 0;
        for (int k = 0; k < res; k++) { sink = sink + *(volatile unsigned char *)where[k]; status[k] = -1; }
        if (syscall(SYS_move_pages, 0, (unsigned long)res, where, nullptr, status, 0) == 0) {
            int count[64] = {0}, best = -1;
            for (int k = 0; k < res; k++)
                if (status[k] >= 0 && status[k] < 64) count[status[k]]++;
            for (int nd = 0; nd < 64; nd++)
                if (count[nd] * 5 >= res * 4) best = nd;             // four pages in five on one node
            fp.node = best;
        }
    }
    munmap(m, len);
    return fp;
}
// the NUMA node a GPU hangs on, and that node's CPUs
static int gpu_numa_node(int device)
{
    char bus[64] = {0}, path[160];
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *c = bus; *c; c++) *c = (char)tolower((unsigned char)*c);
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    int node = -1;
    if (f) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    return node;
}
static bool node_cpus(int node, cpu_set_t *set)
{
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    CPU_ZERO(set);
    int a, b, any = 0;
    char sep;
    while (fscanf(f, "%d", &a) == 1) {
        b = a;
        if (fscanf(f, "%c", &sep) == 1 && sep == '-') {
            if (fscanf(f, "%d", &b) != 1) break;
            if (fscanf(f, "%c", &sep) != 1) sep = 0;
        }
        for (int c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET(c, set); any = 1; }
        if (sep != ',') break;
    }
    fclose(f);
    return any != 0;
}

// Round 5: ONE thread talks to the HIP runtime.  The readers only fill pinned slots (pread / memcpy) and raise a flag; the
// submitter -- the calling thread, or one more background thread when the ingest runs beside the chain -- queues the copies, one
// event per slot, and hands a slot back to its reader when its copy has completed (hipEventQuery on the oldest copy in flight
// of each copy stream).  With eight readers calling hipMemcpyAsync / hipEventRecord / hipEventSynchronize themselves the
// runtime's locks were contended: beside a chain that is launching kernels the ingest fell from 52 to ~41 GB/s, and a burst of
// fifty launches took up to 7 ms instead of 0.2 (round 4: "the readers contend with the launches in the runtime").
int ingest_capture(pdt_ctx *ctx, const IngestSrc &src, size_t bytes, void *dst, IngestJob *job = nullptr)
{
    if (!bytes) return PDT_OK;
    // (hour-long captures: 8 MiB spans -- fewer, larger copies: 3.6 GB in 75-95 ms against 110 with 2 MiB spans)
    const size_t PDT_INGEST_SPAN = ctx->tune.ingest_span_mb > 0 ? ((size_t)ctx->tune.ingest_span_mb << 20)
                                   : (bytes >= ((size_t)512 << 20) ? 4 * PDT_INGEST_SPAN_DEFAULT : PDT_INGEST_SPAN_DEFAULT);
    const size_t nspans = (bytes + PDT_INGEST_SPAN - 1) / PDT_INGEST_SPAN;
    if (src.mem && nspans <= 2 && !job) {
        HIP_TRY(hipMemcpyAsync(dst, src.mem, bytes, hipMemcpyHostToDevice, ctx->stream));
        return PDT_OK;
    }
    unsigned hw = std::thread::hardware_concurrency();
    // (round 4, 3.6 GB from tmpfs: 4 / 6 / 8 / 12 / 16 reader threads -> 100 / 92 / 88 / 90 / 89 ms for the whole call)
    // Several contexts of one process (bin/demodMulti: two per GPU) share the host's cores: half of them, divided by the open
    // contexts, never fewer than two readers (8 GPUs on a 256-thread host: 8 each; on a 32-thread host: 2 each).
    const unsigned share = std::max(2u, (hw ? hw / 2u : 8u) / (unsigned)std::max(1, g_open_contexts.load()));
    const unsigned t_max = ctx->tune.ingest_threads > 0 ? (unsigned)ctx->tune.ingest_threads : std::min(8u, share);
    int T = (int)std::min<size_t>(std::min<unsigned>(hw ? hw / 2u + 1u : 4u, t_max), nspans);
    if (T < 1) T = 1;
    const int nslots = T * PDT_INGEST_SLOTS;
    // ---- how the file is read (see probe_file): O_DIRECT straight into the pinned slots when its pages are not cached; readers
    // and staging on the GPU's NUMA node when -- and only when -- its cached pages lie there
    struct FdGuard { int fd = -1; ~FdGuard() { if (fd >= 0) close(fd); } } direct_guard;
    int fd_direct = -1, bind_node = -1;
    constexpr size_t DIO = 4096;                                   // alignment of O_DIRECT reads (offset, length, address)
    if (!src.mem && bytes >= ((size_t)64 << 20)) {
        FilePlacement fp;
        if (ctx->tune.ingest_direct < 0 || ctx->tune.ingest_numa < 0) fp = probe_file(src.fd, src.off, bytes);
        const bool want_direct = ctx->tune.ingest_direct > 0 || (ctx->tune.ingest_direct < 0 && fp.resident < 0.1);
        if (want_direct) {
            char link[64];
            snprintf(link, sizeof link, "/proc/self/fd/%d", src.fd);
            fd_direct = open(link, O_RDONLY | O_DIRECT | O_CLOEXEC);        // (EINVAL where the file system has no direct I/O: tmpfs)
            direct_guard.fd = fd_direct;
        }
        const int gnode = gpu_numa_node(ctx->cfg.device);
        // (Measured, round 6, one box -- GPU and the capture's pages both on node 1: readers and staging bound there 100 - 106 ms for
        // the 3.6 GB file -> frame file path, not bound 94.6, profiles/r6/e2e_ab_numa.txt: eight readers crowd one socket's cores
        // beside the box's other tenants, as in round 4.  So buffered reads are bound on request only (PDT_INGEST_NUMA); direct
        // reads -- the device writes the staging memory, the GPU's DMA engine reads it -- keep staging and readers on the GPU's node.)
        if (gnode >= 0 && (ctx->tune.ingest_numa > 0 || (ctx->tune.ingest_numa < 0 && fd_direct >= 0))) bind_node = gnode;
        (void)fp.node;
    }
    ctx->ingest_was_direct = fd_direct >= 0 ? 1 : 0;
    ctx->ingest_numa_node = bind_node;
    cpu_set_t node_set;
    const bool have_cpus = bind_node >= 0 && node_cpus(bind_node, &node_set);
    const size_t slot_stride = PDT_INGEST_SPAN + (fd_direct >= 0 ? 2 * DIO : 0);
    const size_t need = (size_t)nslots * slot_stride;
    // The copy streams live at the LOWEST stream priority: streams of one priority share a few hardware queues, and a copy stream
    // that lands on the queue of the demodulation stream waits behind that stream's kernels -- beside a running segment (a 5 ms PLL
    // kernel) one of four copy streams stood still, and with it the ring of pinned slots: the overlapped ingest crawled at
    // ~25 GB/s while kernels ran (round 5, tools/jobs/r5_e2e_ab.sh: four streams 95 ms, two -- no sharing -- 82 ms).  Another
    // priority is another set of queues (the side stream uses the highest for the same reason).
    auto make_copy_stream = [](hipStream_t *out) -> hipError_t {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (least != greatest && hipStreamCreateWithPriority(out, hipStreamNonBlocking, least) == hipSuccess) return hipSuccess;
        (void)hipGetLastError();
        return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    };
    const auto t_setup0 = std::chrono::steady_clock::now();
    // copy streams in use: hour-long captures spread their span copies over four (3.6 GB: 123 -> 107 ms for the whole call,
    // ~51 GB/s from the page cache to HBM); ten-minute captures are no faster for it
    const int NS = std::min(4, std::max(1, ctx->tune.ingest_streams > 0 ? ctx->tune.ingest_streams : (bytes >= ((size_t)512 << 20) ? 4 : 1)));
    // (A context's first ingest sets these up, and a one-shot process -- bin/demodPOES -- pays for it in full: a stream is ~10 ms
    // to create, its first copy ~6 ms more (the runtime starts its DMA queue then), the pinned staging 25 - 30 ms:
    // tools/probes/cold_path_probe.hip.  Setting them up side by side, a thread each, was measured and made it WORSE -- the pinned
    // allocation alone then took 0.1 - 0.7 s: the runtime serialises them, badly.  One after the other.)
    if (need > ctx->ingest_pin_cap) {
        if (ctx->ingest_pin) (void)hipHostFree(ctx->ingest_pin);
        ctx->ingest_pin = nullptr;
        ctx->ingest_pin_cap = 0;
        // (the staging memory on the node the readers run on: MPOL_PREFERRED = 1 for this thread while the runtime allocates and
        // pins it, MPOL_DEFAULT = 0 again afterwards)
        unsigned long mask[16] = {0};
        if (bind_node >= 0 && bind_node < 1024) {
            mask[bind_node / (8 * sizeof(long))] = 1ul << (bind_node % (8 * sizeof(long)));
            (void)syscall(SYS_set_mempolicy, 1, mask, (unsigned long)(8 * sizeof mask));
        }
        const hipError_t he = timed_host_malloc((void **)&ctx->ingest_pin, need);
        if (bind_node >= 0) (void)syscall(SYS_set_mempolicy, 0, nullptr, 0ul);
        if (he != hipSuccess) { (void)hipGetLastError(); return PDT_ERR_NOMEM; }
        ctx->ingest_pin_cap = need;
    }
    if (!ctx->copy_stream) HIP_TRY(make_copy_stream(&ctx->copy_stream));
    if (!ctx->ev_ingest) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_ingest, hipEventDisableTiming));
    for (int q = 1; q < NS; q++) {
        if (!ctx->copy_streams_more[q - 1]) HIP_TRY(make_copy_stream(&ctx->copy_streams_more[q - 1]));
        if (!ctx->ev_ingest_more[q - 1]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_ingest_more[q - 1], hipEventDisableTiming));
    }
    hipStream_t cs[4] = { ctx->copy_stream, nullptr, nullptr, nullptr };
    for (int q = 1; q < NS; q++) cs[q] = ctx->copy_streams_more[q - 1];
    while (ctx->ingest_ev.size() < (size_t)nslots) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->ingest_ev.push_back(e);
    }
    // the destination may still be read by work queued earlier on the demodulation stream
    HIP_TRY(hipEventRecord(ctx->ev_ingest, ctx->stream));
    for (int q = 0; q < NS; q++) HIP_TRY(hipStreamWaitEvent(cs[q], ctx->ev_ingest, 0));
    if (ctx->tune.debug_overlap)
        fprintf(stderr, "ingest_capture: streams / events / pinned staging ready after %.2f ms\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_setup0).count());
    IngestJob local;
    IngestJob *J = job ? job : &local;
    for (int q = 0; q < NS; q++) J->cs[q] = cs[q];
    J->ncs = NS;
    J->submitted.reset(new std::atomic<int>[nspans]);
    for (size_t k = 0; k < nspans; k++) J->submitted[k].store(0);
    J->slot_state.reset(new std::atomic<int>[(size_t)nslots]);
    for (int q = 0; q < nslots; q++) J->slot_state[(size_t)q].store(0);
    J->slot_span.reset(new size_t[(size_t)nslots]);
    J->slot_delta.reset(new size_t[(size_t)nslots]);
    for (int q = 0; q < nslots; q++) J->slot_delta[(size_t)q] = 0;
    if (J->fd_direct >= 0) close(J->fd_direct);
    J->fd_direct = direct_guard.fd;
    direct_guard.fd = -1;
    J->nspans = nspans;
    J->span = PDT_INGEST_SPAN;
    J->waited = 0;
    J->mark_spans.clear();
    for (size_t b : J->mark_bytes) J->mark_spans.push_back(std::min(nspans, (b + PDT_INGEST_SPAN - 1) / PDT_INGEST_SPAN));
    J->mark_ready.reset(new std::atomic<int>[J->mark_spans.size() + 1]);
    for (size_t m = 0; m < J->mark_spans.size(); m++) J->mark_ready[m].store(0);
    while (ctx->span_ev.size() < J->mark_spans.size() * (size_t)NS) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->span_ev.push_back(e);
    }
    std::atomic<int> &failed = J->failed;
    // span k is read by thread k % T into its slot (k / T) % SLOTS: slot index (k % T) * SLOTS + (k / T) % SLOTS
    // (the background form copies what the lambdas need: they outlive this call)
    const IngestSrc src_c = src;
    unsigned char *pin_base = (unsigned char *)ctx->ingest_pin;
    auto reader = [J, src_c, bytes, nspans, T, PDT_INGEST_SPAN, pin_base, slot_stride, have_cpus, node_set](int t) {
        std::atomic<int> &failed = J->failed;
        const IngestSrc &src = src_c;
        if (have_cpus) (void)pthread_setaffinity_np(pthread_self(), sizeof node_set, &node_set);      // the GPU's socket (see probe_file)
        const int fd_direct = J->fd_direct;
        int round = 0;
        for (size_t k = (size_t)t; k < nspans && !failed; k += (size_t)T, round++) {
            const int slot = t * PDT_INGEST_SLOTS + (round % PDT_INGEST_SLOTS);
            std::atomic<int> &st = J->slot_state[(size_t)slot];
            while (st.load(std::memory_order_acquire) != 0) {          // its previous copy is still on its way
                if (failed) return;
                std::this_thread::sleep_for(std::chrono::microseconds(10));
            }
            unsigned char *pin = pin_base + (size_t)slot * slot_stride;
            const size_t at = k * PDT_INGEST_SPAN;
            const size_t len = std::min(PDT_INGEST_SPAN, bytes - at);
            size_t delta = 0;
            if (src.mem) {
                memcpy(pin, src.mem + at, len);
            } else if (fd_direct >= 0) {
                // from the device straight into the pinned slot: the read starts on the 4 KiB boundary below the span's first
                // byte and ends on the one above its last (the slot has the room), the copy to the GPU starts `delta` bytes in
                const uint64_t fo = src.off + at, a0 = fo & ~(uint64_t)4095;
                delta = (size_t)(fo - a0);
                const size_t want = (delta + len + 4095) & ~(size_t)4095;
                size_t got = 0;
                while (got < delta + len) {
                    const ssize_t r = pread(fd_direct, pin + got, want - got, (off_t)(a0 + got));
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) { failed = r == 0 ? 2 : 3; return; }
                    got += (size_t)r;
                }
            } else {
                size_t got = 0;
                while (got < len) {
                    const ssize_t r = pread(src.fd, pin + got, len - got, (off_t)(src.off + at + got));
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) { failed = r == 0 ? 2 : 3; return; }      // 2: the file ends early; 3: read error
                    got += (size_t)r;
                }
            }
            J->slot_span[(size_t)slot] = k;
            J->slot_delta[(size_t)slot] = delta;
            st.store(1, std::memory_order_release);
        }
    };
    const bool background = job != nullptr;
    auto submitter = [ctx, J, bytes, dst, nspans, NS, nslots, PDT_INGEST_SPAN, pin_base, background, slot_stride]() {
        std::atomic<int> &failed = J->failed;
        std::lock_guard<std::mutex> link(g_link_mu[(unsigned)ctx->cfg.device % 64u]);      // (the readers fill their first slots meanwhile)
        if (hipSetDevice(ctx->cfg.device) != hipSuccess) { failed = 1; return; }
        // Spans are queued as their slots fill, in whatever order that is: a reader that is late (descheduled: the hosts are
        // shared) holds up its own two slots only, not the ring (queued strictly in span order, one late reader stopped all
        // copies).  `upto` = every span below it has been queued: what the prefixes' boundary events go by.  Per copy stream the
        // slots in flight form a queue, oldest first.
        std::vector<int> inflight[4];
        size_t head[4] = { 0, 0, 0, 0 };
        size_t upto = 0, queued = 0;
        auto retire = [&](bool block) {
            for (int q = 0; q < NS; q++)
                while (head[q] < inflight[q].size()) {
                    const int slot = inflight[q][head[q]];
                    const hipError_t e = block ? hipEventSynchronize(ctx->ingest_ev[(size_t)slot]) : hipEventQuery(ctx->ingest_ev[(size_t)slot]);
                    if (e == hipErrorNotReady) { (void)hipGetLastError(); break; }
                    if (e != hipSuccess) { failed = 1; return; }
                    J->slot_state[(size_t)slot].store(0, std::memory_order_release);
                    head[q]++;
                }
        };
        size_t mi = 0;
        auto marks = [&]() {                                           // prefixes complete with the spans queued so far
            while (upto < nspans && J->submitted[upto].load(std::memory_order_relaxed)) upto++;
            while (mi < J->mark_spans.size() && J->mark_spans[mi] <= upto) {
                for (int q = 0; q < NS; q++)
                    if (hipEventRecord(ctx->span_ev[mi * (size_t)NS + (size_t)q], J->cs[q]) != hipSuccess) { failed = 1; return; }
                J->mark_ready[mi].store(1, std::memory_order_release);
                mi++;
            }
        };
        int rr = 0;                                                    // copy streams round robin
        // (developer aid, PDT_DEBUG_OVERLAP: how long the submitter had nothing to queue, and how many copies were in flight on
        // average while it waited -- near the ring's half: the copies are the pace (a reader refills a slot as soon as its copy has
        // completed); near zero: the readers are)
        double idle_ms = 0, flying_sum = 0, api_ms = 0;
        long long idle_n = 0;
        const auto t_sub0 = std::chrono::steady_clock::now();
        while (queued < nspans && !failed) {
            bool any = false;
            const auto t_it0 = std::chrono::steady_clock::now();
            for (int slot = 0; slot < nslots && !failed; slot++) {
                if (J->slot_state[(size_t)slot].load(std::memory_order_acquire) != 1) continue;
                const size_t k = J->slot_span[(size_t)slot];
                const size_t at = k * PDT_INGEST_SPAN;
                const size_t len = std::min(PDT_INGEST_SPAN, bytes - at);
                const int q = rr;
                rr = (rr + 1 == NS) ? 0 : rr + 1;
                if (hipMemcpyAsync((unsigned char *)dst + at, pin_base + (size_t)slot * slot_stride + J->slot_delta[(size_t)slot], len, hipMemcpyHostToDevice, J->cs[q]) != hipSuccess ||
                    hipEventRecord(ctx->ingest_ev[(size_t)slot], J->cs[q]) != hipSuccess) {
                    failed = 1;
                    return;
                }
                J->slot_state[(size_t)slot].store(2, std::memory_order_release);
                inflight[q].push_back(slot);
                J->submitted[k].store(1, std::memory_order_release);
                queued++;
                any = true;
                marks();
            }
            retire(false);
            const auto t_it1 = std::chrono::steady_clock::now();
            api_ms += std::chrono::duration<double, std::milli>(t_it1 - t_it0).count();
            if (!any) {
                int flying = 0;
                for (int slot = 0; slot < nslots; slot++) flying += J->slot_state[(size_t)slot].load(std::memory_order_relaxed) == 2;
                std::this_thread::sleep_for(std::chrono::microseconds(15));
                idle_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_it1).count();
                flying_sum += flying;
                idle_n++;
            }
        }
        marks();
        if (ctx->tune.debug_overlap)
            fprintf(stderr, "ingest submitter: %.2f ms in all: %.2f in the runtime (queueing, polling), %.2f with nothing to queue (%.1f of %d slots in flight on average)\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_sub0).count(), api_ms, idle_ms,
                    idle_n ? flying_sum / (double)idle_n : 0.0, nslots);
        (void)nslots;
        // the background form leaves the pinned slots free; the foreground form returns with its last copies still on their way
        // (the caller's stream waits for them, and the call does not return before that stream is idle)
        if (background) retire(true);
    };
    bool spawn_failed = false;
    try {
        for (int t = 0; t < T; t++) J->pool.emplace_back(reader, t);
        if (job) J->pool.emplace_back(submitter);
    } catch (const std::exception &) {
        failed = 1;
        spawn_failed = true;
    }
    // (background form: the readers are already running and may have set `failed` -- a short file, a read error: those codes
    // come through ingest_wait_mark / ingest_join, always; only a thread that could not be started is reported here)
    if (job) return spawn_failed ? PDT_ERR_NOMEM : PDT_OK;             // (the caller joins: ingest_join)
    if (!failed) submitter();
    for (auto &th : J->pool) th.join();
    J->pool.clear();
    if (failed == 2) return PDT_ERR_FORMAT;                // the file is shorter than announced
    if (failed == 3) return PDT_ERR_IO;
    if (failed) { (void)hipGetLastError(); return PDT_ERR_NOGPU; }
    HIP_TRY(hipEventRecord(ctx->ev_ingest, ctx->copy_stream));
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_ingest, 0));
    for (int q = 1; q < NS; q++) {
        HIP_TRY(hipEventRecord(ctx->ev_ingest_more[q - 1], cs[q]));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_ingest_more[q - 1], 0));
    }
    return PDT_OK;
}

// make `stream` wait for the copies of prefix `m` of the job (IngestJob::mark_bytes).  The host waits -- without touching the HIP
// runtime: the submitter is in it all the time -- until the submitter has queued the prefix's last span and recorded one event
// per copy stream behind it; the stream then waits for those events: for the prefix's copies and for nothing that was queued
// later.  (Round 3 recorded an event per span and made the stream wait for each: ~300 calls, and the host's loop lagged 10 ms
// behind the data.)
int ingest_wait_mark(pdt_ctx *ctx, IngestJob &job, size_t m, hipStream_t stream)
{
    if (m >= job.mark_spans.size()) return PDT_ERR_STATE;
    while (!job.mark_ready[m].load(std::memory_order_acquire)) {
        if (job.failed) return job.failed == 2 ? PDT_ERR_FORMAT : job.failed == 3 ? PDT_ERR_IO : PDT_ERR_NOGPU;
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    for (int q = 0; q < job.ncs; q++) HIP_TRY(hipStreamWaitEvent(stream, ctx->span_ev[m * (size_t)job.ncs + (size_t)q], 0));
    return PDT_OK;
}

int ingest_join(IngestJob &job)
{
    for (auto &th : job.pool) th.join();
    job.pool.clear();
    if (job.failed == 2) return PDT_ERR_FORMAT;
    if (job.failed == 3) return PDT_ERR_IO;
    if (job.failed) { (void)hipGetLastError(); return PDT_ERR_NOGPU; }
    return PDT_OK;
}

}  // namespace

// ================================================================================= C ABI
extern "C" {

int pdt_abi_version(void) { return PDT_ABI_VERSION; }
#ifndef PDT_BUILD_TAG
#define PDT_BUILD_TAG "untagged"
#endif
const char *pdt_build_tag(void) { return PDT_BUILD_TAG; }

// test-only (include/pdt_dev.h): set (value != NULL), remove (value == NULL) or clear (name == NULL) developer switches
int pdt_dev_set(const char *name, const char *value)
{
    try {
        std::lock_guard<std::mutex> lock(g_dev_mu);
        if (!name) g_dev.clear();
        else if (!value) g_dev.erase(name);
        else g_dev[name] = value;
    } catch (const std::exception &) { return PDT_ERR_NOMEM; }
    return PDT_OK;
}

unsigned long long pdt_dev_span_rows(const pdt_ctx *ctx, unsigned *n_exits_out, unsigned long long max)
{
    if (!ctx || ctx->gspan_nrows <= 0 || !ctx->gspan_rows.p) return 0;
    const unsigned long long n = (unsigned long long)ctx->gspan_nrows, m = std::min(n, max);
    if (n_exits_out && m) {
        std::vector<GardnerSpanRow> rows((size_t)m);
        if (hipSetDevice(ctx->cfg.device) != hipSuccess ||
            hipMemcpy(rows.data(), ctx->gspan_rows.p, (size_t)m * sizeof(GardnerSpanRow), hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        for (size_t k = 0; k < (size_t)m; k++) n_exits_out[k] = rows[k].n;
    }
    return n;
}

const char *pdt_strerror(int code)
{
    switch (code) {
    case PDT_OK: return "ok";
    case PDT_ERR_ARG: return "bad argument";
    case PDT_ERR_NOGPU: return "no usable HIP device / HIP runtime error";
    case PDT_ERR_NOMEM: return "out of device memory";
    case PDT_ERR_FORMAT: return "unsupported WAV format (need 16-bit PCM, 2 channels)";
    case PDT_ERR_RATE: return "sample rate too high: interpolation factor would be 0";
    case PDT_ERR_STATE: return "call sequence / internal capacity error";
    case PDT_ERR_IO: return "read error on the capture file";
    default: return "unknown error";
    }
}

int pdt_get_device(const pdt_ctx *ctx) { return ctx ? ctx->cfg.device : -1; }

int pdt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// Host evaluation of the library's own restatements of the C-library functions the reference calls (test hook: the CPU
// tests compare them with the C library of the machine, bit for bit).  fn: 0 sincos(x) -> out0 sin, out1 cos; 1 sin; 2 cos;
// 3 sincosf((float)x) widened; 4 hypot(x[2i], x[2i+1]) -> out0[i]; 5 hypotf of the pair, widened; 6 the branch-free form
// of 3 (sincosf_flat: what the fused mix + FIR kernel evaluates); 7 / 8 the error and phase wraps of one float PLL step
// (pll_wrap_error_f32 / pll_wrap_phase_f32), widened.
int pdt_host_math(int fn, const double *x, uint64_t n, double *out0, double *out1)
{
    if (!x || !out0) return PDT_ERR_ARG;
    for (uint64_t i = 0; i < n; i++) {
        switch (fn) {
        case 0: { double sv, cv; sincos_glibc(x[i], sv, cv); out0[i] = sv; if (out1) out1[i] = cv; break; }
        case 1: out0[i] = sin_glibc(x[i]); break;
        case 2: out0[i] = cos_glibc(x[i]); break;
        case 3: { float sf, cf; sincosf_glibc((float)x[i], sf, cf); out0[i] = sf; if (out1) out1[i] = cf; break; }
        case 4: out0[i] = hypot_glibc(x[2 * i], x[2 * i + 1]); break;
        case 5: out0[i] = hypotf_glibc((float)x[2 * i], (float)x[2 * i + 1]); break;
        case 6: { float sf, cf; sincosf_flat((float)x[i], sf, cf); out0[i] = sf; if (out1) out1[i] = cf; break; }
        case 7: out0[i] = pll_wrap_error_f32((float)x[i]); break;
        case 8: out0[i] = pll_wrap_phase_f32((float)x[i]); break;
        default: return PDT_ERR_ARG;
        }
    }
    return PDT_OK;
}

int pdt_make_lpf(int mode, uint32_t sample_rate, void *taps_out, int *ntaps, int *interp)
{
    if (!sample_rate) return PDT_ERR_ARG;
    if (mode == PDT_MODE_ARGOS) {
        if (ntaps) *ntaps = 50;
        if (interp) *interp = 1;
        if (taps_out) make_lpf<double>((double *)taps_out, 50, 700.0, (double)sample_rate, 1);   // ARGOSdemod/main.c:248
        return PDT_OK;
    }
    const int ip = poes_interp(sample_rate);
    if (ip < 1) return PDT_ERR_RATE;
    const int N = 26 * ip;
    if (ntaps) *ntaps = N;
    if (interp) *interp = ip;
    if (taps_out) {
        const float Fs = (float)sample_rate;
        make_lpf<float>((float *)taps_out, N, (float)11000.0, Fs * (float)ip, ip);              // POESTIPdemod/main.c:369
    }
    return PDT_OK;
}

int pdt_wav_parse_header(const uint8_t h[44], uint32_t *sample_rate, uint32_t *channels, uint32_t *bits_per_sample,
                         uint32_t *format, uint32_t *data_bytes)
{
    if (!h) return PDT_ERR_ARG;
    auto r32 = [&](int o) { return (uint32_t)h[o] | ((uint32_t)h[o + 1] << 8) | ((uint32_t)h[o + 2] << 16) | ((uint32_t)h[o + 3] << 24); };
    auto r16 = [&](int o) { return (uint32_t)h[o] | ((uint32_t)h[o + 1] << 8); };
    if (format) *format = r16(20);
    if (channels) *channels = r16(22);
    if (sample_rate) *sample_rate = r32(24);
    if (bits_per_sample) *bits_per_sample = r16(34);
    if (data_bytes) *data_bytes = r32(40);
    return PDT_OK;
}

double pdt_time_axis(int mode, uint32_t sample_rate, uint64_t m)
{
    if (mode == PDT_MODE_ARGOS) {
        TimeAxis<double> ax;
        ax.init(1.0 / (double)sample_rate);
        return ax.at(m);
    }
    TimeAxis<float> ax;
    ax.init((float)(1.0 / (double)(float)sample_rate));
    return (double)ax.at(m);
}

int pdt_open(const pdt_config *cfg, pdt_ctx **out)
{
    if (!cfg || !out) return PDT_ERR_ARG;
    if (cfg->mode != PDT_MODE_POES && cfg->mode != PDT_MODE_ARGOS) return PDT_ERR_ARG;
    if (!cfg->sample_rate) return PDT_ERR_ARG;
    if (cfg->chain != PDT_CHAIN_FILE && cfg->chain != PDT_CHAIN_LIVE) return PDT_ERR_ARG;
    // (ARGOS + PDT_CHAIN_LIVE = the ARGOS sound-card twin: the FLOAT build of the ARGOS chain, ARGOSdemodPortAudio/config.h;
    // round 4)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        fprintf(stderr, "libpdt: no HIP device available -- this library has no CPU path\n");
        return PDT_ERR_NOGPU;
    }
    if (cfg->device < 0 || cfg->device >= ndev) return PDT_ERR_ARG;
    HIP_TRY(hipSetDevice(cfg->device));
    pdt_ctx *ctx = new pdt_ctx();
    ctx->cfg = *cfg;
    ctx->tune.load();
    if (ctx->tune.pll_block && !ctx->cfg.pll_block) ctx->cfg.pll_block = (uint32_t)ctx->tune.pll_block;
    if (ctx->tune.pll_warm_s > 0 && !ctx->cfg.pll_warm) ctx->cfg.pll_warm = (uint32_t)(ctx->tune.pll_warm_s * cfg->sample_rate);
    if (ctx->tune.agc_warm_s > 0 && !ctx->cfg.agc_warm) ctx->cfg.agc_warm = (uint32_t)(ctx->tune.agc_warm_s * cfg->sample_rate);
    if (!ctx->cfg.chunk) ctx->cfg.chunk = (cfg->mode == PDT_MODE_ARGOS || cfg->chain == PDT_CHAIN_LIVE) ? 2400 : 10000;
    const bool argos_twin = cfg->mode == PDT_MODE_ARGOS && cfg->chain == PDT_CHAIN_LIVE;
    ctx->elem = (cfg->mode == PDT_MODE_ARGOS && !argos_twin) ? 8 : 4;
    int nt = 0, ip = 0;
    int rc = pdt_make_lpf(cfg->mode, cfg->sample_rate, nullptr, &nt, &ip);
    if (rc) { pdt_close(ctx); return rc; }
    ctx->interp = (uint32_t)ip;
    ctx->ntaps = (uint32_t)nt;
    ctx->taps_host.resize((size_t)nt * ctx->elem);
    if (argos_twin)           // MakeLPFIR(filterCoeffs, 50, 700, Fs, 1) in float (ARGOSdemodPortAudio/main.c:264)
        make_lpf<float>((float *)ctx->taps_host.data(), 50, (float)700, (float)cfg->sample_rate, 1);
    else
    pdt_make_lpf(cfg->mode, cfg->sample_rate, ctx->taps_host.data(), nullptr, nullptr);
    if ((rc = ctx->taps.ensure(ctx->taps_host.size()))) { pdt_close(ctx); return rc; }
    if (hipMemcpy(ctx->taps.p, ctx->taps_host.data(), ctx->taps_host.size(), hipMemcpyHostToDevice) != hipSuccess) {
        pdt_close(ctx);
        return PDT_ERR_NOGPU;
    }
    if (cfg->mode == PDT_MODE_POES && nt == 26 * ip) {
        // tap table of the register-tiled FIR, rotated per ring residue: rot[c][t][r] = h[N-1-r-((c-t) mod K)*interp]
        const int K = 26, rs = ip;                  // rows of `interp` taps, one residue = K * interp consecutive floats
        const int cs = (K * rs + 15) & ~15;         // residue stride: every residue's block starts on a 64-byte boundary
        std::vector<float> rot((size_t)K * cs, 0.0f);
        const float *h = (const float *)ctx->taps_host.data();
        for (int c = 0; c < K; c++)
            for (int t = 0; t < K; t++)
                for (int r = 0; r < ip; r++) rot[(size_t)c * cs + (size_t)t * rs + r] = h[nt - 1 - r - ((c - t + K) % K) * ip];
        if ((rc = ctx->taps_rot.ensure(rot.size() * sizeof(float)))) { pdt_close(ctx); return rc; }
        if (hipMemcpy(ctx->taps_rot.p, rot.data(), rot.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            pdt_close(ctx);
            return PDT_ERR_NOGPU;
        }
    }
    // (the two streams side by side: ~10 ms each, 20 for a process's first -- tools/probes/cold_path_probe.hip)
    std::thread side_stream_task;
    auto make_side_stream = [ctx]() {
        // the side stream at another priority than the main one: streams of one priority share a few hardware queues round
        // robin, and two streams that land on the same queue run one after the other (seen in a batch trace: the block-parallel
        // PLL kernel and the acquisition it should run beside, serialised); another priority is another set of queues
        if (hipSetDevice(ctx->cfg.device) != hipSuccess) { (void)hipGetLastError(); return; }
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, greatest) != hipSuccess) {
            (void)hipGetLastError();
            if (hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ctx->stream2 = nullptr; }
        }
    };
    try { side_stream_task = std::thread(make_side_stream); } catch (const std::exception &) { make_side_stream(); }
    auto join_side = [&]() { if (side_stream_task.joinable()) side_stream_task.join(); };     // (before the context goes away)
    if (hipStreamCreate(&ctx->stream) != hipSuccess) { ctx->stream = nullptr; join_side(); pdt_close(ctx); return PDT_ERR_NOGPU; }
    ctx->own_stream = true;
    {
        void *small = nullptr;
        if (timed_host_malloc((void **)&small, sizeof(DevScalars) + 128 + 32768) != hipSuccess) { join_side(); pdt_close(ctx); return PDT_ERR_NOMEM; }
        ctx->pend_sc = (DevScalars *)small;
        ctx->pend_info = (unsigned char *)small + ((sizeof(DevScalars) + 15) & ~(size_t)15);
        ctx->seg_pin = ctx->pend_info + 128;          // 4 KiB up (kept bits, history symbols, lock record, gain), the rest down (SegTail)
    }
    (void)hipEventCreate(&ctx->ev0);
    (void)hipEventCreate(&ctx->ev1);
    (void)hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming);
    join_side();
    if (!ctx->stream2) { pdt_close(ctx); return PDT_ERR_NOGPU; }
    memset(&ctx->stats, 0, sizeof ctx->stats);
    memset(ctx->stage_len, 0, sizeof ctx->stage_len);
    ctx->axis_f.init((float)(1.0 / (double)(float)cfg->sample_rate));        // wave.c:96-97
    if (argos_twin) ctx->axis_f.init(1 / (float)cfg->sample_rate);           // "Time += (1/Fs)", all float (ARGOSdemodPortAudio/main.c:285)
    ctx->axis_d.init(1.0 / (double)cfg->sample_rate);
    ctx->counted = true;
    g_open_contexts.fetch_add(1);
    *out = ctx;
    return PDT_OK;
}

void pdt_close(pdt_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    DevBuf *bufs[] = { &ctx->pcm, &ctx->pll, &ctx->lock, &ctx->fir, &ctx->agc, &ctx->sym, &ctx->symidx, &ctx->bits, &ctx->bitsym,
                       &ctx->hits, &ctx->frames, &ctx->taps, &ctx->mag, &ctx->seams_pll, &ctx->seams_agc, &ctx->scal, &ctx->lockinfo,
                       &ctx->term, &ctx->seams_ema, &ctx->gtable, &ctx->gentries, &ctx->gcand,
                       &ctx->gmfirst, &ctx->stiles, &ctx->gsegmap, &ctx->gsegstart, &ctx->gbands, &ctx->gclist, &ctx->gspan_keys, &ctx->gspan_tails, &ctx->gspan_rows, &ctx->gspan_items, &ctx->gspan_ctl, &ctx->gspan_recs, &ctx->gcentries, &ctx->gflags, &ctx->agc_maps, &ctx->pll_head, &ctx->taps_rot, &ctx->pll_scratch, &ctx->tip, &ctx->stream_in, &ctx->sync_scr, &ctx->agc_raw, &ctx->agc_ckpt, &ctx->pll_ckpt, &ctx->packs_dev, &ctx->seg_dev, &ctx->lt_theta, &ctx->lt_phi,
                       &ctx->avgph, &ctx->term_ap, &ctx->seams_q, &ctx->chunkinfo };
    for (DevBuf *b : bufs) b->release();
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->qual_pin) (void)hipHostFree(ctx->qual_pin);
    if (ctx->packs_pin) (void)hipHostFree(ctx->packs_pin);
    if (ctx->ingest_pin) (void)hipHostFree(ctx->ingest_pin);
    for (hipEvent_t e : ctx->ingest_ev) (void)hipEventDestroy(e);
    if (ctx->ev_ingest) (void)hipEventDestroy(ctx->ev_ingest);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (hipEvent_t e : ctx->span_ev) (void)hipEventDestroy(e);
    for (int q = 0; q < 3; q++) {
        if (ctx->ev_ingest_more[q]) (void)hipEventDestroy(ctx->ev_ingest_more[q]);
        if (ctx->copy_streams_more[q]) (void)hipStreamDestroy(ctx->copy_streams_more[q]);
    }
    if (ctx->pend_sc) (void)hipHostFree(ctx->pend_sc);
    for (auto &t : ctx->timers) { if (!t.shared_a) (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->counted) g_open_contexts.fetch_sub(1);
    delete ctx;
}

int pdt_set_stream(pdt_ctx *ctx, void *hip_stream)
{
    if (!ctx) return PDT_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->own_stream = false;
    ctx->stream = (hipStream_t)hip_stream;
    if (!hip_stream) {
        if (hipStreamCreate(&ctx->stream) != hipSuccess) return PDT_ERR_NOGPU;
        ctx->own_stream = true;
    }
    return PDT_OK;
}

int pdt_set_loop_params(pdt_ctx *ctx, const pdt_loop_params *p)
{
    if (!ctx || !p) return PDT_ERR_ARG;
    if (ctx->stream_open) return PDT_ERR_STATE;
    const double v[] = { p->pll_freq_range_hz, p->pll_lock_threshold, p->pll_lock_alpha, p->pll_loopbw_acq, p->pll_loopbw_track, p->agc_attack,
                         p->agc_decay, p->gardner_baud, p->gardner_step_range, p->gardner_kp, p->manchester_threshold };
    for (double x : v)
        if (!(x >= 0.0) || !std::isfinite(x)) return PDT_ERR_ARG;
    if (p->gardner_baud != 0 && (double)ctx->cfg.sample_rate * (double)ctx->interp / p->gardner_baud < 2.0) return PDT_ERR_ARG;
    if (p->pll_freq_range_hz != 0 && p->pll_freq_range_hz >= 0.5 * (double)ctx->cfg.sample_rate) return PDT_ERR_ARG;
    if ((float)p->gardner_step_range > 0.1f) return PDT_ERR_ARG;      // (the samplers' window margins and symbol capacities are sized for the mains' 0.1)
    if (p->zero_mask & ~(uint32_t)(PDT_LP_ZERO_LOCK_THRESHOLD | PDT_LP_ZERO_GARDNER_KP | PDT_LP_ZERO_GARDNER_STEP_RANGE | PDT_LP_ZERO_MANCHESTER_THRESHOLD))
        return PDT_ERR_ARG;
    // a baud rate whose step would not fit the sequential samplers' LDS windows (a symbol and its mid-point must lie inside
    // one; the casts of the step to integers further down stay in range with it): at most 4 096 samples per symbol
    if (p->gardner_baud != 0 && !((double)ctx->cfg.sample_rate * (double)ctx->interp / p->gardner_baud <= 4096.0)) return PDT_ERR_ARG;
    ctx->lp = *p;
    ctx->gcand_key = -1;                    // (the sampler's candidate list depends on the step)
    return PDT_OK;
}

int pdt_keep_quality(pdt_ctx *ctx, int enable)
{
    if (!ctx) return PDT_ERR_ARG;
    ctx->keep_quality = enable != 0;
    if (!enable) ctx->chunk_host.clear();
    return PDT_OK;
}

uint64_t pdt_chunk_reports(const pdt_ctx *ctx, pdt_chunk_report *out, uint64_t max_chunks)
{
    if (!ctx) return 0;
    const uint64_t nc = ctx->chunk_host.size();
    if (!out) return nc;
    const uint64_t m = std::min<uint64_t>(nc, max_chunks);
    chunk_reports_range(ctx, 0, m, ctx->report_samples, out, nullptr);
    return m;
}

int pdt_set_progress(pdt_ctx *ctx, pdt_progress_fn fn, void *user)
{
    if (!ctx) return PDT_ERR_ARG;
    ctx->progress_fn = fn;
    ctx->progress_user = user;
    return PDT_OK;
}

int pdt_keep_pll(pdt_ctx *ctx, int enable)
{
    if (!ctx) return PDT_ERR_ARG;
    ctx->keep_pll = enable != 0;
    ctx->keep_pll_asked = enable != 0;
    return PDT_OK;
}

int pdt_keep_presquelch(pdt_ctx *ctx, int enable)
{
    if (!ctx) return PDT_ERR_ARG;
    ctx->keep_agc_raw = enable != 0;
    if (!enable) ctx->stage_len[PDT_ST_AGC_RAW] = 0;
    return PDT_OK;
}

static int demod_common(pdt_ctx *ctx, uint64_t nframes, int phase = RUN_ALL)
{
    if (phase == RUN_ALL) ctx->batch_hint = 1;
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    ctx->n_samples = nframes;
    ctx->n_out = nframes * ctx->interp;
    if (ctx->elem == 8) return run_capture<double>(ctx, nframes, phase);
    return run_capture<float>(ctx, nframes, phase);                 // POES, both twins
}

static int demod_overlapped(pdt_ctx *ctx, const IngestSrc &src, uint64_t nframes, int fmt, int text_fd, uint64_t *text_bytes);
static int demod_windowed(pdt_ctx *ctx, const IngestSrc &src, uint64_t nframes, int fmt, int text_fd, uint64_t *text_bytes, uint64_t piece);
static uint64_t stream_history(const pdt_ctx *ctx);

// Does a capture of nframes fit the device in one piece?  The reference's chunk loop takes a file of any length in O(chunk)
// memory (POESTIPdemod/main.c:373, while(!feof)); the one-piece path here keeps every stage's stream of the whole capture
// resident -- about 8 x the file for POES at 250 ksps.  Returns 0 when that fits what the device has free (plus
// what this context already holds), otherwise the number of samples per piece of the bounded window (demod_windowed), or a
// negative error code when not even a window of a few chunks fits.
static long long window_piece_for(pdt_ctx *ctx, uint64_t nframes, size_t fb)
{
    const bool need_lock = ctx->cfg.mode == PDT_MODE_ARGOS || ctx->cfg.chain == PDT_CHAIN_LIVE;
    const double es = (double)ctx->elem, ip = (double)ctx->interp;
    // bytes per input sample: input, PLL output, theta / phases (their own buffers when the capture is taken in segments), filter
    // and AGC output, lock signal + its input term, the quality EMA's two streams, symbols / bits / tables (a few per cent)
    const double per = (double)fb + es * (3.0 + 2.5 * ip) + 8.0 + (need_lock ? 2.0 * es : 0.0) + (ctx->keep_quality ? 2.0 * es : 0.0) +
                       (ctx->keep_agc_raw ? es * ip : 0.0);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return 0; }
    // what the context already holds is its to re-use
    const DevBuf *held[] = { &ctx->pcm, &ctx->pll, &ctx->lock, &ctx->fir, &ctx->agc, &ctx->term, &ctx->stream_in, &ctx->lt_theta, &ctx->lt_phi,
                             &ctx->avgph, &ctx->term_ap, &ctx->agc_raw, &ctx->gtable, &ctx->sym, &ctx->symidx };
    double avail = (double)free_b;
    for (const DevBuf *b : held) avail += (double)b->cap;
    if (ctx->tune.hbm_limit_mb > 0) avail = (double)ctx->tune.hbm_limit_mb * 1048576.0;
    const double need = per * (double)nframes + 256.0 * 1048576.0;
    if (need <= 0.92 * avail && !ctx->tune.window_piece) return 0;
    // the window: its history (PLL warm-ups) + one piece, with the growth margins of the stream's buffers
    const double per_win = 1.3 * per + 2.0 * es;
    const uint64_t chunk = ctx->cfg.chunk, hist = stream_history(ctx) + 2 * chunk;
    double piece = 0.5 * avail / per_win - (double)hist;
    piece = std::min(piece, 64.0 * 1048576.0);                        // (a piece is a few ms of GPU time: no need for more)
    if (ctx->tune.window_piece) piece = (double)ctx->tune.window_piece;
    if (!(piece >= 16.0 * (double)chunk)) return PDT_ERR_NOMEM;
    return (long long)((uint64_t)piece / chunk * chunk);
}

int pdt_demod_pcm16(pdt_ctx *ctx, const int16_t *iq_host, uint64_t nframes)
{
    if (!ctx || (!iq_host && nframes)) return PDT_ERR_ARG;
    if (ctx->stream_open) return PDT_ERR_STATE;                      // pdt_stream_end / pdt_stream_begin first
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    IngestSrc src;
    src.mem = (const unsigned char *)iq_host;
    if (const long long piece = window_piece_for(ctx, nframes, 4)) return piece < 0 ? (int)piece : demod_windowed(ctx, src, nframes, 0, -1, nullptr, (uint64_t)piece);
    int rc = ctx->pcm.ensure((size_t)nframes * 4 + 16);
    if (rc) return rc;
    const auto t_in = std::chrono::steady_clock::now();
    if ((rc = ingest_capture(ctx, src, (size_t)nframes * 4, ctx->pcm.p))) return rc;
    ctx->ingest_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count();
    ctx->pcm_dev = ctx->pcm.p;
    ctx->pcm_fmt = 0;
    rc = demod_common(ctx, nframes);
    ctx->stats.ingest_ms = ctx->ingest_ms;
    ctx->stats.ingest_direct = (uint32_t)ctx->ingest_was_direct;
    ctx->stats.ingest_numa_node = ctx->ingest_numa_node;
    return rc;
}

// hour-long POES captures: the chain starts on the part of the capture that has arrived (demod_overlapped)
static bool overlap_ingest(const pdt_ctx *ctx, uint64_t nframes, size_t fb)
{
    // Round 5: on by default (PDT_NO_OVERLAP switches it off) -- the segments are cut where they can take the whole-capture
    // kernels and keep the per-chunk reports (demod_overlapped); 3.6 GB file to frame file 82-91 ms against 89-94 ms with the
    // capture ingested first (profiles/r5).  From 2.5 GiB on: every segment pays the block-parallel stages' latency floors again
    // (~7 ms), hidden only while the next segment's samples take longer than that to arrive -- 0.6 GB: 30.8 ms overlapped
    // against 21.3 ms, 1.2 GB: 40.7 / 37.3, 2.4 GB: 63.5 / 63.5, 3.0 GB: 75.2 / 76.5 (profiles/r5/e2e_ab_overlap_by_size.txt).
    if (ctx->keep_pll_asked) return false;           // the caller asked for the stage arrays of the whole capture
    return ctx->tune.overlap && ctx->cfg.mode == PDT_MODE_POES && ctx->cfg.sampler != PDT_SAMPLER_MM && ctx->cfg.chain != PDT_CHAIN_LIVE &&
           (size_t)nframes * fb >= ((size_t)(ctx->tune.overlap_min_mb > 0 ? ctx->tune.overlap_min_mb : 2560) << 20) && ctx->cfg.chunk > 0 &&
           nframes / ctx->cfg.chunk >= 64;
}

int pdt_demod_fd(pdt_ctx *ctx, int fd, uint64_t byte_offset, uint64_t nframes, int sample_format)
{
    if (!ctx || fd < 0 || (sample_format != PDT_FMT_PCM16 && sample_format != PDT_FMT_F32)) return PDT_ERR_ARG;
    if (sample_format == PDT_FMT_F32 && ctx->elem != 4) return PDT_ERR_FORMAT;   // ARGOSdemod/main.c:238-241
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    const size_t fb = sample_format == PDT_FMT_F32 ? 8 : 4;
    IngestSrc src;
    src.fd = fd;
    src.off = byte_offset;
    if (ctx->stream_open) return PDT_ERR_STATE;
    if (const long long piece = window_piece_for(ctx, nframes, fb))
        return piece < 0 ? (int)piece : demod_windowed(ctx, src, nframes, sample_format == PDT_FMT_F32 ? 1 : 0, -1, nullptr, (uint64_t)piece);
    if (overlap_ingest(ctx, nframes, fb)) return demod_overlapped(ctx, src, nframes, sample_format == PDT_FMT_F32 ? 1 : 0, -1, nullptr);
    int rc = ctx->pcm.ensure((size_t)nframes * fb + 16);
    if (rc) return rc;
    const auto t_in = std::chrono::steady_clock::now();
    if ((rc = ingest_capture(ctx, src, (size_t)nframes * fb, ctx->pcm.p))) return rc;
    ctx->ingest_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count();
    ctx->pcm_dev = ctx->pcm.p;
    ctx->pcm_fmt = sample_format == PDT_FMT_F32 ? 1 : 0;
    rc = demod_common(ctx, nframes);
    ctx->stats.ingest_ms = ctx->ingest_ms;
    ctx->stats.ingest_direct = (uint32_t)ctx->ingest_was_direct;
    ctx->stats.ingest_numa_node = ctx->ingest_numa_node;
    return rc;
}

int pdt_demod_device(pdt_ctx *ctx, const void *iq_device, uint64_t nframes)
{
    if (!ctx || (!iq_device && nframes)) return PDT_ERR_ARG;
    if (ctx->stream_open) return PDT_ERR_STATE;                      // pdt_stream_end / pdt_stream_begin first
    ctx->pcm_dev = iq_device;
    ctx->pcm_fmt = 0;
    return demod_common(ctx, nframes);
}

int pdt_demod_f32(pdt_ctx *ctx, const float *iq_host, uint64_t nframes)
{
    if (!ctx || (!iq_host && nframes)) return PDT_ERR_ARG;
    if (ctx->stream_open) return PDT_ERR_STATE;                      // pdt_stream_end / pdt_stream_begin first
    if (ctx->elem != 4) return PDT_ERR_FORMAT;       // ARGOSdemod/main.c:238-241: "RAW files not yet supported"
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    IngestSrc src;
    src.mem = (const unsigned char *)iq_host;
    if (const long long piece = window_piece_for(ctx, nframes, 8)) return piece < 0 ? (int)piece : demod_windowed(ctx, src, nframes, 1, -1, nullptr, (uint64_t)piece);
    int rc = ctx->pcm.ensure((size_t)nframes * 8 + 16);
    if (rc) return rc;
    const auto t_in = std::chrono::steady_clock::now();
    if ((rc = ingest_capture(ctx, src, (size_t)nframes * 8, ctx->pcm.p))) return rc;
    ctx->ingest_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count();
    ctx->pcm_dev = ctx->pcm.p;
    ctx->pcm_fmt = 1;
    rc = demod_common(ctx, nframes);
    ctx->stats.ingest_ms = ctx->ingest_ms;
    ctx->stats.ingest_direct = (uint32_t)ctx->ingest_was_direct;
    ctx->stats.ingest_numa_node = ctx->ingest_numa_node;
    return rc;
}

int pdt_demod_device_f32(pdt_ctx *ctx, const void *iq_device, uint64_t nframes)
{
    if (!ctx || (!iq_device && nframes)) return PDT_ERR_ARG;
    if (ctx->stream_open) return PDT_ERR_STATE;                      // pdt_stream_end / pdt_stream_begin first
    if (ctx->elem != 4) return PDT_ERR_FORMAT;
    ctx->pcm_dev = iq_device;
    ctx->pcm_fmt = 1;
    return demod_common(ctx, nframes);
}

int pdt_stage_bytesync(pdt_ctx *ctx, const uint8_t *bits_host, uint64_t nbits)
{
    return pdt_stage_bytesync_from(ctx, bits_host, nbits, 0);
}

int pdt_stage_bytesync_from(pdt_ctx *ctx, const uint8_t *bits_host, uint64_t nbits, uint64_t first_sync_end)
{
    if (!ctx || (!bits_host && nbits)) return PDT_ERR_ARG;
    if (nbits >= (1ull << 31)) return PDT_ERR_ARG;
    if (ctx->stream_open) return PDT_ERR_STATE;              // (the stage buffers carry an open stream's tails)
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    const bool argos = ctx->cfg.mode == PDT_MODE_ARGOS;
    const SyncParams SP = make_sync_params(argos);
    const long long bit_cap = (long long)nbits + 64;
    const uint32_t hit_cap = next_pow2((uint32_t)(bit_cap / 4 + 4096));
    const uint32_t frame_cap = (uint32_t)(bit_cap / SP.span + 16);
    int rc;
    if ((rc = ctx->bits.ensure((size_t)bit_cap))) return rc;
    if ((rc = ctx->bitsym.ensure((size_t)bit_cap * sizeof(unsigned)))) return rc;
    if ((rc = ctx->symidx.ensure((size_t)bit_cap * sizeof(long long)))) return rc;
    if ((rc = ctx->hits.ensure((size_t)hit_cap * sizeof(unsigned)))) return rc;
    if ((rc = ctx->sync_scr.ensure((2 * ((size_t)hit_cap + 1) + hit_cap / 32 + 2) * sizeof(unsigned)))) return rc;
    if ((rc = ctx->frames.ensure((size_t)frame_cap * sizeof(FrameRec)))) return rc;
    if ((rc = ctx->stiles.ensure((size_t)((bit_cap + 4095) / 4096 + 1) * sizeof(SyncTile)))) return rc;
    if ((rc = ctx->scal.ensure(sizeof(DevScalars)))) return rc;
    hipStream_t st = ctx->stream;
    DevScalars *d_sc = (DevScalars *)ctx->scal.p;
    DevScalars sc;
    memset(&sc, 0, sizeof sc);
    sc.nbits = nbits;
    Plan &PL = ctx->plan;
    PL.clear();
    PL.side_stream = ctx->stream2;
    PL.copy(OP_H2D, d_sc, &sc, sizeof sc);
    if (nbits) PL.copy(OP_H2D, ctx->bits.p, bits_host, (size_t)nbits);
    PL.memset_async(ctx->frames.p, 0, (size_t)frame_cap * sizeof(FrameRec));
    PDT_LAUNCH(256, k_iota, dim3((unsigned)((bit_cap + 255) / 256)), dim3(256), 0, st, (unsigned *)ctx->bitsym.p,
               (long long *)ctx->symidx.p, bit_cap);
    launch_bytesync(ctx, PL, st, SP, d_sc, bit_cap, hit_cap, frame_cap, (long long)first_sync_end);
    DevScalars *back = ctx->pend_sc;                                  // pinned
    PL.copy(OP_D2H, back, d_sc, sizeof sc);
    {
        pdt_ctx *self = ctx;
        if ((rc = execute_plans(&self, 1))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(st));
    sc = *back;
    if (sc.nframes > frame_cap || (sc.sync_overflow && sc.nhits > hit_cap)) return PDT_ERR_STATE;
    std::vector<FrameRec> recs(sc.nframes);
    if (sc.nframes) HIP_TRY(hipMemcpy(recs.data(), ctx->frames.p, (size_t)sc.nframes * sizeof(FrameRec), hipMemcpyDeviceToHost));
    ctx->frames_host.resize(sc.nframes);
    ctx->frames_on_device = sc.nframes;
    ctx->have_frames = true;
    ctx->tip_host.clear();
    for (unsigned f = 0; f < sc.nframes; f++) {
        pdt_frame &o = ctx->frames_host[f];
        memset(&o, 0, sizeof o);
        o.bit_index = recs[f].bit_index;
        o.time_src = recs[f].time_src;
        o.time = (double)recs[f].time_src;
        o.inverted = recs[f].inverted;
        o.nbytes = recs[f].nbytes;
        o.complete = recs[f].complete;
        memcpy(o.bytes, recs[f].bytes, 104);
    }
    memset(&ctx->stats, 0, sizeof ctx->stats);
    ctx->stats.bits = nbits;
    ctx->stats.frames = sc.nframes;
    ctx->stats.sync_overflow = sc.sync_overflow;
    memset(ctx->stage_len, 0, sizeof ctx->stage_len);
    return PDT_OK;
}


int pdt_stage_pll(pdt_ctx *ctx, const void *iq_host, uint64_t n, int sample_format, pdt_pll_state *state, void *out_host,
                  void *lock_out_host, double *avg_phase_ret)
{
    if (!ctx || (!iq_host && n) || n >= (1ull << 31) || (sample_format != PDT_FMT_PCM16 && sample_format != PDT_FMT_F32)) return PDT_ERR_ARG;
    if (ctx->elem != 4 && sample_format == PDT_FMT_F32) return PDT_ERR_FORMAT;   // (double contexts: no `double complex` sample source)
    if (ctx->cfg.profile || ctx->stream_open) return PDT_ERR_STATE;
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    if (ctx->elem == 8) return stage_pll<double>(ctx, iq_host, n, sample_format, state, out_host, lock_out_host, avg_phase_ret);
    return stage_pll<float>(ctx, iq_host, n, sample_format, state, out_host, lock_out_host, avg_phase_ret);
}

int pdt_stage_gardner(pdt_ctx *ctx, const void *in_host, uint64_t n, uint64_t capacity, const void *neighbour_host,
                      pdt_gardner_state *state, void *out_host, uint64_t *pick_out, uint64_t *nsym_out)
{
    if (!ctx || !in_host || capacity == 0 || n > capacity || capacity >= (1ull << 30)) return PDT_ERR_ARG;
    if (ctx->cfg.sampler != PDT_SAMPLER_GARDNER) return PDT_ERR_ARG;
    if (ctx->stream_open) return PDT_ERR_STATE;              // (the stage buffers carry an open stream's tails)
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    if (ctx->elem == 8) return stage_gardner<double>(ctx, in_host, n, capacity, neighbour_host, state, out_host, pick_out, nsym_out);
    return stage_gardner<float>(ctx, in_host, n, capacity, neighbour_host, state, out_host, pick_out, nsym_out);
}

int pdt_stage_static_gain(pdt_ctx *ctx, const void *iq_host, uint64_t n, int sample_format, double level, double *gain_out)
{
    if (!ctx || !iq_host || n == 0 || n >= (1ull << 31) || (sample_format != PDT_FMT_PCM16 && sample_format != PDT_FMT_F32)) return PDT_ERR_ARG;
    if (ctx->stream_open) return PDT_ERR_STATE;              // (the stage buffers carry an open stream's tails)
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    if (ctx->elem == 8) return stage_static_gain<double>(ctx, iq_host, n, sample_format, level, gain_out);
    return stage_static_gain<float>(ctx, iq_host, n, sample_format, level, gain_out);
}

int pdt_stage_mm(pdt_ctx *ctx, const void *in_host, uint64_t n, pdt_mm_state *state, void *out_host, uint64_t *pick_out,
                 uint64_t *nsym_out)
{
    if (!ctx || (!in_host && n) || n >= (1ull << 30)) return PDT_ERR_ARG;
    {
        const double rg = ctx->cfg.mm_step_range != 0 ? ctx->cfg.mm_step_range : 3.0;
        const double baud = ctx->cfg.mode == PDT_MODE_ARGOS ? 800.0 : 16640.3;
        if (!(rg >= 0) || rg >= baud * 0.5) return PDT_ERR_ARG;                            // stepMax must stay positive and finite
    }
    if (ctx->stream_open) return PDT_ERR_STATE;              // (the stage buffers carry an open stream's tails)
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    if (ctx->elem == 8) return stage_mm<double>(ctx, in_host, n, state, out_host, pick_out, nsym_out);
    return stage_mm<float>(ctx, in_host, n, state, out_host, pick_out, nsym_out);
}

int pdt_stage_agc(pdt_ctx *ctx, void *data_host, uint64_t n, double initial, double attack, double decay, pdt_agc_state *state)
{
    if (!ctx || (!data_host && n) || n >= (1ull << 31)) return PDT_ERR_ARG;
    if (ctx->stream_open) return PDT_ERR_STATE;              // (the stage buffers carry an open stream's tails)
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    if (ctx->elem == 8) return stage_agc<double>(ctx, data_host, n, initial, attack, decay, state);
    return stage_agc<float>(ctx, data_host, n, initial, attack, decay, state);
}

int pdt_stage_squelch(pdt_ctx *ctx, void *data_host, const void *lock_host, uint64_t n, double threshold)
{
    if (!ctx || ((!data_host || !lock_host) && n) || n >= (1ull << 31)) return PDT_ERR_ARG;
    if (ctx->stream_open) return PDT_ERR_STATE;              // (the stage buffers carry an open stream's tails)
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    if (ctx->elem == 8) return stage_squelch<double>(ctx, data_host, lock_host, n, threshold);
    return stage_squelch<float>(ctx, data_host, lock_host, n, threshold);
}

int pdt_stage_manchester(pdt_ctx *ctx, const void *symbols_host, uint64_t nsymbols, double resync_threshold,
                         pdt_manchester_state *state, uint8_t *bits_out, uint32_t *bit_symbol_out, uint64_t *nbits_out)
{
    if (!ctx || (!symbols_host && nsymbols) || nsymbols >= (1ull << 31)) return PDT_ERR_ARG;
    if (nbits_out) *nbits_out = 0;
    if (nsymbols == 0) return PDT_OK;                                 // the loop body never runs: nothing changes
    if (ctx->stream_open) return PDT_ERR_STATE;              // (the stage buffers carry an open stream's tails)
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    if (ctx->elem == 8)
        return stage_manchester<double>(ctx, symbols_host, nsymbols, resync_threshold, state, bits_out, bit_symbol_out, nbits_out);
    return stage_manchester<float>(ctx, symbols_host, nsymbols, resync_threshold, state, bits_out, bit_symbol_out, nbits_out);
}

int pdt_stage_fir(pdt_ctx *ctx, const void *in_host, uint64_t n, pdt_fir_state *state, void *out_host)
{
    if (!ctx || (!in_host && n) || n >= (1ull << 31)) return PDT_ERR_ARG;
    if (ctx->stream_open) return PDT_ERR_STATE;              // (the stage buffers carry an open stream's tails)
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    if (ctx->elem == 8) return stage_fir<double>(ctx, in_host, n, state, out_host);
    return stage_fir<float>(ctx, in_host, n, state, out_host);
}

// ---------------------------------------------------------------- batched many-capture mode
int pdt_demod_batch_device(pdt_ctx *const *ctxs, const void *const *iq_device, const uint64_t *nframes, int count)
{
    if (count < 0 || (count && (!ctxs || !iq_device || !nframes))) return PDT_ERR_ARG;
    for (int i = 0; i < count; i++) {
        if (!ctxs[i] || (!iq_device[i] && nframes[i])) return PDT_ERR_ARG;
        if (ctxs[i]->stream_open) return PDT_ERR_STATE;
        for (int j = 0; j < i; j++)
            if (ctxs[j] == ctxs[i]) return PDT_ERR_ARG;               // one context per capture
    }
    int first_err = PDT_OK, enq = 0;
    for (; enq < count; enq++) {                                        // the plan of every capture (host only)
        pdt_ctx *c = ctxs[enq];
        c->batch_hint = 0;
        for (int k = 0; k < count; k++) c->batch_hint += (ctxs[k]->cfg.device == c->cfg.device) ? 1 : 0;
        c->pcm_dev = iq_device[enq];
        c->pcm_fmt = 0;
        const int rc = demod_common(c, nframes[enq], RUN_ENQUEUE);
        if (rc) { first_err = rc; break; }
    }
    // captures whose plans have the same shape (same mode, rate, chunk geometry, code path) and live on the same GPU
    // share their launches: one launch per stage for the whole group, the capture index in blockIdx.z
    std::vector<char> done((size_t)enq, 0);
    for (int i = 0; i < enq; i++) {
        if (done[(size_t)i]) continue;
        std::vector<pdt_ctx *> group{ctxs[i]};
        done[(size_t)i] = 1;
        for (int j = i + 1; j < enq && (int)group.size() < 65535; j++)
            if (!done[(size_t)j] && ctxs[j]->cfg.device == ctxs[i]->cfg.device && ctxs[i]->plan.same_shape(ctxs[j]->plan)) {
                group.push_back(ctxs[j]);
                done[(size_t)j] = 1;
            }
        const int rc = execute_plans(group.data(), (int)group.size());
        if (rc) {
            if (!first_err) first_err = rc;
            for (pdt_ctx *c : group) c->pending = false;
        }
    }
    for (int i = 0; i < enq; i++) {                                     // then collect them in order
        if (!ctxs[i]->pending) continue;
        const int rc = demod_common(ctxs[i], nframes[i], RUN_FINISH);
        if (rc && !first_err) first_err = rc;
    }
    return first_err;
}

// ---------------------------------------------------------------- streaming front end
int pdt_stream_begin(pdt_ctx *ctx)
{
    if (!ctx) return PDT_ERR_ARG;
    ctx->sc = StreamCarry();
    ctx->stream_have = 0;
    ctx->stream_done = 0;
    ctx->stream_total = 0;
    ctx->stream_fmt = -1;
    ctx->stream_open = false;
    ctx->stream_new.clear();
    ctx->frames_host.clear();
    ctx->chunk_host.clear();
    ctx->report_samples = 0;
    ctx->tip_host.clear();
    ctx->frames_on_device = 0;
    memset(&ctx->stats, 0, sizeof ctx->stats);
    memset(ctx->stage_len, 0, sizeof ctx->stage_len);
    ctx->stats.lock_sample = -1;
    ctx->stream_gpu_ms = 0;
    return PDT_OK;
}

// history a segment keeps in front of its new samples: the longest PLL warm-up (0.6 s of signal at most; the ARGOS build
// warms up for 8 s), the starting guess, and slack for the look-ahead of the block walkers
static uint64_t stream_history(const pdt_ctx *ctx)
{
    const double fs = (double)ctx->cfg.sample_rate;
    const double w = ctx->cfg.pll_warm ? (double)ctx->cfg.pll_warm : (ctx->cfg.mode == PDT_MODE_ARGOS ? 2.0 : 0.6) * fs;
    return (uint64_t)(w + 0.03 * fs) + 8192;
}

// local origin alignment: reference chunks (sampler seams), and for the interpolating FIR its ring of 26 inputs, whose
// phase decides the order of every output's accumulation
static uint64_t stream_align(const pdt_ctx *ctx)
{
    const uint64_t chunk = ctx->cfg.chunk;
    if (ctx->cfg.mode == PDT_MODE_ARGOS) return chunk;
    uint64_t a = chunk, b = 26;
    while (b) { const uint64_t t = a % b; a = b; b = t; }
    return chunk / a * 26;
}

// Demodulate the samples [stream_done, upto) of the window (whole chunks, or everything at the end of the stream) from the
// carried state, append the frames that became final to frames_host / stream_new, then let the window slide.
// segment_begin / segment_end bracket the run (demod_common in one piece, or its enqueue and finish phases with the wait for
// the segment's last bytes in between: demod_overlapped).
static void segment_begin(pdt_ctx *ctx, bool final_seg)
{
    StreamCarry &C = ctx->sc;
    C.active = true;
    C.final_seg = final_seg;
    C.first = (long long)ctx->stream_done;
    // (in place: the window is a view into the resident capture -- it "slides" by moving its base, not its samples)
    ctx->pcm_dev = (const unsigned char *)ctx->stream_in.p + (C.in_place ? (size_t)C.origin * (ctx->stream_fmt ? 8 : 4) : 0);
    ctx->pcm_fmt = ctx->stream_fmt;
}

static int segment_end(pdt_ctx *ctx, uint64_t upto, bool final_seg);

static int stream_segment(pdt_ctx *ctx, uint64_t upto, bool final_seg)
{
    segment_begin(ctx, final_seg);
    const int rc = demod_common(ctx, upto);
    ctx->sc.active = false;
    if (rc) return rc;
    return segment_end(ctx, upto, final_seg);
}

static int segment_end(pdt_ctx *ctx, uint64_t upto, bool final_seg)
{
    StreamCarry &C = ctx->sc;
    for (const pdt_frame &f : C.seg_frames) {
        ctx->frames_host.push_back(f);
        ctx->stream_new.push_back(f);
    }
    ctx->stream_done = upto;
    ctx->have_frames = true;
    ctx->frames_on_device = 0;                      // (the device holds the last segment's records only)
    pdt_stats &S = ctx->stats;
    S.samples = C.origin + upto;
    S.out_samples = S.samples * ctx->interp;
    S.symbols = C.nsym_total;
    S.bits = C.nbits_total;
    S.frames = ctx->frames_host.size();
    S.lock_sample = C.lock_sample;
    S.lock_freq_hz = C.lock_freq_hz;
    S.avg_phase = C.avg_at_lock;
    S.norm_factor = C.norm_factor;
    S.interp = ctx->interp;
    S.ntaps = ctx->ntaps;
    ctx->stream_gpu_ms += S.gpu_ms;
    S.gpu_ms = ctx->stream_gpu_ms;
    S.segments += 1;
    if (final_seg) return PDT_OK;
    // ---- slide the window: the new origin is the largest aligned position that leaves the history in front of the next
    // new sample; the input window and the tails later segments look back on move with it
    uint64_t align = stream_align(ctx);
    if (C.in_place && (align & 3)) align *= (align & 1) ? 4 : 2;     // the view's base stays 16-byte aligned
    // (the overlapped ingest cuts its segments where the whole-capture kernels' units begin, and keeps the window's origin --
    // and with it every segment's first new sample -- on the same grid: demod_overlapped, run_capture's seg_fast)
    if (C.place_align) align = C.place_align;
    const uint64_t hist = stream_history(ctx);
    const uint64_t done_g = C.origin + ctx->stream_done;
    const uint64_t new_origin = done_g > hist ? (done_g - hist) / align * align : 0;
    if (new_origin > C.origin) {
        const uint64_t d = new_origin - C.origin;
        const size_t fb = ctx->stream_fmt ? 8 : 4, es = (size_t)ctx->elem;
        const uint32_t ip = ctx->interp;
        const bool need_lock = ctx->cfg.mode == PDT_MODE_ARGOS || ctx->cfg.chain == PDT_CHAIN_LIVE;
        const uint64_t chunk = ctx->cfg.chunk;
        // (source and destination overlap: through a scratch buffer)
        auto slide = [&](DevBuf &buf, size_t elem, uint64_t src_first, uint64_t count) -> int {
            if (!count || !buf.p) return PDT_OK;
            int r = ctx->mag.ensure((size_t)count * elem + 64);
            if (r) return r;
            HIP_TRY(hipMemcpyAsync(ctx->mag.p, (unsigned char *)buf.p + (size_t)src_first * elem, (size_t)count * elem, hipMemcpyDeviceToDevice, ctx->stream));
            HIP_TRY(hipMemcpyAsync((unsigned char *)buf.p + (size_t)(src_first - d) * elem, ctx->mag.p, (size_t)count * elem, hipMemcpyDeviceToDevice, ctx->stream));
            return PDT_OK;
        };
        int r;
        // input: everything from the new origin on
        if (!C.in_place && (r = slide(ctx->stream_in, fb, d, ctx->stream_have - d))) return r;
        // PLL output: the FIR looks 25 inputs back; lock signal / AGC output: the sampler's stale reads reach one chunk back
        const uint64_t keep_pll = std::min<uint64_t>(ctx->stream_done - d, 256);
        {
            // these windows are indexed like the input (x interp for the AGC output): move [done - keep, done) down by d
            auto slide_tail = [&](DevBuf &buf, uint64_t scale, uint64_t keep) -> int {
                if (!buf.p) return PDT_OK;
                const uint64_t end = ctx->stream_done * scale, dd = d * scale;
                const uint64_t k = std::min<uint64_t>(keep, end - dd);
                int r2 = ctx->mag.ensure((size_t)k * es + 64);
                if (r2) return r2;
                HIP_TRY(hipMemcpyAsync(ctx->mag.p, (unsigned char *)buf.p + (size_t)(end - k) * es, (size_t)k * es, hipMemcpyDeviceToDevice, ctx->stream));
                HIP_TRY(hipMemcpyAsync((unsigned char *)buf.p + (size_t)(end - k - dd) * es, ctx->mag.p, (size_t)k * es, hipMemcpyDeviceToDevice, ctx->stream));
                return PDT_OK;
            };
            if ((r = slide_tail(ctx->pll, 1, keep_pll))) return r;
            if (need_lock && (r = slide_tail(ctx->lock, 1, chunk + 256))) return r;
            if ((r = slide_tail(ctx->agc, ip, (chunk + 256) * ip))) return r;
        }
        C.origin = new_origin;
        ctx->stream_have -= d;
        ctx->stream_done -= d;
    }
    return PDT_OK;
}

// A large capture from a file: the spans arrive in the background (ingest_capture with a job) straight into the stream window,
// which holds the whole capture (its origin moves, its samples never do); the chain runs over it in a few segments with carried
// state -- the streaming path -- each as soon as its samples are there.  What the call leaves behind is what pdt_stream_end
// leaves: frames and statistics; the stage arrays are those of the last segment.
//
// Round 5.  (i) The segments are cut on the grid where a segment may use the whole-capture kernels (run_capture's seg_fast:
// k_mix_fir, k_agc_block_tr, table rows of several chunks): a multiple of the reference chunk, of the AGC maps' runs, of a
// 128-byte line and of 16 x 13 chunks (so that the table rows of either span begin at the first new chunk); the window's origin
// stays on that grid too.  (ii) They are UNEQUAL.  A segment of the fraction x of an hour at 250 ksps costs about 7 + 11.5 x ms
// of GPU time (the 7: one PLL warm-up of ~100 000 steps and the other stages' latency floors, whatever the length), the
// segments run one after the other, and the ingest takes ~67 ms: segment j + 1's samples must take at least as long to arrive
// as segment j takes to run -- x_{j+1} >= 0.104 + 0.172 x_j -- and what is exposed behind the last byte is the last segment
// alone.  By that model three segments of 64 / 22 / 14 % leave 8.6 ms there (two: 76 / 24 %, 9.7 ms; four equal ones, round 3:
// 12 ms of floor each, 49 ms in all); measured round robin on one box, 55 / 28 / 17 % did best (beside the running ingest a
// segment costs 5 + 25 x ms) and is the default below.  (iii) A segment's launch plan is recorded BEFORE the host waits for its last span, and the text
// of a finished segment is formatted and written (text_fd) while the next one runs.
// ... and so are its per-chunk reports handed to the caller's progress function (pdt_set_progress)
struct TextSink {
    int fd = -1;
    uint64_t bytes = 0;
    int rc = PDT_OK;
    pdt_progress_fn fn = nullptr;
    void *user = nullptr;
    std::thread th;
    std::vector<pdt_frame> batch;
    std::vector<pdt_chunk_report> reports;
    uint64_t rep_first = 0;
    pdt_stats so_far;
    void wait() { if (th.joinable()) th.join(); }
    void work()
    {
        if (fd >= 0 && !batch.empty()) {
            uint64_t w = 0;
            const int r = pdt_write_records(batch.data(), batch.size(), fd, &w);
            bytes += w;
            if (r) rc = r;
        }
        if (fn && !reports.empty()) fn(user, rep_first, reports.data(), reports.size(), &so_far);
    }
    // (the previous batch is on the file and reported before the next one starts)
    void push(const std::vector<pdt_frame> &frames, std::vector<pdt_chunk_report> &&rep, uint64_t first_chunk, const pdt_stats &st, bool in_background)
    {
        if ((fd < 0 || frames.empty()) && (!fn || rep.empty())) return;
        wait();
        if (rc) return;
        reports = std::move(rep);
        rep_first = first_chunk;
        so_far = st;
        try {
            if (fd >= 0) batch = frames; else batch.clear();
            if (in_background) {
                th = std::thread([this] { work(); });
                return;
            }
        } catch (const std::bad_alloc &) {
            rc = PDT_ERR_NOMEM;
            return;
        } catch (const std::exception &) {
        }
        work();
    }
};

static uint64_t lcm_u64(uint64_t a, uint64_t b)
{
    uint64_t x = a, y = b;
    while (y) { const uint64_t t = x % y; x = y; y = t; }
    return a / x * b;
}

static int demod_overlapped(pdt_ctx *ctx, const IngestSrc &src, uint64_t nframes, int fmt, int text_fd, uint64_t *text_bytes)
{
    const size_t fb = fmt ? 8 : 4;
    const auto t_call = std::chrono::steady_clock::now();
    int rc = pdt_stream_begin(ctx);
    if (rc) return rc;
    ctx->stream_fmt = fmt;
    if ((rc = ctx->stream_in.ensure(((size_t)nframes + 64) * fb))) return rc;
    IngestJob job;
    ctx->sc.in_place = true;
    ctx->sc.quality = ctx->keep_quality;          // (every cut below is a chunk boundary)
    ctx->report_samples = nframes;
    const uint64_t chunk = ctx->cfg.chunk;
    // the grid of the segment boundaries and of the window's origin (see above)
    const uint64_t grid = lcm_u64(lcm_u64(chunk * 208, 64 * 26 * (uint64_t)ctx->interp), 416);
    std::vector<double> cut;                                  // cumulative fractions of the capture at the segments' ends
    if (ctx->tune.overlap_segments > 0) {
        for (int k = 1; k < ctx->tune.overlap_segments; k++) cut.push_back((double)k / ctx->tune.overlap_segments);
    } else if (ctx->tune.overlap_split[0] > 0) {
        double acc = 0;
        for (int k = 0; k < 8 && ctx->tune.overlap_split[k] > 0; k++) { acc += ctx->tune.overlap_split[k]; if (acc < 1.0) cut.push_back(acc); }
    } else {
        cut = { 0.55, 0.83 };
    }
    std::vector<uint64_t> ends;
    if (grid * 8 <= nframes) {
        ctx->sc.place_align = grid;
        for (double c : cut) {
            const uint64_t e = (uint64_t)llround(c * (double)nframes / (double)grid) * grid;
            if (e > (ends.empty() ? 0 : ends.back()) && e + grid <= nframes) ends.push_back(e);
        }
    } else {
        // (an unusual chunk size: no such grid inside the capture -- the round-3 form: equal segments on chunk boundaries, the
        // stream path's kernels)
        for (double c : cut) {
            const uint64_t e = (uint64_t)(c * (double)nframes) / chunk * chunk;
            if (e > (ends.empty() ? 0 : ends.back()) && e < nframes) ends.push_back(e);
        }
    }
    ends.push_back(nframes);
    for (uint64_t e : ends) job.mark_bytes.push_back((size_t)e * fb);
    if ((rc = ingest_capture(ctx, src, (size_t)nframes * fb, ctx->stream_in.p, &job))) {
        (void)ingest_join(job);
        ctx->sc = StreamCarry();
        return rc;
    }
    TextSink sink;
    sink.fd = text_fd;
    sink.fn = ctx->progress_fn;
    sink.user = ctx->progress_user;
    uint64_t reported = 0;                        // chunks whose reports have been handed on
    auto segment_reports = [&]() {                // those of the segment that has just ended
        std::vector<pdt_chunk_report> r;
        const uint64_t have = ctx->chunk_host.size();
        if (sink.fn && have > reported) {
            r.resize((size_t)(have - reported));
            chunk_reports_range(ctx, reported, have, nframes, r.data(), ctx->sc.have_pending ? &ctx->sc.pending : nullptr);
        }
        return r;
    };
    ctx->batch_hint = 1;
    for (size_t k = 0; k < ends.size() && !rc; k++) {
        const bool last = k + 1 == ends.size();
        const uint64_t upto = ends[k];
        const auto t0 = std::chrono::steady_clock::now();
        ctx->stream_have = upto - ctx->sc.origin;                   // (window-local, as the pushes keep it)
        ctx->stream_total = upto;
        const uint64_t win = upto - ctx->sc.origin;
        segment_begin(ctx, last);
        rc = demod_common(ctx, win, RUN_ENQUEUE);                   // the segment's launch plan (host only) ...
        const auto t1 = std::chrono::steady_clock::now();
        if (!rc) rc = ingest_wait_mark(ctx, job, k, ctx->stream);      // ... then its last bytes ...
        const auto t2 = std::chrono::steady_clock::now();
        if (last) ctx->ingest_ms = std::chrono::duration<double, std::milli>(t2 - t_call).count();
        if (!rc) {
            pdt_ctx *self = ctx;
            rc = execute_plans(&self, 1);                           // ... then the launches
        }
        const auto t2a = std::chrono::steady_clock::now();
        if (!rc) rc = demod_common(ctx, win, RUN_FINISH);
        const auto t2b = std::chrono::steady_clock::now();
        ctx->sc.active = false;
        ctx->pending = false;
        if (!rc) rc = segment_end(ctx, win, last);
        const auto t2c = std::chrono::steady_clock::now();
        if (!rc && !last) {
            try {
                const uint64_t first_chunk = reported;
                std::vector<pdt_chunk_report> r = segment_reports();
                reported += r.size();
                sink.push(ctx->sc.seg_frames, std::move(r), first_chunk, ctx->stats, true);
            } catch (const std::bad_alloc &) {
                rc = PDT_ERR_NOMEM;
            }
        }
        if (ctx->tune.debug_overlap) {
            const auto t3 = std::chrono::steady_clock::now();
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "segment %zu: upto %llu window %llu: plan %.2f ms, waited %.2f ms for the spans, run %.2f ms = launch %.2f + finish %.2f + end %.2f + sink %.2f (gpu %.2f so far)\n", k,
                    (unsigned long long)upto, (unsigned long long)win, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t2, t2a), ms(t2a, t2b), ms(t2b, t2c), ms(t2c, t3), ctx->stats.gpu_ms);
            for (const pdt_kernel_time &kt : ctx->ktimes) fprintf(stderr, "    %-16s %8.3f ms\n", kt.name, kt.total_ms);
        }
    }
    ctx->sc.in_place = false;
    const int rj = ingest_join(job);
    sink.wait();
    if (!rc && !rj) {
        if (sink.rc) rc = sink.rc;
        else {
            try {
                const uint64_t first_chunk = reported;
                std::vector<pdt_chunk_report> r = segment_reports();
                sink.push(ctx->sc.seg_frames, std::move(r), first_chunk, ctx->stats, false);
                rc = sink.rc;
            } catch (const std::bad_alloc &) {
                rc = PDT_ERR_NOMEM;
            }
        }
    }
    if (text_bytes) *text_bytes = sink.bytes;
    ctx->stats.ingest_ms = ctx->ingest_ms;
    ctx->stats.ingest_direct = (uint32_t)ctx->ingest_was_direct;
    ctx->stats.ingest_numa_node = ctx->ingest_numa_node;
    // the stream machinery was borrowed: leave no stream behind (a later push starts a new one), keep frames and statistics
    ctx->sc = StreamCarry();
    ctx->stream_have = ctx->stream_done = ctx->stream_total = 0;
    ctx->stream_fmt = -1;
    ctx->stream_open = false;
    return rc ? rc : rj;
}

// A capture that does not fit the device in one piece (window_piece_for): the streaming path with a bounded window, fed from
// the file (or the caller's memory) `piece` samples at a time -- what a caller of pdt_stream_push_* would do by hand, with the
// threaded ingest in place of one pageable copy per push, the text of a finished piece written and its per-chunk reports
// handed on while the next piece is read (as demod_overlapped does).  Every cut is a chunk boundary; the window's origin
// stays on the grid where a segment may take the whole-capture kernels (run_capture: seg_fast) when the pieces are large
// enough for that to matter.  The state it leaves is demod_overlapped's: frames, text, statistics and reports of the whole
// capture, stage arrays of the last piece.
static int demod_windowed(pdt_ctx *ctx, const IngestSrc &src, uint64_t nframes, int fmt, int text_fd, uint64_t *text_bytes, uint64_t piece)
{
    if (ctx->keep_agc_raw) return PDT_ERR_NOMEM;      // (the pre-Squelch stream of the WHOLE capture was asked for: that does not fit)
    const size_t fb = fmt ? 8 : 4;
    const auto t_call = std::chrono::steady_clock::now();
    int rc = pdt_stream_begin(ctx);
    if (rc) return rc;
    ctx->stream_fmt = fmt;
    ctx->sc.quality = ctx->keep_quality;              // (every cut below is a chunk boundary)
    ctx->report_samples = nframes;
    const uint64_t chunk = ctx->cfg.chunk;
    const uint64_t grid = lcm_u64(lcm_u64(chunk * 208, 64 * 26 * (uint64_t)ctx->interp), 416);
    if (piece >= 4 * grid) {
        piece = piece / grid * grid;
        ctx->sc.place_align = grid;
    }
    TextSink sink;
    sink.fd = text_fd;
    sink.fn = ctx->progress_fn;
    sink.user = ctx->progress_user;
    uint64_t reported = 0, pushed = 0;
    double ingest_ms = 0;
    auto segment_reports = [&]() {
        std::vector<pdt_chunk_report> r;
        const uint64_t have = ctx->chunk_host.size();
        if (sink.fn && have > reported) {
            r.resize((size_t)(have - reported));
            chunk_reports_range(ctx, reported, have, nframes, r.data(), ctx->sc.have_pending ? &ctx->sc.pending : nullptr);
        }
        return r;
    };
    for (bool last = false; !last && !rc;) {
        const uint64_t cnt = std::min<uint64_t>(piece, nframes - pushed);
        last = pushed + cnt == nframes;
        if ((rc = ctx->stream_in.ensure_keep(((size_t)(ctx->stream_have + cnt) + 64) * fb, (size_t)ctx->stream_have * fb))) break;
        IngestSrc s = src;
        if (s.mem) s.mem += (size_t)pushed * fb; else s.off += pushed * fb;
        const auto t0 = std::chrono::steady_clock::now();
        if ((rc = ingest_capture(ctx, s, (size_t)cnt * fb, (unsigned char *)ctx->stream_in.p + (size_t)ctx->stream_have * fb))) break;
        ingest_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        ctx->stream_have += cnt;
        ctx->stream_total += cnt;
        pushed += cnt;
        const uint64_t upto = last ? ctx->stream_have : (ctx->sc.origin + ctx->stream_have) / chunk * chunk - ctx->sc.origin;
        if (upto <= ctx->stream_done && !last) continue;
        if ((rc = stream_segment(ctx, upto, last))) break;
        try {
            const uint64_t first_chunk = reported;
            std::vector<pdt_chunk_report> r = segment_reports();
            reported += r.size();
            sink.push(ctx->sc.seg_frames, std::move(r), first_chunk, ctx->stats, !last);
        } catch (const std::bad_alloc &) {
            rc = PDT_ERR_NOMEM;
        }
    }
    sink.wait();
    if (!rc && sink.rc) rc = sink.rc;
    if (text_bytes) *text_bytes = sink.bytes;
    (void)t_call;
    ctx->ingest_ms = ingest_ms;
    ctx->stats.ingest_ms = ingest_ms;
    ctx->stats.ingest_direct = (uint32_t)ctx->ingest_was_direct;
    ctx->stats.ingest_numa_node = ctx->ingest_numa_node;
    ctx->stats.windowed = 1;
    ctx->sc = StreamCarry();
    ctx->stream_have = ctx->stream_done = ctx->stream_total = 0;
    ctx->stream_fmt = -1;
    ctx->stream_open = false;
    return rc;
}

int pdt_demod_file(pdt_ctx *ctx, int fd, uint64_t byte_offset, uint64_t nframes, int sample_format, int text_fd, uint64_t *text_bytes)
{
    if (text_bytes) *text_bytes = 0;
    if (!ctx || fd < 0 || text_fd < 0 || (sample_format != PDT_FMT_PCM16 && sample_format != PDT_FMT_F32)) return PDT_ERR_ARG;
    if (sample_format == PDT_FMT_F32 && ctx->elem != 4) return PDT_ERR_FORMAT;
    if (ctx->stream_open) return PDT_ERR_STATE;
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    const size_t fb = sample_format == PDT_FMT_F32 ? 8 : 4;
    if (const long long piece = window_piece_for(ctx, nframes, fb)) {
        IngestSrc src;
        src.fd = fd;
        src.off = byte_offset;
        return piece < 0 ? (int)piece : demod_windowed(ctx, src, nframes, sample_format == PDT_FMT_F32 ? 1 : 0, text_fd, text_bytes, (uint64_t)piece);
    }
    if (overlap_ingest(ctx, nframes, fb)) {
        IngestSrc src;
        src.fd = fd;
        src.off = byte_offset;
        return demod_overlapped(ctx, src, nframes, sample_format == PDT_FMT_F32 ? 1 : 0, text_fd, text_bytes);
    }
    const int rc = pdt_demod_fd(ctx, fd, byte_offset, nframes, sample_format);
    if (rc) return rc;
    return pdt_write_frames(ctx, text_fd, text_bytes);
}

static int stream_push(pdt_ctx *ctx, const void *host, uint64_t nframes, int fmt, uint64_t *new_frames)
{
    if (!ctx || (!host && nframes)) return PDT_ERR_ARG;
    if (fmt == 1 && ctx->elem != 4) return PDT_ERR_FORMAT;
    if (!ctx->stream_open) {                         // the first push opens a stream (as if pdt_stream_begin had been called)
        int rb = pdt_stream_begin(ctx);
        if (rb) return rb;
        ctx->stream_open = true;
    }
    if (ctx->stream_fmt >= 0 && ctx->stream_fmt != fmt) return PDT_ERR_STATE;
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    ctx->stream_fmt = fmt;
    const size_t fb = fmt ? 8 : 4;
    ctx->stream_new.clear();
    if (new_frames) *new_frames = 0;
    // append to the window (it grows only with the size of the pushes, not with the length of the stream)
    int rc = ctx->stream_in.ensure_keep(((size_t)(ctx->stream_have + nframes) + 64) * fb, (size_t)ctx->stream_have * fb);
    if (rc) return rc;
    if (nframes)
        HIP_TRY(hipMemcpyAsync((unsigned char *)ctx->stream_in.p + (size_t)ctx->stream_have * fb, host, (size_t)nframes * fb,
                               hipMemcpyHostToDevice, ctx->stream));
    ctx->stream_have += nframes;
    ctx->stream_total += nframes;
    const uint64_t chunk = ctx->cfg.chunk;
    const uint64_t whole = (ctx->sc.origin + ctx->stream_have) / chunk * chunk - ctx->sc.origin;   // local end of the complete chunks
    if (whole > ctx->stream_done) {
        rc = stream_segment(ctx, whole, false);
        if (rc) return rc;
    } else if (nframes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));      // the caller's buffer is free to be reused when this returns
    }
    if (new_frames) *new_frames = ctx->stream_new.size();
    return PDT_OK;
}

int pdt_stream_push_pcm16(pdt_ctx *ctx, const int16_t *iq_host, uint64_t nframes, uint64_t *new_frames)
{
    return stream_push(ctx, iq_host, nframes, 0, new_frames);
}

int pdt_stream_push_f32(pdt_ctx *ctx, const float *iq_host, uint64_t nframes, uint64_t *new_frames)
{
    return stream_push(ctx, iq_host, nframes, 1, new_frames);
}

int pdt_stream_end(pdt_ctx *ctx, uint64_t *new_frames)
{
    if (!ctx) return PDT_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    ctx->stream_open = false;                        // (whatever happens below, the stream is over)
    if (ctx->stream_fmt < 0) {                       // nothing was pushed: an empty capture
        ctx->stream_fmt = 0;
        int rc = ctx->stream_in.ensure(64);
        if (rc) return rc;
    }
    ctx->stream_new.clear();
    int rc = PDT_OK;
    if (ctx->stream_have > ctx->stream_done || ctx->stream_total == 0) {
        rc = stream_segment(ctx, ctx->stream_have, true);            // the short last chunk
    } else if (ctx->sc.have_pending) {
        // the stream ended on a chunk boundary inside a frame: the reference leaves that frame partial (Q11)
        ctx->frames_host.push_back(ctx->sc.pending);
        ctx->stream_new.push_back(ctx->sc.pending);
        ctx->sc.have_pending = false;
        ctx->stats.frames = ctx->frames_host.size();
    }
    if (new_frames) *new_frames = ctx->stream_new.size();
    return rc;
}

uint64_t pdt_stream_retained(const pdt_ctx *ctx) { return ctx ? ctx->stream_have : 0; }

uint64_t pdt_stream_frames(const pdt_ctx *ctx, pdt_frame *out, uint64_t max_frames)
{
    if (!ctx) return 0;
    const uint64_t n = std::min<uint64_t>(max_frames, ctx->stream_new.size());
    if (out && n) memcpy(out, ctx->stream_new.data(), (size_t)n * sizeof(pdt_frame));
    return n;
}

// ---------------------------------------------------------------- frame validation
static_assert(sizeof(pdt_tip_frame) == sizeof(pdt::TipFrame) && sizeof(pdt_tip_frame) == 12, "pdt_tip_frame layout");

int pdt_tip_check(pdt_ctx *ctx, pdt_tip_summary *out)
{
    if (!ctx || !out) return PDT_ERR_ARG;
    if (ctx->cfg.mode != PDT_MODE_POES) return PDT_ERR_ARG;             // TIP minor frames only
    if (hipSetDevice(ctx->cfg.device) != hipSuccess) return PDT_ERR_NOGPU;
    const unsigned nf = ctx->frames_on_device;
    if (!ctx->have_frames || nf != ctx->frames_host.size()) return PDT_ERR_STATE;   // nothing demodulated yet
    memset(out, 0, sizeof *out);
    out->spacecraft = -1;
    out->day = -1;
    out->t0_ms = -1;
    ctx->tip_host.assign(nf, pdt_tip_frame{});
    if (nf == 0) return PDT_OK;
    int rc;
    if ((rc = ctx->tip.ensure(sizeof(TipCounters) + (size_t)nf * sizeof(TipFrame)))) return rc;
    TipCounters *d_cnt = (TipCounters *)ctx->tip.p;
    TipFrame *d_rec = (TipFrame *)(d_cnt + 1);
    hipStream_t st = ctx->stream;
    Plan &PL = ctx->plan;
    PL.clear();
    PL.side_stream = ctx->stream2;
    PL.memset_async(d_cnt, 0, sizeof(TipCounters));
    PDT_LAUNCH(256, k_tip_check, dim3((nf + 255) / 256), dim3(256), 0, st, (const FrameRec *)ctx->frames.p, nf, d_rec, d_cnt);
    TipCounters cnt;
    PL.copy(OP_D2H, &cnt, d_cnt, sizeof cnt);
    PL.copy(OP_D2H, ctx->tip_host.data(), d_rec, (size_t)nf * sizeof(TipFrame));
    {
        pdt_ctx *self = ctx;
        if ((rc = execute_plans(&self, 1))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(st));
    out->frames_checked = cnt.frames_checked;
    out->good_frames = cnt.good_frames;
    out->bad_chunks = cnt.bad_chunks;
    out->good_chunks = 5 * cnt.frames_checked - cnt.bad_chunks;
    out->time_frames = cnt.time_frames;
    // MATLAB mode(): most frequent value, the smallest one on ties
    unsigned best = 0;
    for (int v = 0; v < 256; v++) if (cnt.hist_sc[v] > best) { best = cnt.hist_sc[v]; out->spacecraft = v; }
    best = 0;
    for (int v = 0; v < 512; v++) if (cnt.hist_day[v] > best) { best = cnt.hist_day[v]; out->day = v; }
    // T0 = spacecraft ms of day minus the local frame time (daytimeDecode.m:26,34); frameTime is what the text
    // file holds ("%.5f"); the time stamps live on the host (pdt_timeaxis.h), so this small reduction does too
    std::vector<double> t0;
    for (unsigned f = 0; f < nf; f++) {
        const pdt_tip_frame &r = ctx->tip_host[f];
        if (r.has_time && r.day_ms >= 0) {
            const double t = round(ctx->frames_host[f].time * 1e5) / 1e5;
            const double v = (double)r.day_ms - t * 1000.0;
            if (v > 0) t0.push_back(round(v));
        }
    }
    std::sort(t0.begin(), t0.end());
    size_t best_n = 0;
    for (size_t i = 0; i < t0.size();) {
        size_t j = i;
        while (j < t0.size() && t0[j] == t0[i]) j++;
        if (j - i > best_n) { best_n = j - i; out->t0_ms = (int64_t)t0[i]; }
        i = j;
    }
    return PDT_OK;
}

uint64_t pdt_tip_frames(const pdt_ctx *ctx, pdt_tip_frame *out, uint64_t max_frames)
{
    if (!ctx) return 0;
    const uint64_t n = std::min<uint64_t>(max_frames, ctx->tip_host.size());
    if (out && n) memcpy(out, ctx->tip_host.data(), (size_t)n * sizeof(pdt_tip_frame));
    return n;
}

uint64_t pdt_num_frames(const pdt_ctx *ctx) { return ctx ? ctx->frames_host.size() : 0; }

uint64_t pdt_frames(const pdt_ctx *ctx, pdt_frame *out, uint64_t max_frames)
{
    if (!ctx) return 0;
    const uint64_t n = std::min<uint64_t>(max_frames, ctx->frames_host.size());
    if (out && n) memcpy(out, ctx->frames_host.data(), (size_t)n * sizeof(pdt_frame));
    return n;
}

int pdt_get_stats(const pdt_ctx *ctx, pdt_stats *out)
{
    if (!ctx || !out) return PDT_ERR_ARG;
    *out = ctx->stats;
    out->alloc_ms = (double)g_alloc_ns.load() * 1e-6;
    return PDT_OK;
}

// "%.5f" of a non-negative double without printf: value * 10^5 rounded to nearest, ties to even, on the EXACT binary value
// -- what glibc prints.  x = m * 2^e with a 53-bit m; m * 100000 fits 70 bits.  Returns the number of characters.
enum { PDT_TIME5_MAX = 336 };
static int format_time5(double x, char *out)
{
    if (!(x >= 0.0) || x >= 1e15) return snprintf(out, PDT_TIME5_MAX, "%.5f", x);   // (never the reference's range; DBL_MAX prints 315 characters)
    uint64_t bits;
    memcpy(&bits, &x, sizeof bits);
    const int be = (int)((bits >> 52) & 0x7ff);
    uint64_t m = bits & ((1ull << 52) - 1);
    int e;
    if (be == 0) e = -1074; else { m |= 1ull << 52; e = be - 1075; }
    unsigned __int128 N = (unsigned __int128)m * 100000u;
    unsigned __int128 q;
    if (e >= 0) {
        q = N << e;                                   // x < 1e15: no overflow
    } else {
        const int sh = -e;
        if (sh >= 120) q = 0;                         // x * 1e5 < 2^-49: rounds to zero
        else {
            q = N >> sh;
            const unsigned __int128 rem = N & (((unsigned __int128)1 << sh) - 1), half = (unsigned __int128)1 << (sh - 1);
            if (rem > half || (rem == half && (q & 1))) q++;
        }
    }
    const uint64_t ip = (uint64_t)(q / 100000u);
    unsigned fp = (unsigned)(q % 100000u);
    char tmp[24];
    int n = 0;
    uint64_t v = ip;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    int w = 0;
    while (n) out[w++] = tmp[--n];
    out[w++] = '.';
    for (int d = 4; d >= 0; d--) { out[w + d] = (char)('0' + fp % 10); fp /= 10; }
    return w + 5;
}

uint64_t pdt_format_records(const pdt_frame *frames, uint64_t nframes, char *buf, uint64_t cap)
{
    // ByteSync.c:96-99,126-129 ("%.5f " / "%.5fi "), :62,100-101 ("%.2X "), :66-70 (newline after the last byte)
    static const char hex[] = "0123456789ABCDEF";
    uint64_t need = 0;
    char line[PDT_TIME5_MAX + 4 + 3 * 104 + 2];
    for (uint64_t k = 0; k < nframes; k++) {
        const pdt_frame &f = frames[k];
        int w = format_time5(f.time, line);
        if (f.inverted) line[w++] = 'i';
        line[w++] = ' ';
        const unsigned nb = f.nbytes > 104 ? 104u : f.nbytes;
        for (unsigned b = 0; b < nb; b++) {
            line[w++] = hex[f.bytes[b] >> 4];
            line[w++] = hex[f.bytes[b] & 15];
            line[w++] = ' ';
        }
        if (f.complete) line[w++] = '\n';
        if (buf && need < cap) memcpy(buf + need, line, (size_t)std::min<uint64_t>((uint64_t)w, cap - need));
        need += (uint64_t)w;
    }
    return need;
}

uint64_t pdt_format_frames(const pdt_ctx *ctx, char *buf, uint64_t cap)
{
    if (!ctx) return 0;
    return pdt_format_records(ctx->frames_host.data(), ctx->frames_host.size(), buf, cap);
}

// The reference writes its output file byte by byte while it demodulates (fprintf, ByteSync.c:62-101).  Here the text exists
// only after the run: an hour of POES is 36 000 lines, 12 MB.  A few threads take a slice of the frames each: format it into a
// buffer of their own and pwrite it at its place as soon as the sizes of the slices in front are known (the writes queue on
// the inode, but behind the formatting of the other slices instead of after it).  Measured on the GPU box's tmpfs: formatting
// 2.5 ms on one thread, the write 2.3 ms; a shared mapping of the file filled in place by all threads was slower (4.3 ms: the
// page faults of a fresh tmpfs mapping cost more than the copy they save).  A descriptor that cannot seek takes the text in order.
int pdt_write_records(const pdt_frame *frames, uint64_t nframes, int fd, uint64_t *bytes_written)
{
    if (bytes_written) *bytes_written = 0;
    if (fd < 0 || (!frames && nframes)) return PDT_ERR_ARG;
    if (!nframes) return PDT_OK;
    off_t at0 = lseek(fd, 0, SEEK_CUR);
    {
        // pwrite on an O_APPEND descriptor ignores its offset and appends (Linux): the slices would land in completion order
        const int fl = fcntl(fd, F_GETFL);
        if (fl >= 0 && (fl & O_APPEND)) at0 = -1;            // ... so such a descriptor takes the text in order, like a pipe
    }
    const int T = at0 < 0 ? 1 : (int)std::max<uint64_t>(1, std::min<uint64_t>(6, nframes / 2048));
    std::vector<uint64_t> first;
    std::unique_ptr<std::atomic<long long>[]> size;
    try {                                                     // (no exception crosses the C boundary, in this thread or a worker)
        first.resize((size_t)T + 1);
        size.reset(new std::atomic<long long>[(size_t)T]);
    } catch (const std::bad_alloc &) { return PDT_ERR_NOMEM; }
    for (int t = 0; t <= T; t++) first[(size_t)t] = nframes * (uint64_t)t / (uint64_t)T;
    for (int t = 0; t < T; t++) size[(size_t)t].store(-1);
    std::atomic<int> bad{0};                                  // 1: write error, 2: out of memory
    auto work = [&](int t) {
        const uint64_t a = first[(size_t)t], b = first[(size_t)t + 1];
        std::unique_ptr<char[]> buf(new (std::nothrow) char[(size_t)(b - a) * (PDT_TIME5_MAX + 4 + 3 * 104 + 2)]);     // (not touched beyond the text)
        if (!buf) {
            bad = 2;
            size[(size_t)t].store(0, std::memory_order_release);         // (the slices behind this one must not wait for ever)
            return;
        }
        const uint64_t sz = pdt_format_records(frames + a, b - a, buf.get(), ~0ull);
        size[(size_t)t].store((long long)sz, std::memory_order_release);
        uint64_t off = 0;
        for (int u = 0; u < t; u++) {
            long long v;
            while ((v = size[(size_t)u].load(std::memory_order_acquire)) < 0) std::this_thread::yield();
            off += (uint64_t)v;
        }
        size_t done = 0;
        while (done < sz && !bad) {
            const ssize_t r = at0 >= 0 ? pwrite(fd, buf.get() + done, sz - done, at0 + (off_t)off + (off_t)done)
                                       : write(fd, buf.get() + done, sz - done);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) { bad = 1; return; }
            done += (size_t)r;
        }
    };
    std::vector<std::thread> pool;
    int started = 1;
    try {
        for (int t = 1; t < T; t++, started++) pool.emplace_back(work, t);
    } catch (const std::exception &) {                        // std::system_error (no thread), std::bad_alloc
    }
    work(0);
    for (int t = started; t < T; t++) work(t);                // slices that got no thread: here, in order
    for (auto &th : pool) th.join();
    if (bad == 2) return PDT_ERR_NOMEM;
    if (bad) return PDT_ERR_IO;
    uint64_t total = 0;
    for (int t = 0; t < T; t++) total += (uint64_t)size[(size_t)t].load();
    if (at0 >= 0) (void)lseek(fd, at0 + (off_t)total, SEEK_SET);
    if (bytes_written) *bytes_written = total;
    return PDT_OK;
}

int pdt_write_frames(const pdt_ctx *ctx, int fd, uint64_t *bytes_written)
{
    if (!ctx) return PDT_ERR_ARG;
    return pdt_write_records(ctx->frames_host.data(), ctx->frames_host.size(), fd, bytes_written);
}

uint64_t pdt_stage_len(const pdt_ctx *ctx, int stage)
{
    if (!ctx || stage < 0 || stage >= PDT_ST_COUNT) return 0;
    return ctx->stage_len[stage];
}

int64_t pdt_read_stage(const pdt_ctx *ctx, int stage, uint64_t first, uint64_t count, void *out)
{
    if (!ctx || !out || stage < 0 || stage >= PDT_ST_COUNT) return PDT_ERR_ARG;
    const uint64_t len = ctx->stage_len[stage];
    if (first >= len) return 0;
    count = std::min<uint64_t>(count, len - first);
    const void *src = nullptr;
    size_t es = ctx->elem;
    switch (stage) {
    case PDT_ST_PLL: src = ctx->pll.p; break;
    case PDT_ST_LOCK: src = ctx->lock.p; break;
    case PDT_ST_FIR: src = ctx->fir.p; break;
    case PDT_ST_AGC: src = ctx->agc.p; break;
    case PDT_ST_AGC_RAW: src = ctx->agc_raw.p; break;
    case PDT_ST_SYM: src = ctx->sym.p; break;
    case PDT_ST_SYMIDX: src = ctx->symidx.p; es = 8; break;
    case PDT_ST_BITS: src = ctx->bits.p; es = 1; break;
    case PDT_ST_BITSYM: src = ctx->bitsym.p; es = 4; break;
    }
    if (!src) return PDT_ERR_STATE;
    if (hipSetDevice(ctx->cfg.device) != hipSuccess) return PDT_ERR_NOGPU;
    if (hipMemcpy(out, (const unsigned char *)src + first * es, (size_t)count * es, hipMemcpyDeviceToHost) != hipSuccess)
        return PDT_ERR_NOGPU;
    return (int64_t)count;
}

int pdt_kernel_times(const pdt_ctx *ctx, pdt_kernel_time *out, int max_entries)
{
    if (!ctx) return 0;
    const int n = std::min<int>(max_entries, (int)ctx->ktimes.size());
    if (out)
        for (int i = 0; i < n; i++) out[i] = ctx->ktimes[i];
    return (int)ctx->ktimes.size();
}

}  // extern "C"

