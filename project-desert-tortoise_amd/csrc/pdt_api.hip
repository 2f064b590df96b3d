// pdt_api.hip -- libpdt.so: context management, kernel orchestration and the C ABI of
// include/pdt.h.  Compiled for gfx950 only, with -ffp-contract=off (see pdt_device_math.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <errno.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <functional>
#include <map>
#include <mutex>
#include <atomic>
#include <chrono>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/pdt.h"
#include "../../include/pdt_dev.h"
#include "pdt_kernels_back.h"
#include "pdt_kernels_front.h"
#include "pdt_timeaxis.h"

using namespace pdt;

static std::atomic<int> g_open_contexts{0};          // contexts alive in this process (pdt_open / pdt_close)
// One ingest per GPU at a time.  Two contexts on one GPU (bin/demodMulti's two lanes) that read their captures at once would
// share the PCIe link and both arrive late; taking turns, the second one's capture arrives while the first one's chain runs.
static std::mutex g_link_mu[64];

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "libpdt: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, \
                    __LINE__);                                                                         \
            return PDT_ERR_NOGPU;                                                                      \
        }                                                                                              \
    } while (0)

namespace {

// host time this process has spent allocating device and pinned memory (pdt_stats.alloc_ms: the cold path's breakdown)
static std::atomic<long long> g_alloc_ns{0};
struct AllocTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~AllocTimer() { g_alloc_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
static hipError_t timed_host_malloc(void **p, size_t bytes)
{
    AllocTimer t;
    return hipHostMalloc(p, bytes, hipHostMallocDefault);
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return PDT_OK;
        AllocTimer t;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 16 + 4096;
        if (hipMalloc(&p, want) != hipSuccess) {
            (void)hipGetLastError();
            return PDT_ERR_NOMEM;
        }
        cap = want;
        return PDT_OK;
    }
    // grow, keeping the first `keep` bytes (the windows of a stream hold history the next segment reads)
    int ensure_keep(size_t bytes, size_t keep)
    {
        if (bytes <= cap) return PDT_OK;
        AllocTimer t;
        void *np = nullptr;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipMalloc(&np, want) != hipSuccess) {
            (void)hipGetLastError();
            return PDT_ERR_NOMEM;
        }
        if (p && keep) (void)hipMemcpy(np, p, std::min(keep, cap), hipMemcpyDeviceToDevice);
        if (p) (void)hipFree(p);
        p = np;
        cap = want;
        return PDT_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct KTimer {
    std::string name;
    hipEvent_t a, b;
    bool shared_a = false;      // a is the previous group's b (returned to the pool once)
};

// device-side scalar block shared by all stages of one run
struct DevScalars {
    unsigned long long nsym;
    unsigned long long nbits;
    unsigned nhits;
    unsigned nframes;
    unsigned counters[4];   // pll blocks, pll fixes, agc blocks, agc fixes
    unsigned sync_overflow; // a 4096-bit tile held more than 31 sync hits: generic path used
    unsigned pad0_;
    unsigned gstats[4];     // boundary-state tables: [0] exits outside the domain, [1] full-domain chunks,
                            //                        [2] chunks the chain had to walk, [3] unused
    double norm;            // storage for the normalisation factor (float or double)
    long long agc_first_bad; // first AGC seam that does not close (k_agc_scan -> k_agc_fix)
    PllPhaseHint phase_hint;     // k_pll_phase -> k_pll_head: workgroups finished, the clock at its start (walk on while it runs)
};

// ---------------------------------------------------------------- launch plans
// Every kernel of pdt_kernels_*.h is a __device__ body; it is entered through k_run, which reads the body's
// arguments from a device array of argument packs indexed by blockIdx.z = the capture.  A demodulation call first
// records its launches, memsets, stream fork/joins and read-back copies as a PLAN (host only); the plan is then
// executed -- alone (grid.z = 1) or zipped with the plans of other captures of the same shape (grid.z = M): ONE launch per
// stage for the whole batch, so that the serial, few-wavefront kernels of all captures run side by side instead of queueing
// behind each other on the hardware queues (batched many-capture mode).
template <typename... A> struct Pack {};
template <typename H, typename... R> struct Pack<H, R...> { H h; Pack<R...> r; };
template <typename Sig> struct BodyTraits;
template <typename... P> struct BodyTraits<void (*)(P...)> { using pack = Pack<P...>; };

// Pointers that arrive through the pack are device-memory addresses; say so (address space 1), as the compiler does by itself
// for pointer kernel arguments: otherwise every access through them is a flat access (no scalar loads of uniform data,
// no global_load addressing modes) -- measured: FIR 0.26 -> 0.50 ms, PLL phase 0.97 -> 1.19 ms.
template <typename H> __device__ __forceinline__ H as_global(H v)
{
    if constexpr (std::is_pointer<H>::value) {
        using E = typename std::remove_pointer<H>::type;
        __attribute__((address_space(1))) E *g = (__attribute__((address_space(1))) E *)v;
        asm("" : "+s"(g));      // opaque (and still uniform): the optimizer would fold the cast pair away
        return (H)g;
    } else {
        return v;
    }
}
__device__ __forceinline__ IqSrc as_global(IqSrc v)
{
    v.p = as_global(v.p);
    return v;
}
template <typename T> __device__ __forceinline__ AgcParams<T> as_global(AgcParams<T> v)
{
    v.raw_out = as_global(v.raw_out);
    return v;
}

template <auto Body, typename... Done>
__device__ __forceinline__ void call_body(const Pack<> &, Done... d) { Body(d...); }
template <auto Body, typename H, typename... R, typename... Done>
__device__ __forceinline__ void call_body(const Pack<H, R...> &p, Done... d) { call_body<Body>(p.r, d..., as_global(p.h)); }

template <auto Body, int TB>
__global__ void __launch_bounds__(TB) k_run(const typename BodyTraits<decltype(Body)>::pack *__restrict__ packs)
{
    call_body<Body>(packs[blockIdx.z]);
}

inline void pack_fill(Pack<> &) {}
template <typename H, typename... R, typename A0, typename... AR> void pack_fill(Pack<H, R...> &p, A0 &&a0, AR &&...ar)
{
    p.h = (H)a0;
    pack_fill(p.r, ar...);
}

typedef void (*GoFn)(dim3, dim3, size_t, hipStream_t, const void *);
enum { OP_LAUNCH, OP_MEMSET, OP_FORK, OP_JOIN_RECORD, OP_JOIN_WAIT, OP_TBEGIN, OP_TEND, OP_TGAP, OP_D2H, OP_H2D, OP_EV0, OP_EV1 };
struct PlanOp {
    int op = OP_LAUNCH, side = 0;
    GoFn go = nullptr;
    dim3 grid, block;
    size_t shmem = 0, pack_off = 0, pack_size = 0;
    void *dst = nullptr;
    const void *src = nullptr;
    int value = 0;
    size_t bytes = 0;
    const char *name = nullptr;
};
struct Plan {
    std::vector<PlanOp> ops;
    std::vector<unsigned char> packs;
    hipStream_t side_stream = nullptr;             // the context's second stream (launch sites name streams, the plan keeps sides)
    void clear() { ops.clear(); packs.clear(); }
    int side_of(hipStream_t s) const { return (side_stream && s == side_stream) ? 1 : 0; }
    template <auto Body, int TB, typename... A> void launch(const char *kname, dim3 grid, dim3 block, size_t shmem, int side, A &&...args)
    {
        using PackT = typename BodyTraits<decltype(Body)>::pack;
        static_assert(std::is_trivially_copyable<PackT>::value, "kernel arguments travel as plain bytes");
        PackT pk;
        memset((void *)&pk, 0, sizeof pk);
        pack_fill(pk, args...);
        PlanOp o;
        o.op = OP_LAUNCH;
        o.name = kname;
        o.side = side;
        o.grid = grid;
        o.block = block;
        o.shmem = shmem;
        o.pack_off = (packs.size() + 15) & ~(size_t)15;
        o.pack_size = sizeof(PackT);               // the stride k_run indexes the batch's packs with
        static_assert(alignof(PackT) <= 16, "pack alignment");
        packs.resize(o.pack_off + o.pack_size, 0);
        memcpy(packs.data() + o.pack_off, &pk, sizeof pk);
        o.go = [](dim3 g, dim3 b, size_t sh, hipStream_t st, const void *dp) {
            hipLaunchKernelGGL((k_run<Body, TB>), g, b, sh, st, (const PackT *)dp);
        };
        ops.push_back(o);
    }
    void simple(int op, int side = 0, const char *name = nullptr)
    {
        PlanOp o;
        o.op = op; o.side = side; o.name = name;
        ops.push_back(o);
    }
    void memset_async(void *dst, int value, size_t bytes, int side = 0)
    {
        PlanOp o;
        o.op = OP_MEMSET; o.side = side; o.dst = dst; o.value = value; o.bytes = bytes;
        ops.push_back(o);
    }
    void copy(int op, void *dst, const void *src, size_t bytes)
    {
        PlanOp o;
        o.op = op; o.dst = dst; o.src = src; o.bytes = bytes;
        ops.push_back(o);
    }
    // two plans can share their launches when they are the same sequence of operations with the same kernels, block
    // shapes and LDS sizes (grids may differ: the larger one is launched and every body checks its own bounds)
    bool same_shape(const Plan &o) const
    {
        if (ops.size() != o.ops.size() || packs.size() != o.packs.size()) return false;
        for (size_t i = 0; i < ops.size(); i++) {
            const PlanOp &a = ops[i], &b = o.ops[i];
            if (a.op != b.op || a.side != b.side || a.go != b.go || a.block.x != b.block.x || a.block.y != b.block.y ||
                a.shmem != b.shmem || a.pack_off != b.pack_off || a.pack_size != b.pack_size)
                return false;
        }
        return true;
    }
};
#define PDT_LAUNCH(TB, KERNEL, grid, block, shmem, stream, ...) \
    PL.launch<&KERNEL, TB>(#KERNEL, grid, block, shmem, PL.side_of(stream), __VA_ARGS__)

}  // namespace

// Developer switches (A/B runs of older kernel variants, tuning sweeps).  The library never reads the environment: the
// switches come from a process-wide registry that only the TEST-ONLY entry pdt_dev_set fills (include/pdt_dev.h; the Python
// binding used by tests/ and bench.py mirrors the PDT_* environment variables into it), and a context takes its copy ONCE,
// when it is opened.
static std::mutex g_dev_mu;
static std::map<std::string, std::string> g_dev;
struct Tuning {
    double band_pad = 0.0, pll_warm_scale = 1.0, head_taus = 0.0, agc_k = 0.0, pll_warm_s = 0.0, agc_warm_s = 0.0;
    double overlap_split[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    long long hbm_limit_mb = 0, window_piece = 0;
    int scout_syms = 0, gspan = 0, gspan_cap = 0, ingest_threads = 0, ingest_span_mb = 0, ingest_streams = 0, overlap_segments = 0, overlap_min_mb = 0, fir_wg_per_cu = 0, agc_tpb = 0, gseg = 0, pll_block = 0, fix_passes = 2;
    bool fir_generic = false, mix_unfused = false, quality_inline = false, gemit_groups = false, agc_unfused = false, no_excl = false, gardner_onebuf = false, gardner_noring = false, gardner_sequential = false, seg_sequential = false, agc_lanes = false, overlap = true, debug_overlap = false, chain_one_range = false, ema_noguess = false, debug_sync = false, pll_noshort = false, pll_nockpt = false, pll_noconsensus = false, seg_plain = false, sync_block = false, gardner_nostride = false;
    void load()
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
        if (const char *e = get("PDT_OVERLAP_SPLIT")) {          // "0.64,0.22,0.14": the segments' fractions of the capture
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
        seg_plain = get("PDT_SEG_PLAIN") != nullptr;
        sync_block = get("PDT_SYNC_BLOCK") != nullptr;
    }
};

// A stream is demodulated segment by segment (whole reference chunks).  Between segments every stage's exact state is
// carried here -- T values as doubles (exact for float and double) -- and the device keeps a bounded window of the input
// and of the few streams a later segment looks back on.
struct StreamCarry {
    bool active = false;          // run_capture works on a window of a stream
    bool final_seg = false;       // the stream ends with this segment (short last chunk, partial frame reported)
    bool in_place = false;        // the whole capture has its place in the window (pdt_demod_fd of a large file): never slides
    uint64_t place_align = 0;     // in place: the grid the window's origin stays on (0 = stream_align)
    bool quality = false;         // in place: the segments keep the per-chunk reports (pdt_keep_quality; chunk-aligned cuts)
    long long first = 0;          // local index of the first new input sample (a multiple of the chunk)
    uint64_t origin = 0;          // global sample index of local sample 0 (a multiple of lcm(chunk, FIR ring length))
    // StaticGain / AGC
    bool have_norm = false;
    double norm_factor = 0, gain = 0;
    // PLL
    bool locked = false;
    double phase = 0, freq = 0, avg = 0, locksig = 0, sweep = 0;
    int64_t lock_sample = -1;     // global
    double lock_freq_hz = 0, avg_at_lock = 0;
    // symbol sampler (Gardner: nextSample, prev, halfSample; M&M: nextSample, stepSize, sampleLast)
    bool have_sampler = false;
    double sa = 0, sb = 0, sc = 0;
    // Manchester
    double sym_m2 = 0, sym_m1 = 0;
    unsigned clockmod = 0;
    uint64_t nsym_total = 0;
    // byte sync: the last bits (from the sync word of a frame still open, else the last len-1), their time sources
    std::vector<unsigned char> kept_bits;
    std::vector<long long> kept_src;           // global interpolated-sample index per kept bit
    uint64_t bit_base = 0;                     // global index of kept_bits[0]
    uint64_t nbits_total = 0;
    long long next_free = 0;                   // global bit index before which no new frame may open
    bool have_pending = false;                 // an incomplete frame at the end of the last segment
    pdt_frame pending;
    // per segment, filled by run_capture for the stream code
    std::vector<pdt_frame> seg_frames;
    uint64_t seg_new_symbols = 0, seg_new_bits = 0;
};

struct pdt_ctx {
    pdt_config cfg;
    pdt_loop_params lp = {};     // pdt_set_loop_params: 0 = the mains' constant
    Tuning tune;
    StreamCarry sc;
    int elem;                 // sizeof(DT)
    uint32_t interp, ntaps;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_fork = nullptr, ev_join = nullptr;
    hipStream_t stream2 = nullptr;     // side stream: block-parallel PLL phase runs beside the sequential acquisition

    DevBuf pcm, pll, lock, fir, agc, sym, symidx, bits, bitsym, hits, frames, taps, mag, seams_pll, seams_agc, scal, lockinfo, term, seams_ema, gtable, gentries, gcand, gmfirst, stiles, gsegmap, gsegstart, gbands, gclist, gneed, gchain, gspan_keys, gspan_tails, gspan_rows, gspan_items, gspan_ctl, gspan_recs, gcentries, gflags, agc_maps, pll_head, taps_rot, pll_scratch, tip, sync_scr, agc_raw, agc_ckpt, pll_ckpt;
    bool counted = false;        // this context is in g_open_contexts
    bool keep_agc_raw = false;   // pdt_keep_presquelch: also keep the AGC output before Squelch (stage PDT_ST_AGC_RAW)
    // pdt_keep_quality: the averagePhase stream (what CarrierTrackPLL returns, chunk by chunk) and the per-chunk counts
    bool keep_quality = false;
    bool keep_pll_asked = false; // pdt_keep_pll(ctx, 1) was called: the caller reads the PLL stream -- one piece, no overlapped segments
    bool keep_pll = true;        // pdt_keep_pll: the PLL output stream (stage PDT_ST_PLL) is written out although only the filter reads it
    DevBuf avgph, term_ap, seams_q, chunkinfo;
    // pdt_stage_pll: the next run starts the PLL from this state, keeps the lock and averagePhase streams and stops after the PLL
    struct PllInject {
        bool active = false, started = false, locked = false;
        double phase = 0, freq = 0, avg = 0, locksig = 0, sweep = 0;
    } inj;
    long long last_pll_block = 0;       // PLL block length of the last run (where the end state sits in seams_pll)
    void *qual_pin = nullptr;
    size_t qual_pin_cap = 0;
    uint64_t pend_chunks = 0;           // ChunkInfo records in flight (0 = none asked for)
    std::vector<pdt::ChunkInfo> chunk_host;      // per chunk of the capture, counts cumulative from its first sample
    uint64_t report_samples = 0;        // length of the capture the reports describe
    pdt_progress_fn progress_fn = nullptr;
    void *progress_user = nullptr;
    // (stream_in is declared with the streaming state below)
    long long gcand_key = -1;          // (chunk_out, step) the candidate list on the device was built for
    int gardner_mode = 0;              // 0 sequential, 1 state table (last run)
    const void *pcm_dev = nullptr;     // input actually used (own copy or caller's buffer)
    int pcm_fmt = 0;                   // 0 = int16 pairs, 1 = float32 pairs

    std::vector<unsigned char> taps_host;
    // results
    uint64_t n_samples = 0, n_out = 0;
    std::vector<pdt_frame> frames_host;
    std::vector<pdt_tip_frame> tip_host;
    uint32_t frames_on_device = 0;      // FrameRec records of the last demodulation still in ctx->frames
    bool have_frames = false;           // a demodulation (or stage-level byte sync) has run
    // streaming front end: everything received so far (device), what has been reported
    Plan plan;                          // the operations of the demodulation call being issued
    DevBuf packs_dev;                   // argument packs of the plan(s) being executed (this context leads the batch)
    void *packs_pin = nullptr;          // pinned staging of the same
    size_t packs_pin_cap = 0;
    pdt_ctx *leader = nullptr;          // context whose streams / events carried the last execution
    int batch_hint = 1;                 // captures demodulated together with this one (sizes the block-parallel geometry)
    // host -> HBM ingest of a capture (file or memory): pinned slots filled by a few host threads, copies on a stream of their own
    void *ingest_pin = nullptr;
    size_t ingest_pin_cap = 0;
    hipStream_t copy_stream = nullptr, copy_streams_more[3] = { nullptr, nullptr, nullptr };   // span copies go round robin over them
    hipEvent_t ev_ingest = nullptr, ev_ingest_more[3] = { nullptr, nullptr, nullptr };
    std::vector<hipEvent_t> ingest_ev, span_ev;      // per pinned slot; per span (overlapped ingest)
    double stream_gpu_ms = 0;
    double ingest_ms = 0;               // host wall time of the last ingest (issue of the last copy)
    DevBuf stream_in, seg_dev, lt_theta, lt_phi;          // input window of the stream; small device block for the segment's carried-out state
    uint64_t stream_have = 0, stream_done = 0;   // samples in the window / of them already demodulated (local indices)
    uint64_t stream_total = 0;          // samples pushed since pdt_stream_begin
    int stream_fmt = -1;                // -1 = no push yet, 0 = pcm16, 1 = float32
    bool stream_open = false;           // between the first push and pdt_stream_end / _begin: the stage buffers hold the tails the next push continues from
    std::vector<pdt_frame> stream_new;
    unsigned char *seg_pin = nullptr;   // pinned staging for the small per-segment transfers (part of the pend_sc block)
    pdt_stats stats;
    std::vector<pdt_kernel_time> ktimes;
    std::vector<KTimer> timers;
    std::vector<hipEvent_t> event_pool;
    void *pinned = nullptr;             // pinned staging buffer for the frame records
    size_t pinned_cap = 0;
    uint32_t last_nframes = 0;
    // results in flight between the enqueue and the finish phase of a capture
    DevScalars *pend_sc = nullptr;      // both in one small pinned block (pageable targets would make the
    unsigned char *pend_info = nullptr; // "asynchronous" read-back copies wait for the stream)
    uint32_t pend_got_frames = 0;
    uint64_t pend_n = 0;
    bool pending = false;
    uint64_t stage_len[PDT_ST_COUNT];
    TimeAxis<float> axis_f;
    TimeAxis<double> axis_d;
};

namespace {

// ---------------------------------------------------------------- FIR taps (LowPassFilter.c:127-175)
// Evaluated on the host with the operations the reference performs (sinf / sin of the sinc argument, cos in the Blackman
// window) -- through this library's own restatements of those C-library functions (pdt_device_math.h), so that the taps do
// not depend on the libm of the machine the library runs on.
template <typename T> void make_lpf(T *h, int N, T Fc, T Fs, int interp)
{
    const T Tt = (T)(1.0 / (double)Fs);
    const T wc = (T)(2.0 * M_PI * (double)Fc * (double)Tt);
    const T tou = (T)((N - 1.0) / 2.0);
    for (int n = 0; n < N; n++) {
        const T arg = wc * ((T)n - tou);
        T sv;
        if (sizeof(T) == 4) {
            float sf, cf;
            sincosf_glibc((float)arg, sf, cf);          // sinf: the sine half of glibc's shared sinf / sincosf evaluation
            sv = (T)sf;
        } else {
            sv = (T)sin_glibc((double)arg);
        }
        T hd = (T)((double)sv / (M_PI * (double)((T)n - tou)));
        if (((T)n == tou) && ((N / 2) * 2 != N)) hd = (T)((double)wc / M_PI);
        const T wn = (T)(0.42 - 0.5 * cos_glibc((2 * M_PI * n) / (N - 1)) + 0.08 * cos_glibc((4 * M_PI * n) / (N - 1)));
        h[n] = hd * wn * (T)interp;
    }
}

int poes_interp(uint32_t rate) { return (int)rint(150000.0 / (double)(float)rate); }   // POESTIPdemod/main.c:347

class Launcher {
  public:
    Launcher(pdt_ctx *c) : ctx(c) {}
    // profile mode: one event per group boundary -- a group that starts right where the previous one ended on
    // the same stream shares that event (every recorded event is a small gap in the stream)
    void begin(const char *name, hipStream_t s = nullptr)
    {
        if (!ctx->cfg.profile) return;
        KTimer t;
        t.name = name;
        cur = s ? s : ctx->stream;
        if (have_last && last_stream == cur) {
            t.a = last_b;
            t.shared_a = true;
        } else {
            t.a = take();
            t.shared_a = false;
            (void)hipEventRecord(t.a, cur);
        }
        t.b = take();
        ctx->timers.push_back(t);
        open_idx = ctx->timers.size() - 1;
        have_last = false;
    }
    void end()
    {
        if (!ctx->cfg.profile) return;
        (void)hipEventRecord(ctx->timers[open_idx].b, cur);
        last_b = ctx->timers[open_idx].b;
        last_stream = cur;
        have_last = true;
    }
    // work enqueued outside any group (copies, memsets, stream waits) breaks the sharing
    void gap() { have_last = false; }

  private:
    hipEvent_t take()
    {
        hipEvent_t e;
        if (!ctx->event_pool.empty()) { e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    }
    pdt_ctx *ctx;
    hipStream_t cur = nullptr, last_stream = nullptr;
    hipEvent_t last_b = nullptr;
    bool have_last = false;
    size_t open_idx = 0;
};

// run_capture records its timer groups into the plan; the Launcher above turns them into events when the plan runs
struct PlanGroups {
    Plan &pl;
    bool on;
    void begin(const char *name, hipStream_t s = nullptr) { if (on) pl.simple(OP_TBEGIN, pl.side_of(s), name); }
    void end() { if (on) pl.simple(OP_TEND); }
    void gap() { if (on) pl.simple(OP_TGAP); }
};


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

template <typename T> PllParams<T> make_pll_params(const pdt_ctx *ctx)
{
    // call-site constants: POESTIPdemod/main.c:32-46,413 / ARGOSdemod/main.c:33-44,265
    PllParams<T> P;
    const T Fs = (T)ctx->cfg.sample_rate;
    const bool argos = ctx->cfg.mode == PDT_MODE_ARGOS;
    const bool live = !argos && ctx->cfg.chain == PDT_CHAIN_LIVE;        // POESTIPdemodPortAudio/main.c:41-57
    const pdt_loop_params &lp = ctx->lp;                                  // (what the caller's CarrierTrackPLL would have been handed)
    const T freqRange = lp.pll_freq_range_hz != 0 ? (T)lp.pll_freq_range_hz : argos ? (T)550.0 : (T)4500.0;
    const double w = 2.0 * M_PI / (double)Fs;
    const T bw_acq = lp.pll_loopbw_acq != 0 ? (T)lp.pll_loopbw_acq : (T)((argos ? 16.0 : live ? 198.9437 : 127.3240) * w);
    const T bw_trk = lp.pll_loopbw_track != 0 ? (T)lp.pll_loopbw_track : (T)((argos ? 16.0 : 10.3451) * w);
    P.Fs = Fs;
    P.lock_thr = (lp.pll_lock_threshold != 0 || (lp.zero_mask & PDT_LP_ZERO_LOCK_THRESHOLD)) ? (T)lp.pll_lock_threshold : argos ? (T)0.1 : live ? (T)0.10 : (T)0.08;
    P.lock_alpha = lp.pll_lock_alpha != 0 ? (T)lp.pll_lock_alpha : (T)((argos ? 3.1831 : 0.3979) * w);
    const T damp = (T)0.999;
    const T four = 4, one = 1, two = 2;
    P.alpha_acq = (four * damp * bw_acq) / (one + two * damp * bw_acq + bw_acq * bw_acq);     // :90-91, all DT
    P.beta_acq = (four * bw_acq * bw_acq) / (one + two * damp * bw_acq + bw_acq * bw_acq);
    const double dd = (double)damp, db = (double)bw_trk;
    P.alpha_trk = (T)((4.0 * dd * db) / (1.0 + 2.0 * dd * db + (double)(bw_trk * bw_trk)));   // :272-273, double
    P.beta_trk = (T)((4.0 * db * db) / (1.0 + 2.0 * dd * db + (double)(bw_trk * bw_trk)));
    {
        const T bw_w = bw_acq * (T)8;
        P.alpha_wide = (four * damp * bw_w) / (one + two * damp * bw_w + bw_w * bw_w);
        P.beta_wide = (four * bw_w * bw_w) / (one + two * damp * bw_w + bw_w * bw_w);
    }
    P.max_freq = (T)(2.0 * M_PI * (double)freqRange / (double)Fs);
    P.min_freq = (T)(-2.0 * M_PI * (double)freqRange / (double)Fs);
    {
        // sweep gate |pi/2 - averagePhase| < 0.05 (CarrierTrackingPLL.c:236, evaluated as there: the difference
        // narrowed to DT, fabs, compared as double) is a monotone function of averagePhase on either side of
        // pi/2: bisect the two edges over the ordered bit patterns so that the kernels need two compares only
        auto gate = [](T av) { return (double)std::fabs((T)(M_PI / 2.0 - (double)av)) < 0.05; };
        typedef typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type U;
        auto bits = [](T v) { U u; memcpy(&u, &v, sizeof u); return u; };
        auto val = [](U u) { T v; memcpy(&v, &u, sizeof v); return v; };
        const T mid = (T)(M_PI / 2.0);
        U in_lo = bits(mid), out_lo = bits((T)1.0);          // gate(mid) true, gate(1.0) false
        while (in_lo - out_lo > 1) {
            const U m = out_lo + (in_lo - out_lo) / 2;
            if (gate(val(m))) in_lo = m; else out_lo = m;
        }
        U in_hi = bits(mid), out_hi = bits((T)2.5);
        while (out_hi - in_hi > 1) {
            const U m = in_hi + (out_hi - in_hi) / 2;
            if (gate(val(m))) in_hi = m; else out_hi = m;
        }
        P.cond_lo = val(in_lo);
        P.cond_hi = val(in_hi);
    }
    P.sweep0 = (T)(0.2 * (2.0 * M_PI / (double)Fs));
    P.avg0 = (T)(M_PI / 2.0);
    P.phase0 = (T)0.1;
    P.freq0 = 0;
    P.locksig0 = 0;
    P.i0 = 0;
    P.want_lock = (argos || live) ? 1 : 0;
    return P;
}

uint32_t next_pow2(uint32_t v)
{
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

SyncParams make_sync_params(bool argos, bool argos_twin = false)
{
    SyncParams SP;
    if (argos) {
        SP.pattern = 0x02F0ull;   // "0001011110000"
        SP.len = 13; SP.allow_inverse = 0; SP.span = 56; SP.first_bits = 8; SP.nbytes = 7; SP.prefix = 0;
        if (argos_twin) SP.allow_inverse = 1;                 // ARGOSdemodPortAudio/ByteSync.c:112: the inverse word is looked for too
    } else {
        SP.pattern = 0x76F10ull;  // "1110110111100010000"
        SP.len = 19; SP.allow_inverse = 1; SP.span = 813; SP.first_bits = 5; SP.nbytes = 102; SP.prefix = 2;
    }
    return SP;
}

void launch_bytesync(pdt_ctx *ctx, Plan &PL, hipStream_t st, const SyncParams &SP, DevScalars *d_sc, long long bit_cap, uint32_t hit_cap,
                     uint32_t frame_cap, long long min_pos = 0)
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
static void chunk_reports_range(const pdt_ctx *ctx, uint64_t c0, uint64_t c1, uint64_t total, pdt_chunk_report *out, const pdt_frame *open_frame)
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


// What run_capture's recording phase hands to its finish phase (finish_capture): the capacities and flags the read-back needs.
struct FinishArgs {
    bool argos, need_lock, fuse_mix;
    long long N, n_out, chunk, chunk_out, first, first_out, sym_cap;
    int interp, ntaps;
    uint32_t hit_cap, frame_cap;
    SyncParams SP;
};

// The finish phase of a demodulation call (RUN_ALL / RUN_FINISH): wait for the launches, read the scalars, the lock record and
// the frame records back, put the host-side time stamps on the frames, fill the statistics; for
// a stream segment, carry every stage's state to the next one.  (Round 5: split off run_capture.)
template <typename T> int finish_capture(pdt_ctx *ctx, uint64_t n, const FinishArgs &FA)
{
    const bool argos = FA.argos, need_lock = FA.need_lock, fuse_mix = FA.fuse_mix;
    const long long N = FA.N, n_out = FA.n_out, chunk = FA.chunk, chunk_out = FA.chunk_out, first = FA.first, first_out = FA.first_out,
                    sym_cap = FA.sym_cap;
    const int interp = FA.interp, ntaps = FA.ntaps;
    const uint32_t hit_cap = FA.hit_cap, frame_cap = FA.frame_cap;
    const SyncParams &SP = FA.SP;
    const T Fs = (T)ctx->cfg.sample_rate;
    StreamCarry *seg = ctx->sc.active ? &ctx->sc : nullptr;
    FrameRec *d_frames = (FrameRec *)ctx->frames.p;
    (void)first_out;
    if (!ctx->pending || ctx->pend_n != n) return PDT_ERR_STATE;
    ctx->pending = false;
    pdt_ctx *lead = ctx->leader ? ctx->leader : ctx;
    if (seg && seg->in_place && !ctx->tune.sync_block) {
        // the overlapped ingest: wait for the segment WITHOUT sitting inside the runtime -- the ingest's submitter thread is
        // queueing copies all the while (a blocking hipStreamSynchronize here held it up: the copies stopped for as long as a
        // segment's kernels ran, tools/jobs/r5_e2e_ab.sh)
        for (;;) {
            const hipError_t e = hipEventQuery(lead->ev1);
            if (e == hipSuccess) break;
            if (e != hipErrorNotReady) { HIP_TRY(e); }
            (void)hipGetLastError();
            std::this_thread::sleep_for(std::chrono::microseconds(40));
        }
    }
    HIP_TRY(hipStreamSynchronize(lead->stream));
    const DevScalars &sc = *ctx->pend_sc;
    PllLockInfo<T> info;
    memcpy(&info, ctx->pend_info, sizeof info);
    const uint32_t got_frames = ctx->pend_got_frames;
    if (sc.nframes > frame_cap || sc.nhits > hit_cap || (long long)sc.nsym > sym_cap) {
        fprintf(stderr, "libpdt: internal capacity exceeded (frames %u/%u hits %u/%u symbols %llu/%lld)\n", sc.nframes,
                frame_cap, sc.nhits, hit_cap, sc.nsym, sym_cap);
        return PDT_ERR_STATE;
    }
    std::vector<FrameRec> recs(sc.nframes);
    if (sc.nframes) {
        const uint32_t have = std::min<uint32_t>(sc.nframes, got_frames);
        if (have) memcpy(recs.data(), ctx->pinned, (size_t)have * sizeof(FrameRec));
        if (sc.nframes > have)
            HIP_TRY(hipMemcpy(recs.data() + have, d_frames + have, (size_t)(sc.nframes - have) * sizeof(FrameRec), hipMemcpyDeviceToHost));
    }
    ctx->last_nframes = sc.nframes;
    const uint64_t got_chunks = ctx->pend_chunks;     // (a segment: those of its new chunks, window-local counts -- see below)
    if (!seg) {
        ctx->chunk_host.clear();
        ctx->report_samples = n;
        if (got_chunks) {
            ctx->chunk_host.resize((size_t)got_chunks);
            memcpy(ctx->chunk_host.data(), ctx->qual_pin, (size_t)got_chunks * sizeof(ChunkInfo));
        }
    }
    ctx->pend_chunks = 0;

    float ms = 0;
    (void)hipEventElapsedTime(&ms, lead->ev0, lead->ev1);

    // time stamp of the bit whose symbol was taken at global interpolated-sample index g, in a capture of n_all samples
    // 
    auto frame_time = [&](long long g, long long n_all) -> double {
        if (argos && sizeof(T) == 4) return (double)ctx->axis_f.at((uint64_t)g + 1);     // the twin: float stamps (pdt_open)
        if (argos) return ctx->axis_d.at((uint64_t)g + 1);                   // waveDataTime[i] = (i+1)-th partial sum
        const long long c = g / chunk_out, rr = g % chunk_out;
        const long long j = rr / interp + 1;                                  // Q2: time of the *next* input sample
        const long long ns_c = std::min<long long>(chunk, n_all - c * chunk);
        if (j < ns_c) return (double)ctx->axis_f.at((uint64_t)(c * chunk + j + 1));
        if (ns_c == chunk || c == 0) return 0.0;                              // one past the array: never-written zero
        return (double)ctx->axis_f.at((uint64_t)((c - 1) * chunk + j + 1));   // stale value of the previous chunk
    };

    // ---- per-kernel timings (profile mode)
    auto collect_timers = [&]() {
    ctx->ktimes.clear();
    for (auto &t : ctx->timers) {
        float tms = 0;
        (void)hipEventElapsedTime(&tms, t.a, t.b);
        bool found = false;
        for (auto &k : ctx->ktimes)
            if (t.name == k.name) {
                k.launches++;
                k.total_ms += tms;
                found = true;
            }
        if (!found) {
            pdt_kernel_time k;
            memset(&k, 0, sizeof k);
            snprintf(k.name, sizeof k.name, "%s", t.name.c_str());
            k.launches = 1;
            k.total_ms = tms;
            ctx->ktimes.push_back(k);
        }
        if (!t.shared_a) ctx->event_pool.push_back(t.a);
        ctx->event_pool.push_back(t.b);
    }
    ctx->timers.clear();
    };

    if (seg) {
        // ---- stream segment: hand the new frames to the stream code, carry every stage's state to the next segment
        SegTail<T> tail;
        memcpy(&tail, ctx->seg_pin + 4096, sizeof tail);
        const long long org_out = (long long)seg->origin * interp;
        const long long n_all = (long long)seg->origin + N;                   // samples of the stream so far
        const long long kept = (long long)seg->kept_bits.size();
        const long long sym_pad = 2 + (long long)(seg->nsym_total & 1u);
        const long long nbits_loc = (long long)sc.nbits;
        if ((long long)sc.nsym < sym_pad || nbits_loc < kept) return PDT_ERR_STATE;
        const uint64_t new_syms = sc.nsym - (uint64_t)sym_pad, new_bits = (uint64_t)(nbits_loc - kept);
        // PLL
        if (!seg->locked && info.lock_sample >= 0) {
            seg->locked = true;
            seg->lock_sample = info.lock_sample + (long long)seg->origin;
            seg->lock_freq_hz = (double)(info.freq_at_lock * Fs) / (2.0 * M_PI);
            seg->avg_at_lock = (double)info.avg_at_lock;
        }
        if (N > first || first == 0) {
            if (seg->locked && N > 0) {
                seg->phase = (double)tail.pll_phase;
                seg->freq = (double)tail.pll_freq;
                if (info.lock_sample == N - 1) {                              // locked on the very last sample: nothing walked after it
                    seg->phase = (double)info.st.phase;
                    seg->freq = (double)info.st.freq;
                }
                seg->locksig = need_lock ? (double)tail.locksig : (double)info.st.locksig;
                seg->avg = (double)info.st.avg_phase;
                seg->sweep = (double)info.st.sweep;
            } else if (N > 0) {
                seg->phase = (double)info.st.phase; seg->freq = (double)info.st.freq; seg->avg = (double)info.st.avg_phase;
                seg->locksig = (double)info.st.locksig; seg->sweep = (double)info.st.sweep;
            }
        }
        // StaticGain / AGC
        if (!seg->have_norm && N > 0) {
            T nv;
            memcpy(&nv, &sc.norm, sizeof(T));
            seg->norm_factor = (double)nv;
            seg->gain = (double)nv;
            seg->have_norm = true;
        }
        if (n_out - first_out > 0) seg->gain = (double)tail.agc_gain;
        // sampler, Manchester
        seg->sa = (double)tail.sampler.a; seg->sb = (double)tail.sampler.b; seg->sc = (double)tail.sampler.c;
        seg->have_sampler = true;
        if (sc.nsym >= 1) { seg->sym_m2 = (double)tail.sym_m2; seg->sym_m1 = (double)tail.sym_m1; }
        seg->clockmod = tail.clock;
        if (got_chunks) {
            // the reports of this segment's chunks: the window's counts include the history symbols and the kept bits (all in
            // front of the first new chunk), the capture's counts those of the earlier segments
            const ChunkInfo *ci = (const ChunkInfo *)ctx->qual_pin;
            for (uint64_t c = 0; c < got_chunks; c++) {
                ChunkInfo o = ci[c];
                if ((long long)o.sym_upto < sym_pad || (long long)o.bits_upto < kept) return PDT_ERR_STATE;
                o.sym_upto = o.sym_upto - (uint64_t)sym_pad + seg->nsym_total;
                o.bits_upto = o.bits_upto - (uint64_t)kept + seg->nbits_total;
                o.t0_src += org_out;
                ctx->chunk_host.push_back(o);
            }
            // averagePhase goes on behind the lock (the lock record holds its value AT the lock): the next segment's walkers
            // start from the value behind this segment's last sample = its last chunk's
            if (seg->locked && N > first) seg->avg = ci[got_chunks - 1].avg_phase;
        }
        seg->nsym_total += new_syms;
        seg->nbits_total += new_bits;
        seg->seg_new_symbols = new_syms;
        seg->seg_new_bits = new_bits;
        // bits the device kept: local indices [b0, nbits_loc)
        const long long b0 = nbits_loc - (long long)tail.nkeep;
        auto src_of = [&](long long pos) -> long long {                      // global interpolated-sample index behind local bit pos
            if (pos < kept) return seg->kept_src[(size_t)pos];
            return tail.src[pos - b0] + org_out;
        };
        // frames
        seg->seg_frames.clear();
        long long open_pos = -1;
        bool pend = false;
        for (unsigned f = 0; f < sc.nframes; f++) {
            const FrameRec &r = recs[f];
            if (!r.complete && !seg->final_seg) { open_pos = r.bit_index; }
            pdt_frame o;
            memset(&o, 0, sizeof o);
            o.bit_index = r.bit_index + (long long)seg->bit_base;
            o.inverted = argos ? 0 : r.inverted;      // (the ARGOS twin re-inverts such a packet's bits but stamps it like any other)
            o.nbytes = r.nbytes;
            o.complete = r.complete;
            memcpy(o.bytes, r.bytes, 104);
            const long long g = (r.bit_index < kept) ? seg->kept_src[(size_t)r.bit_index] : r.time_src + org_out;
            o.time_src = g;
            o.time = frame_time(g, n_all);
            if (!r.complete && !seg->final_seg) {
                seg->pending = o;                                             // reported when it completes (or at the stream's end)
                pend = true;
                break;
            }
            seg->seg_frames.push_back(o);
            seg->next_free = std::max<long long>(seg->next_free, o.bit_index + (long long)SP.span);
        }
        seg->have_pending = pend;
        // what the next segment sees of these bits: from the sync word of the open frame, else the last len - 1
        long long keep_from = std::max<long long>(0, nbits_loc - (long long)(SP.len - 1));
        if (open_pos >= 0) keep_from = std::max<long long>(0, open_pos - (long long)(SP.len - 1));
        if (keep_from < b0) return PDT_ERR_STATE;                             // (an open frame is shorter than the kept tail)
        std::vector<unsigned char> nb((size_t)(nbits_loc - keep_from));
        std::vector<long long> ns((size_t)(nbits_loc - keep_from));
        for (long long q = keep_from; q < nbits_loc; q++) {
            nb[(size_t)(q - keep_from)] = tail.bits[q - b0];
            ns[(size_t)(q - keep_from)] = src_of(q);
        }
        seg->kept_bits.swap(nb);
        seg->kept_src.swap(ns);
        seg->bit_base += (uint64_t)keep_from;
        ctx->stats.gpu_ms = ms;
        // pdt_read_stage after a push: the segment's own arrays, window-local (PLL / FIR / AGC from the window's origin; symbols
        // behind the two or three history symbols; bits behind the kept ones) -- a debugging aid, see tools/probes/stream_debug.py
        ctx->stage_len[PDT_ST_PLL] = (uint64_t)N;
        ctx->stage_len[PDT_ST_LOCK] = need_lock ? (uint64_t)N : 0;
        ctx->stage_len[PDT_ST_FIR] = ctx->stage_len[PDT_ST_AGC] = (uint64_t)n_out;
        ctx->stage_len[PDT_ST_AGC_RAW] = 0;
        ctx->stage_len[PDT_ST_SYM] = ctx->stage_len[PDT_ST_SYMIDX] = sc.nsym;
        ctx->stage_len[PDT_ST_BITS] = ctx->stage_len[PDT_ST_BITSYM] = sc.nbits;
        collect_timers();                     // (profile mode: the kernel groups of the LAST segment)
        return PDT_OK;
    }

    T norm_val;
    memcpy(&norm_val, &sc.norm, sizeof(T));
    pdt_stats &S = ctx->stats;
    memset(&S, 0, sizeof S);
    S.samples = n;
    S.out_samples = (uint64_t)n_out;
    S.symbols = sc.nsym;
    S.bits = sc.nbits;
    S.frames = sc.nframes;
    S.lock_sample = info.lock_sample;
    S.lock_freq_hz = (double)(info.freq_at_lock * Fs) / (2.0 * M_PI);         // CarrierTrackingPLL.c:269
    S.avg_phase = (double)info.avg_at_lock;
    S.norm_factor = (double)norm_val;
    S.interp = (uint32_t)interp;
    S.ntaps = (uint32_t)ntaps;
    S.pll_blocks = sc.counters[0];
    S.pll_seam_fixes = sc.counters[1];
    S.agc_blocks = sc.counters[2];
    S.agc_seam_fixes = sc.counters[3];
    S.gpu_ms = ms;
    S.gardner_parallel = (uint32_t)ctx->gardner_mode;
    const bool tabled = ctx->gardner_mode != 0;
    S.gardner_walked = tabled ? sc.gstats[2] : 0u;
    S.gardner_full_domain = tabled ? sc.gstats[1] : 0u;
    S.gardner_candidates = tabled ? sc.gstats[3] : 0u;
    S.sync_overflow = sc.sync_overflow;
    S.segments = 1;

    ctx->stage_len[PDT_ST_PLL] = (fuse_mix && !ctx->keep_pll) ? 0 : n;       // (k_mix_fir: the stream exists on request only)
    ctx->stage_len[PDT_ST_LOCK] = need_lock ? n : 0;
    ctx->stage_len[PDT_ST_FIR] = (uint64_t)n_out;
    ctx->stage_len[PDT_ST_AGC] = (uint64_t)n_out;
    ctx->stage_len[PDT_ST_AGC_RAW] = ctx->keep_agc_raw ? (uint64_t)n_out : 0;
    ctx->stage_len[PDT_ST_SYM] = sc.nsym;
    ctx->stage_len[PDT_ST_SYMIDX] = sc.nsym;
    ctx->stage_len[PDT_ST_BITS] = sc.nbits;
    ctx->stage_len[PDT_ST_BITSYM] = sc.nbits;

    // ---- time stamps (host): SURVEY Appendix B Q1/Q2/Q4
    ctx->frames_host.resize(sc.nframes);
    ctx->frames_on_device = sc.nframes;
    ctx->have_frames = true;
    ctx->tip_host.clear();
    for (unsigned f = 0; f < sc.nframes; f++) {
        pdt_frame &o = ctx->frames_host[f];
        const FrameRec &r = recs[f];
        memset(&o, 0, sizeof o);
        o.bit_index = r.bit_index;
        o.time_src = r.time_src;
        o.inverted = argos ? 0 : r.inverted;          // ARGOSdemodPortAudio/ByteSync.c:128: "%.5f " for the inverse word as well
        o.nbytes = r.nbytes;
        o.complete = r.complete;
        memcpy(o.bytes, r.bytes, 104);
        o.time = frame_time(r.time_src, N);
    }

    collect_timers();
    if (ctx->progress_fn && !ctx->chunk_host.empty()) {          // pdt_set_progress: a whole capture reports once, at its end
        try {
            std::vector<pdt_chunk_report> rep(ctx->chunk_host.size());
            chunk_reports_range(ctx, 0, rep.size(), ctx->report_samples, rep.data(), nullptr);
            ctx->progress_fn(ctx->progress_user, 0, rep.data(), rep.size(), &ctx->stats);
        } catch (const std::bad_alloc &) {
            return PDT_ERR_NOMEM;
        }
    }
    return PDT_OK;
}

// phase: the whole call, or split for the batched entry point -- enqueue every kernel and the read-back copies of
// one capture (no host synchronisation), later wait for them and build the host-side results
enum { RUN_ALL = 0, RUN_ENQUEUE = 1, RUN_FINISH = 2 };

template <typename T> int run_capture(pdt_ctx *ctx, uint64_t n, int phase = RUN_ALL)
{
    const bool argos = ctx->cfg.mode == PDT_MODE_ARGOS;
    // the sound-card twin's chain (POESTIPdemodPortAudio/main.c:324-393): the twin's constants, the lock signal kept and
    // Squelch between PLL and FIR
    const bool live = !argos && ctx->cfg.chain == PDT_CHAIN_LIVE;
    const bool inject = ctx->inj.active && !ctx->sc.active;          // pdt_stage_pll
    const bool need_lock = argos || live || inject;
    const long long N = (long long)n;
    const int interp = (int)ctx->interp;
    const int ntaps = (int)ctx->ntaps;
    const long long n_out = N * interp;
    const long long chunk = (long long)ctx->cfg.chunk;
    const long long chunk_out = chunk * interp;
    hipStream_t st = ctx->stream;
    Plan &PL = ctx->plan;
    PlanGroups L{PL, ctx->cfg.profile != 0};
    // stream segment: the window [0, n) holds history before `first`; every stage starts at `first` from the carried state
    StreamCarry *seg = ctx->sc.active ? &ctx->sc : nullptr;
    const long long first = seg ? seg->first : 0;
    const long long first_out = first * interp;

    // ---- parameters
    const T Fs = (T)ctx->cfg.sample_rate;
    PllParams<T> PP = make_pll_params<T>(ctx);
    if (inject) {
        PP.want_lock = 1;
        if (ctx->inj.started && !ctx->inj.locked) {  // the acquisition goes on from the caller's state record
            PP.phase0 = (T)ctx->inj.phase; PP.freq0 = (T)ctx->inj.freq; PP.avg0 = (T)ctx->inj.avg; PP.locksig0 = (T)ctx->inj.locksig;
            PP.sweep0 = (T)ctx->inj.sweep; PP.i0 = 0;
        }
    }
    if (seg && first > 0 && !seg->locked) {          // the acquisition goes on where the last segment left it
        PP.phase0 = (T)seg->phase; PP.freq0 = (T)seg->freq; PP.avg0 = (T)seg->avg; PP.locksig0 = (T)seg->locksig;
        PP.sweep0 = (T)seg->sweep; PP.i0 = first;
    }
    AgcParams<T> AP;
    const T fsi = Fs * (T)interp;                                              // POESTIPdemod/main.c:429
    AP.attack = ctx->lp.agc_attack != 0 ? (T)ctx->lp.agc_attack : (T)(79.5775 * (2.0 * M_PI / (double)fsi));
    AP.decay = ctx->lp.agc_decay != 0 ? (T)ctx->lp.agc_decay : (T)(159.1549 * (2.0 * M_PI / (double)fsi));
    AP.squelch = argos ? 1 : 0;
    AP.squelch_thr = (T)0.15;                                                  // ARGOSdemod/main.c:46,276
    AP.raw_out = nullptr;
    GardnerParams<T> GP;
    const T baud = ctx->lp.gardner_baud != 0 ? (T)ctx->lp.gardner_baud : argos ? (T)(400 * 2.0) : (T)(8320 * 2 + 0.3);   // main.c:90 / ARGOS main.c:64
    GP.step = (T)(int)fsi / baud;                                              // GardenerClockRecovery.c:19
    GP.kp = (ctx->lp.gardner_kp != 0 || (ctx->lp.zero_mask & PDT_LP_ZERO_GARDNER_KP)) ? (T)ctx->lp.gardner_kp : (T)3.0;
    GP.lim = (ctx->lp.gardner_step_range != 0 || (ctx->lp.zero_mask & PDT_LP_ZERO_GARDNER_STEP_RANGE)) ? (T)ctx->lp.gardner_step_range : (T)0.1;
    GP.n_total = n_out;
    GP.chunk_out = chunk_out;
    GP.argos_heap = 0;
    GP.argos_field_bits = 0;
    GP.argos_even = 0;
    if (argos && (size_t)chunk * sizeof(T) < 128 * 1024) {                    // Q16, below M_MMAP_THRESHOLD
        // (sizeof(T) = 4: the ARGOS sound-card twin, whose buffers are `chunk` floats, ARGOSdemodPortAudio/main.c:61-63)
        const unsigned long long req = (unsigned long long)sizeof(T) * (unsigned long long)chunk;
        const unsigned long long csz = (req + 8 + 15) & ~15ull;
        GP.argos_heap = 1;
        GP.argos_field_bits = csz | 1ull;
        GP.argos_even = (int)((csz - 8 - req) / sizeof(T));
    }
    const bool argos_twin = argos && ctx->cfg.chain == PDT_CHAIN_LIVE;
    const T manch_thr = (ctx->lp.manchester_threshold != 0 || (ctx->lp.zero_mask & PDT_LP_ZERO_MANCHESTER_THRESHOLD)) ? (T)ctx->lp.manchester_threshold : argos ? (T)0.5 : live ? (T)0.75 : (T)1.0;              // main.c:445 / ARGOS main.c:282 / twin :65,393
    const SyncParams SP = make_sync_params(argos, argos_twin);

    // ---- block-parallel geometry (any values give the same output; they only move time around)
    const double fs_d = (double)ctx->cfg.sample_rate;
    auto round4 = [](long long v) { return (v + 3) / 4 * 4; };
    long long Bp = ctx->cfg.pll_block ? ctx->cfg.pll_block : (long long)((argos ? 0.5 : 0.05) * fs_d);
    // (ARGOS: the loops only contract while a burst is on the air -- between bursts the detector sees noise and two close
    // trajectories are kicked apart as fast as they are pulled together -- so a warm-up must contain one whole burst: 2 s cover
    // a 1.5 s repetition period + one 0.36 s burst; measured on the 5-minute capture: 8 / 4 / 2 / 1 s -> 0 / 0 / 0 / 192 repaired
    // seams.  Sparser transmissions fall back on the seam repairs, as ever.)
    long long Wp = ctx->cfg.pll_warm ? ctx->cfg.pll_warm : (long long)((argos ? 2.0 : 0.3) * fs_d);
    if (!ctx->cfg.pll_warm && !argos) {
        // Measured (tools/pll_geom.py, 50 ksps): the probability that a block has not merged bit for bit after a
        // warm-up of W samples falls like 200 exp(-W / 2.82 tau), tau = 2 / alpha_trk the tracking loop's time
        // constant.  Aim at ~0.1 unhealthy seam per capture (a repair costs a sequential block walk, ~0.2 ms, against ~0.08 ms
        // for the extra 1.6 tau of warm-up): short captures get away with less than 0.3 s, hour-long ones need a little more.
        const double tau = 2.0 / (double)PP.alpha_trk;
        const double nb = std::max(1.0, (double)N / (double)std::max<long long>(1, Bp));
        // 200 nb exp(-w / 2.82 tau) = 0.1 expected unhealthy seams per capture.  (Round 3, an hour of 250 ksps, 61 000 blocks:
        // warm-up x1 / 0.85 / 0.7 / 0.55 -> 1 / 1 / 52 / 837 repairs, phase kernel 7.6 / 6.7 / 6.0 / 5.2 ms, repairs 1.4 / 1.2 / 2.0 /
        // 10.0 ms -- but the law with 50 in place of 2000 (x0.8) turned the same capture at six times the noise from 26 repairs in
        // 3 ms into 315 in 202 ms: where the loop barely contracts every open seam starts a cascade.  The long warm-up stays.)
        double w = 2.82 * tau * log(2000.0 * nb);
        w *= ctx->tune.pll_warm_scale;
        Wp = (long long)std::min(std::max(w, 0.15 * fs_d), 0.6 * fs_d);
    }
    long long Wacq = (long long)(0.02 * fs_d);             // acquisition-gain stage of the warm-up
    long long Ba = ctx->cfg.agc_block ? ctx->cfg.agc_block : (long long)((argos ? 0.125 : 0.0625) * fs_d * interp);
    // (the cap of a block's warm-up, which is 11 gain time constants = 11 gain / decay samples: weak input -- the noise in front
    // of a pass -- means a high gain and a long memory; capped at one second, every seam of a minute of noise failed its check
    // and was repaired in order, 219 ms; eight seconds cover gains up to ~700)
    long long Wa = ctx->cfg.agc_warm ? ctx->cfg.agc_warm : (long long)((argos ? 2.0 : 8.0) * fs_d * interp);
    if (!ctx->cfg.pll_block) {
        // One walker wavefront saturates the vector ALU of its SIMD (a wave64 instruction occupies the 16 lanes for 4 clocks), so
        // a second one on the same SIMD doubles the time of both: keep the walkers of everything that runs together -- this
        // capture, or the whole batch -- under ~one per SIMD (240 groups of four; the rest is left to the serial kernels).
        // (A capture on its own, round 3: 180 -- an hour at 250 ksps, where the warm-ups' re-reads press on HBM: blocks of 15 / 20 /
        // 25 / 30 / 40 thousand samples -> phase kernel 7.6 / 6.9 / 6.8 / 7.7 / 8.3 ms.)
        // (Round 4, with wavefront 0 out of the way -- it had been walking 7 B steps and favoured short blocks: the kernel sits under
        // the issue roof, (W + B) steps, and the HBM roof of its re-reads, W / B + 2 passes over the capture, at once; blocks of
        // 20 / 22.5 / 24 / 25 / 28 thousand samples -> 5.55 / 5.45 / 5.27 / 5.25 / 5.29 ms: 150 groups.)
        const long long groups_max = ctx->batch_hint > 1 ? 240 : 150;
        const long long share = std::max<long long>(1, groups_max / std::max(1, ctx->batch_hint));
        const long long b_min = (N + share * 256 - 1) / (share * 256);
        Bp = std::max(Bp, b_min);
    }
    Bp = std::min<long long>(Bp, std::max<long long>(N, 1));   // (one block at most: the LT layout keeps 64 blocks per tile)
    Bp = std::max<long long>(64, (Bp + 63) / 64 * 64);         // whole transposition groups of the LT layout (pdt_kernels_front.h)
    // mix and filter in one kernel (k_mix_fir: the PLL output never goes through HBM): the float chain at INTERP 1 with the
    // register-tiled taps; its runs and ring phases need blocks of a multiple of lcm(64, 208) = 832 samples
    // A stream segment may take the whole-capture kernels (k_mix_fir, the FIR kernel's AGC maps, k_agc_block_tr, table rows of
    // several chunks) when its first new sample sits where those kernels' units begin: a multiple of the mix + FIR kernel's run
    // (208 outputs: the AGC maps) and of a 128-byte line of the output streams (32 floats).  Round 5: the overlapped ingest of
    // an hour-long file cuts its segments there (demod_overlapped); a pushed stream gets there when its pushes happen to.
    const bool seg_fast = seg && first_out % PDT_MF_RUN == 0 && first_out % 32 == 0 && !ctx->tune.seg_plain;
    bool fuse_mix = false;
    if constexpr (std::is_same<T, float>::value) {
        fuse_mix = !argos && !live && !inject && (!seg || seg_fast) && interp == 1 && ntaps == 26 && ctx->taps_rot.p && !ctx->tune.fir_generic &&
                   !ctx->tune.mix_unfused && N >= 832;
        if (fuse_mix && !ctx->cfg.pll_block) Bp = (Bp + 831) / 832 * 832;
        if (fuse_mix && Bp % 832 != 0) fuse_mix = false;
    }
    Ba = std::max<long long>(64, round4(Ba));
    // POES with the register-tiled FIR: AGC blocks made of whole FIR tiles (64 * 26 inputs), so that the FIR kernel can
    // deliver the AGC's affine tile maps itself (0 = not fused: explicit block size, ARGOS, generic FIR)
    long long agc_tiles_per_block = 0, fused_tiles = 0, agc_maps_per_block = 0;
    if (!argos && !ctx->cfg.agc_block && ntaps == 26 * interp && ctx->taps_rot.p && !ctx->tune.fir_generic &&
        !ctx->tune.agc_unfused && (!seg || (seg_fast && first_out % (64ll * 26 * interp) == 0) || (seg_fast && fuse_mix))) {
        const long long tile_out = 64ll * 26 * interp;
        agc_tiles_per_block = std::max<long long>(1, (Ba + tile_out / 2) / tile_out);
        // a walker's look-ahead ring takes 64 KiB of LDS: two wavefronts per CU, 512 on the chip.  An hour at 250 ksps has more
        // blocks than that (940 wavefronts = two rounds, the second one of a few stragglers): longer blocks, one round
        // (4.2 -> 3.5 ms; tools/jobs/agc_tpb.sh).  (A batch shares the chip: no gain measured there.)
        {
            const long long tiles = (n_out - first_out + tile_out - 1) / tile_out;
            const long long one_round = (tiles + 500ll * 64 - 1) / (500ll * 64);
            if (ctx->batch_hint <= 1 && tiles / agc_tiles_per_block > 512ll * 64 && one_round <= 4 * agc_tiles_per_block)
                agc_tiles_per_block = one_round;
        }
        if (ctx->tune.agc_tpb) agc_tiles_per_block = ctx->tune.agc_tpb;
        Ba = agc_tiles_per_block * tile_out;
    }
    Wp = round4(Wp);
    Wa = round4(Wa);
    Wacq = round4(Wacq);
    // lag of the autocorrelation frequency guess: at least one Manchester symbol (so that the
    // modulation of the two samples is independent and the carrier term dominates the mean), while
    // pi/lag stays above the PLL's frequency limit
    const double sym_rate = (double)baud, f_lim = (double)PP.max_freq * fs_d / (2.0 * M_PI);
    int lag = (int)ceil(fs_d / sym_rate);
    lag = std::max(1, std::min(lag, (int)(fs_d / (2.2 * f_lim))));
    lag = std::min(lag, 64);
    const long long nb_pll = N / Bp + 2;
    const long long nb_agc = (n_out - first_out + Ba - 1) / Ba + 1;

    // ---- capacities
    double min_step = (double)GP.step - 0.25;
    if (ctx->cfg.sampler == PDT_SAMPLER_MM) {
        const double rg = ctx->cfg.mm_step_range != 0 ? ctx->cfg.mm_step_range : 3.0;
        if (!(rg >= 0) || rg >= (double)baud * 0.5) return PDT_ERR_ARG;        // stepMax must stay positive and finite
        min_step = (double)(int)fsi / ((double)baud + rg) * 0.999;
    }
    const long long n_chunks = chunk_out > 0 ? (n_out + chunk_out - 1) / chunk_out : 0;
    const long long sym_cap = (long long)((double)n_out / min_step) + n_chunks + 64 + (seg ? 2 * PDT_SEG_KEEP : 0);
    const long long bit_cap = sym_cap;
    // worst legal hit density: the ARGOS word overlaps itself by 3 bits (one hit per 10 bits), the POES word followed by
    // its inverse by 3 (one per 16) -- e.g. a repeated sync-word test pattern; the reference decodes such streams
    const uint32_t hit_cap = next_pow2((uint32_t)(bit_cap / 10 + 4096));
    const uint32_t frame_cap = (uint32_t)(bit_cap / SP.span + 16);
    const long long n_tiles = (sym_cap + PDT_TILE - 1) / PDT_TILE;
    const long long n0 = std::min<long long>(chunk, N);

    int rc;
    if (seg) {
        // the windows of a stream keep what earlier segments wrote (FIR history, stale reads of the sampler's chunk seams)
        if ((rc = ctx->pll.ensure_keep((size_t)(N + 1) * sizeof(T), (size_t)first * sizeof(T)))) return rc;
        if (need_lock && (rc = ctx->lock.ensure_keep((size_t)(N + 1) * sizeof(T), (size_t)first * sizeof(T)))) return rc;
    }
    if ((rc = ctx->pll.ensure((size_t)(N + 1) * sizeof(T)))) return rc;
    if (need_lock && (rc = ctx->lock.ensure((size_t)(N + 1) * sizeof(T)))) return rc;
    // theta and the phases live in the FIR / AGC buffers before those are written, in the lane-tiled layout: whole tiles of
    // 64 blocks, plus the rows the walkers' look-ahead loads may touch past a tile
    const long long lt_tiles = (N / Bp + 1 + 63) / 64;
    const long long lt_elems = lt_tiles * 64 * Bp + 48 * 64 * (16 / (long long)sizeof(T)) + 64;
    if (seg) {
        // ... so theta and the phases get buffers of their own there (the AGC window must survive the PLL)
        if ((rc = ctx->agc.ensure_keep((size_t)(n_out + 1) * sizeof(T), (size_t)first_out * sizeof(T)))) return rc;
        if ((rc = ctx->lt_theta.ensure((size_t)(lt_elems + 1) * sizeof(T)))) return rc;
        if ((rc = ctx->lt_phi.ensure((size_t)(lt_elems + 1) * sizeof(T)))) return rc;
        if ((rc = ctx->seg_dev.ensure(sizeof(SegTail<T>) + 256))) return rc;
    }
    // (+ Ba + 1024: the full-line AGC walkers read whole super-batches of a last, partial block and a few beyond it)
    if ((rc = ctx->fir.ensure((size_t)((seg ? n_out : std::max(n_out, lt_elems)) + 1 + Ba + 1024) * sizeof(T)))) return rc;
    if ((rc = ctx->agc.ensure((size_t)((seg ? n_out : std::max(n_out, lt_elems)) + 1 + Ba + 1024) * sizeof(T)))) return rc;
    if (ctx->keep_agc_raw && (rc = ctx->agc_raw.ensure((size_t)(n_out + 1) * sizeof(T)))) return rc;
    if (ctx->keep_agc_raw) AP.raw_out = (T *)ctx->agc_raw.p;
    // quality figure (pdt_keep_quality; whole captures only): averagePhase is one more EMA of the lock detector's kind
    // (CarrierTrackingPLL.c:80,124,152), alpha 0.00005 -> blocks of one time constant, 16 of warm-up behind the affine guess
    // (in the segments of the overlapped ingest too: they begin and end on chunk boundaries, StreamCarry::quality)
    const bool quality = (ctx->keep_quality || inject) && (!seg || (seg->quality && first % chunk == 0)) && N > 0;
    const T avg_alpha = (T)0.00005;
    // Blocks of one time constant for captures up to ~8 000 of them; longer captures get longer blocks (up to eight time
    // constants), the warm-up stays 16 time constants: with one-time-constant blocks an hour at 250 ksps re-read its input 17
    // times (61 GB, 18.7 ms -- and 59 ms beside the chain's kernels on the side stream); 45 000 blocks -> 8 192: 14 GB.
    const long long tau_q = std::max<long long>(64, round4((long long)(1.0 / (double)avg_alpha)));
    const long long Bq = std::max(tau_q, std::min(8 * tau_q, round4(N / 8192)));
    const long long Wq = 16 * tau_q;
    const long long nb_q = N / Bq + 2;
    double *d_q_zresp = nullptr, *d_q_guess = nullptr;
    if (quality) {
        if ((rc = ctx->avgph.ensure((size_t)(N + 1) * sizeof(T)))) return rc;
        if ((rc = ctx->term_ap.ensure((size_t)(N + 1) * sizeof(T)))) return rc;
        if ((rc = ctx->seams_q.ensure((size_t)nb_q * (sizeof(EmaSeam<T>) + 2 * sizeof(double))))) return rc;
        if ((rc = ctx->chunkinfo.ensure((size_t)(n_chunks + 1) * sizeof(ChunkInfo)))) return rc;
        d_q_zresp = (double *)((EmaSeam<T> *)ctx->seams_q.p + nb_q);
        d_q_guess = d_q_zresp + nb_q;
        const size_t want = (size_t)(n_chunks + 1) * sizeof(ChunkInfo);
        if (want > ctx->qual_pin_cap) {
            if (ctx->qual_pin) (void)hipHostFree(ctx->qual_pin);
            ctx->qual_pin = nullptr;
            ctx->qual_pin_cap = 0;
            if (timed_host_malloc((void **)&ctx->qual_pin, want + want / 4) != hipSuccess) {
                (void)hipGetLastError();
                return PDT_ERR_NOMEM;
            }
            ctx->qual_pin_cap = want + want / 4;
        }
    }
    T *d_avgph = quality ? (T *)ctx->avgph.p : nullptr;
    if ((rc = ctx->sym.ensure((size_t)sym_cap * sizeof(T)))) return rc;
    if ((rc = ctx->symidx.ensure((size_t)sym_cap * sizeof(long long)))) return rc;
    if ((rc = ctx->bits.ensure((size_t)bit_cap))) return rc;
    if ((rc = ctx->bitsym.ensure((size_t)bit_cap * sizeof(unsigned)))) return rc;
    if ((rc = ctx->hits.ensure((size_t)hit_cap * sizeof(unsigned) + (size_t)n_tiles * sizeof(ManchTile)))) return rc;
    if ((rc = ctx->sync_scr.ensure((2 * ((size_t)hit_cap + 1) + hit_cap / 32 + 2) * sizeof(unsigned)))) return rc;
    if ((rc = ctx->frames.ensure((size_t)frame_cap * sizeof(FrameRec)))) return rc;
    if ((rc = ctx->stiles.ensure((size_t)((bit_cap + 4095) / 4096 + 1) * sizeof(SyncTile)))) return rc;
    if ((rc = ctx->mag.ensure((size_t)(n0 + 1) * sizeof(T)))) return rc;
    if ((rc = ctx->seams_pll.ensure((size_t)nb_pll * sizeof(PllSeam<T>)))) return rc;
    if ((rc = ctx->seams_agc.ensure((size_t)nb_agc * sizeof(AgcSeam<T>)))) return rc;
    if (need_lock && (rc = ctx->term.ensure((size_t)(N + 1) * sizeof(T)))) return rc;
    // lock-detector EMA (ARGOS): a pure contraction with factor 1 - lockSigAlpha per sample, so its own, much
    // shorter geometry: 45 time constants of warm-up agree in 53 bits (20 in 24), blocks a quarter of that
    // With the affine guess of the state at every block boundary (k_lock_ema_zero / _guess) 16 time constants do.
    const bool ema_guess = !ctx->tune.ema_noguess;
    long long We = round4((long long)((sizeof(T) == 8 ? 45.0 : 20.0) / (double)PP.lock_alpha) + 64);
    long long Be = std::max<long long>(64, round4(We / 4));
    if (ema_guess) {
        Be = std::max<long long>(64, round4((long long)(1.0 / (double)PP.lock_alpha)));
        We = 16 * Be;                  // a difference of D ulps survives n steps with probability ~ D (1 - alpha)^n; D is a few tens
    }
    const long long nb_ema = N / Be + 2;
    if (need_lock && (rc = ctx->seams_ema.ensure((size_t)nb_ema * (sizeof(EmaSeam<T>) + 2 * sizeof(double))))) return rc;
    double *d_ema_zresp = need_lock ? (double *)((EmaSeam<T> *)ctx->seams_ema.p + nb_ema) : nullptr;
    double *d_ema_guess = need_lock ? d_ema_zresp + nb_ema : nullptr;
    if ((rc = ctx->scal.ensure(sizeof(DevScalars)))) return rc;
    if ((rc = ctx->lockinfo.ensure(sizeof(PllLockInfo<T>)))) return rc;

    IqSrc d_pcm;
    d_pcm.p = ctx->pcm_dev;
    d_pcm.fmt = ctx->pcm_fmt;
    T *d_pll = (T *)ctx->pll.p;
    T *d_lock = need_lock ? (T *)ctx->lock.p : nullptr;
    T *d_fir = (T *)ctx->fir.p;
    T *d_agc = (T *)ctx->agc.p;
    T *d_sym = (T *)ctx->sym.p;
    long long *d_symidx = (long long *)ctx->symidx.p;
    unsigned char *d_bits = (unsigned char *)ctx->bits.p;
    unsigned *d_bitsym = (unsigned *)ctx->bitsym.p;
    ManchTile *d_tiles = (ManchTile *)((unsigned char *)ctx->hits.p + (size_t)hit_cap * sizeof(unsigned));
    FrameRec *d_frames = (FrameRec *)ctx->frames.p;
    DevScalars *d_sc = (DevScalars *)ctx->scal.p;
    PllLockInfo<T> *d_info = (PllLockInfo<T> *)ctx->lockinfo.p;
    T *d_norm = (T *)&d_sc->norm;
    T *d_taps = (T *)ctx->taps.p;

    if (phase != RUN_FINISH) {
    PL.clear();
    PL.side_stream = ctx->stream2;
    PL.simple(OP_EV0);
    PL.memset_async(d_sc, 0, sizeof(DevScalars));
    PL.memset_async(d_frames, 0, (size_t)frame_cap * sizeof(FrameRec));

    // ---- StaticGain over the first chunk (main.c:384-389)
    // per-segment staging in pinned memory (lives until the plan has run): [0] AGC gain, [64] lock record, [256] two
    // history symbols, [512] kept bits
    unsigned char *spin = ctx->seg_pin;
    if (seg && seg->have_norm) {
        const T g = (T)seg->gain;                    // the AGC goes on from its gain at the end of the last segment
        memcpy(spin, &g, sizeof g);
        PL.copy(OP_H2D, d_norm, spin, sizeof(T));
    } else {
        L.begin("static_gain");
        PDT_LAUNCH(256, k_static_gain<T>, dim3(1), dim3(256), 0, st, d_pcm, n0, (T *)ctx->mag.p, (T)1.0,
                           ctx->cfg.norm_override, d_norm);
        L.end();
    }

    // ---- PLL: sequential acquisition, then theta / block-parallel phase recurrence / seam repair / mix.
    // theta and phase live in the (not yet used) FIR and AGC buffers.
    T *d_theta = seg ? (T *)ctx->lt_theta.p : d_fir;
    T *d_phi = seg ? (T *)ctx->lt_phi.p : d_agc;
    const long long lt_groups = lt_tiles * (Bp / (16 * (16 / (long long)sizeof(T))));      // workgroups of the transposing kernels
    if (N > 0) {
        L.begin("pll_theta");
        PDT_LAUNCH(256, k_pll_theta<T>, dim3((unsigned)lt_groups), dim3(256), 0, st, d_pcm, N, Bp, d_theta);
        L.end();
    }
    // fork: the block-parallel phase recurrence (side stream) runs beside the sequential acquisition
    // workgroups of four wavefronts: the dispatcher spreads a workgroup's wavefronts over the four SIMDs of a CU, so the
    // walkers are balanced over the SIMDs by construction (single-wavefront groups piled up on some SIMDs once there were
    // more than ~1000 of them: 250 ksps hour-long captures, batches)
    const long long grid_pll = (nb_pll + 255) / 256;
    // one more workgroup for the walkers whose warm-up begins at sample 0 (k_pll_phase), when they fit into a wavefront
    // ... and when it pays: in place, wavefront 0 runs every segment of the tracking stage for the longer of its two kinds of lane --
    // the early walkers' first segment is B - (wide + acquisition stage), everybody else's W mod B (a wavefront with few lanes
    // is not faster, rather the opposite: ARGOS, W = 4 B, +0.3 ms with the extra workgroup)
    bool short_pays = false;
    {
        const long long w0 = ((Wacq / 4 + 3) & ~3ll) + Wacq;
        const long long a = (Wp % Bp) ? Wp % Bp : Bp, s_first = Bp - w0 % Bp;
        short_pays = (s_first - a) * 50 > w0 + Wp + Bp;
    }
    const int short_group = (short_pays && (((Wacq / 4 + 3) & ~3ll) + Wacq + Wp + Bp - 1) / Bp <= 64 && nb_pll > 64 && !ctx->tune.pll_noshort) ? (int)grid_pll : -1;
    // outliers behind the wide-band stage take their wavefront's median frequency (k_pll_phase): POES only -- an ARGOS capture holds
    // many transmitters, each at its own offset
    const T pll_consensus = (!argos && !ctx->tune.pll_noconsensus) ? (T)(2.0 * M_PI * 800.0 / fs_d) : (T)0;
    const long long phase_groups = grid_pll + (short_group >= 0 ? 1 : 0);
    // a single +-2pi correction per step is exact as long as one step cannot move the phase by 2pi
    const double worst = (double)PP.max_freq + M_PI * std::max({(double)PP.alpha_acq + (double)PP.beta_acq,
                                                                (double)PP.alpha_trk + (double)PP.beta_trk,
                                                                (double)PP.alpha_wide + (double)PP.beta_wide});
    const bool slow_wrap = worst >= 2.0 * M_PI - 0.05;
    // the walkers' frequency checkpoints (pll_phase_range): with them a seam repair stops where it has merged with the stored
    // trajectory.  All ones = a NaN no loop state equals: a checkpoint nobody wrote never matches.
    T *d_ckpt = nullptr;
    if (N > 0 && !ctx->tune.pll_nockpt) {
        const size_t ck_bytes = (size_t)((nb_pll + 63) / 64 + 1) * (size_t)pll_ckpt_count(Bp) * 64 * sizeof(T);
        if ((rc = ctx->pll_ckpt.ensure(ck_bytes))) return rc;
        d_ckpt = (T *)ctx->pll_ckpt.p;
        PL.memset_async(d_ckpt, 0xff, ck_bytes);
    }
    PllPhaseHint *d_hint = &d_sc->phase_hint;
    if (N > 0) {
        PL.simple(OP_FORK);
        L.begin("pll_phase", ctx->stream2);
        if (slow_wrap)
            PDT_LAUNCH(256, (k_pll_phase<T, true>), dim3((unsigned)phase_groups), dim3(256), 0, ctx->stream2, d_pcm, d_theta, N, PP, Bp,
                               Wacq, Wp, lag, d_phi, (PllSeam<T> *)ctx->seams_pll.p, d_hint, short_group, d_ckpt, pll_consensus);
        else
            PDT_LAUNCH(256, (k_pll_phase<T, false>), dim3((unsigned)phase_groups), dim3(256), 0, ctx->stream2, d_pcm, d_theta, N, PP, Bp,
                               Wacq, Wp, lag, d_phi, (PllSeam<T> *)ctx->seams_pll.p, d_hint, short_group, d_ckpt, pll_consensus);
        L.end();
        PL.simple(OP_JOIN_RECORD);
    }
    // the serial kernels (acquisition, head) ask for SIMDs of their own while the block-parallel kernel beside them
    // leaves some free (1 024 SIMDs; it runs one wavefront per 64 blocks)
    const bool serial_excl = 4 * grid_pll <= 960 && !ctx->tune.no_excl;
    long long fix_regions = 1, fix_region_blocks = 0;
    bool quality_side = false;            // the averagePhase EMA of pdt_keep_quality runs on the side stream
    bool quality_term_fused = false, quality_after_fir = false;   // ... its input term written by k_mix_fir, its walkers launched behind that kernel
    std::function<void(bool)> quality_walkers_late;   // (captures function-scope state only)
    L.begin("pll_acquire");
    if (inject && ctx->inj.locked) {
        // pdt_stage_pll after the lock: sample 0 is a dummy the caller put in front, "locked at sample 0" with the state record
        // = the state after it; the kernels that start behind the lock do the rest
        PllLockInfo<T> li;
        memset(&li, 0, sizeof li);
        li.lock_sample = 0;
        li.st.phase = (T)ctx->inj.phase; li.st.freq = (T)ctx->inj.freq; li.st.avg_phase = (T)ctx->inj.avg;
        li.st.locksig = (T)ctx->inj.locksig; li.st.sweep = (T)ctx->inj.sweep;
        memcpy(spin + 64, &li, sizeof li);
        PL.copy(OP_H2D, d_info, spin + 64, sizeof li);
        PL.memset_async(d_pll, 0, sizeof(T));
        PL.memset_async(d_lock, 0, sizeof(T));
        PL.memset_async(d_avgph, 0, sizeof(T));
    } else if (seg && seg->locked) {
        // the lock happened in an earlier segment: the kernels that start "after the lock" start at `first` with the carried
        // true state (the head walks the first samples, the block-parallel results are validated against it as ever)
        PllLockInfo<T> li;
        memset(&li, 0, sizeof li);
        li.lock_sample = first - 1;
        li.st.phase = (T)seg->phase; li.st.freq = (T)seg->freq; li.st.avg_phase = (T)seg->avg;
        li.st.locksig = (T)seg->locksig; li.st.sweep = (T)seg->sweep;
        li.freq_at_lock = 0; li.avg_at_lock = (T)seg->avg_at_lock;
        memcpy(spin + 64, &li, sizeof li);
        PL.copy(OP_H2D, d_info, spin + 64, sizeof li);
    } else if (slow_wrap)
        PDT_LAUNCH(128, (k_pll_acquire_pipe<T, true>), dim3(1), dim3(128), 0, st, d_pcm, N, PP, d_pll, d_lock,
                           d_info, d_avgph);
    else if (serial_excl)
        PDT_LAUNCH(128, (k_pll_acquire_pipe<T, false, true>), dim3(1), dim3(128), 0, st, d_pcm, N, PP, d_pll, d_lock,
                           d_info, d_avgph);
    else
        PDT_LAUNCH(128, (k_pll_acquire_pipe<T, false>), dim3(1), dim3(128), 0, st, d_pcm, N, PP, d_pll, d_lock,
                           d_info, d_avgph);
    L.end();
    if (N > 0) {
        const long long grid = grid_pll;
        // the sequential head (true state from the lock onwards, ~W samples) also runs beside k_pll_phase
        // its length: until the true state has (nearly) forgotten the acquisition -- time constant of the critically
        // damped tracking loop = 2 / alpha samples; 30 of them for 24 bits (measured: 28 suffice on every test
        // capture, 40 never needed), 90 for 53 bits; a block that still disagrees afterwards is simply re-run
        const double tau_trk = 2.0 / (double)PP.alpha_trk;
        double head_taus = (sizeof(T) == 4) ? 30.0 : 90.0;
        if (ctx->tune.head_taus > 0) head_taus = ctx->tune.head_taus;
        long long Hd = std::min<long long>(Wp, (long long)(head_taus * tau_trk));
        // A stream segment that continues a lock taken long ago starts from a state that remembers the acquisition no more than
        // any other sample of the capture does: the head then only has to reach the next block boundary -- the warm-ups of the
        // blocks behind it merge with the truth as anywhere else, and the seams are validated as ever (round 5: the mandatory
        // 30 time constants kept a SIMD busy for 2.5 ms of every segment of the overlapped ingest).  What is left of the 30 since
        // the lock stays.
        if (seg && seg->locked && first > 0 && seg->in_place && !ctx->tune.head_taus) {
            const long long since = (long long)seg->origin + first - 1 - (long long)seg->lock_sample;
            Hd = std::max<long long>(0, std::min<long long>(Hd, Hd - since));
        }
        // ... and further -- up to half as long again -- for as long as the block-parallel kernel beside it is still running: those
        // blocks cost nothing, and the first block behind a 30-tau head fails its seam now and then (an hour at 250 ksps: one
        // repair of two block walks, 1.6 ms; the head stopped 0.9 ms before the kernel beside it)
        // Only where the kernel beside it is bound by HBM (its warm-ups re-read more than ~12 GB: hour-long captures) -- there it
        // runs ~25 % longer than acquisition + head; elsewhere the two finish together and an extra block is a block too many
        // (10 min at 50 ksps: +0.1 ms; 10 min at 250 ksps: +0.45 ms).
        const bool head_slack = (double)nb_pll * (double)(Wp + Bp) * sizeof(T) >= 12e9 && !ctx->tune.head_taus;
        const long long Hd_max = head_slack ? std::min<long long>(Wp, Hd + Hd / 2) : Hd;
        const long long head_blocks = Hd_max / Bp + 3;
        if ((rc = ctx->pll_head.ensure((size_t)(head_blocks * Bp + 64) * sizeof(T) + (size_t)head_blocks * sizeof(PllSeam<T>) +
                                       sizeof(PllHeadInfo<T>) + 64)))
            return rc;
        PllHeadInfo<T> *d_hinfo = (PllHeadInfo<T> *)ctx->pll_head.p;
        PllSeam<T> *d_hseams = (PllSeam<T> *)((unsigned char *)ctx->pll_head.p + 64);
        T *d_hphi = (T *)((unsigned char *)d_hseams + (((size_t)head_blocks * sizeof(PllSeam<T>) + 63) & ~(size_t)63));
        // seam repairs: regions of the capture are validated and repaired side by side before the final in-order pass
        // (k_pll_fix); every region workgroup has scratch for its 16 concurrent block re-runs
        const long long nb_fix = (N + Bp - 1) / Bp;
        fix_regions = (nb_fix >= 128) ? std::min<long long>(64, nb_fix / 16) : 1;
        fix_region_blocks = (nb_fix + fix_regions - 1) / fix_regions;
        if ((rc = ctx->pll_scratch.ensure((size_t)(fix_regions + 2) * (size_t)(PDT_FIX_THREADS / 64) *
                                          (size_t)(((Bp + 63) & ~63ll) + ((pll_ckpt_count(Bp) + 63) & ~63ll) + 64) * sizeof(T)))) return rc;
        L.begin("pll_head");
        if (slow_wrap)
            PDT_LAUNCH(64, (k_pll_head<T, true>), dim3(1), dim3(64), 0, st, d_theta, N, PP, d_info, Bp, Hd, d_hphi, d_hseams,
                               d_hinfo, head_blocks, Hd_max, (const PllPhaseHint *)d_hint, (unsigned)phase_groups,
                               ((Wacq / 4 + 3) & ~3ll) + Wacq + Wp + Bp);
        else if (serial_excl)
            PDT_LAUNCH(64, (k_pll_head<T, false, true>), dim3(1), dim3(64), 0, st, d_theta, N, PP, d_info, Bp, Hd, d_hphi, d_hseams,
                               d_hinfo, head_blocks, Hd_max, (const PllPhaseHint *)d_hint, (unsigned)phase_groups,
                               ((Wacq / 4 + 3) & ~3ll) + Wacq + Wp + Bp);
        else
            PDT_LAUNCH(64, (k_pll_head<T, false>), dim3(1), dim3(64), 0, st, d_theta, N, PP, d_info, Bp, Hd, d_hphi, d_hseams,
                               d_hinfo, head_blocks, Hd_max, (const PllPhaseHint *)d_hint, (unsigned)phase_groups,
                               ((Wacq / 4 + 3) & ~3ll) + Wacq + Wp + Bp);
        L.end();
        PL.simple(OP_JOIN_WAIT);                                           // join
        L.gap();                                                           // (the wait is not part of pll_fix)
        L.begin("pll_fix");
        {
            // graft the head, two optimistic region passes (region boundaries half a region apart), the final pass
            auto fix = [&](int mode, unsigned grid, long long rb, long long ro) {
                if (slow_wrap)
                    PDT_LAUNCH(PDT_FIX_THREADS, (k_pll_fix<T, true>), dim3(grid), dim3(PDT_FIX_THREADS), 0, st, d_theta, N, PP, d_info, Bp, d_phi,
                               (PllSeam<T> *)ctx->seams_pll.p, (const T *)d_hphi, (const PllSeam<T> *)d_hseams,
                               (const PllHeadInfo<T> *)d_hinfo, (T *)ctx->pll_scratch.p, d_sc->counters, mode, rb, ro, d_ckpt);
                else
                    PDT_LAUNCH(PDT_FIX_THREADS, (k_pll_fix<T, false>), dim3(grid), dim3(PDT_FIX_THREADS), 0, st, d_theta, N, PP, d_info, Bp, d_phi,
                               (PllSeam<T> *)ctx->seams_pll.p, (const T *)d_hphi, (const PllSeam<T> *)d_hseams,
                               (const PllHeadInfo<T> *)d_hinfo, (T *)ctx->pll_scratch.p, d_sc->counters, mode, rb, ro, d_ckpt);
            };
            fix(0, 32, 0, 0);
            if (fix_regions > 1) {
                for (int pass = 0; pass < ctx->tune.fix_passes; pass++) {
                    if (pass & 1) fix(1, (unsigned)fix_regions + 1, fix_region_blocks, fix_region_blocks / 2);
                    else fix(1, (unsigned)fix_regions, fix_region_blocks, 0);
                }
            }
            fix(2, 1, 0, 0);
        }
        L.end();
        (void)grid;
        if (!fuse_mix) {
        L.begin("pll_mix");
        if (need_lock)
            PDT_LAUNCH(256, (k_pll_mix<T, true>), dim3((unsigned)lt_groups), dim3(256), 0, st, d_pcm, d_phi, N, Bp, PP,
                               d_info, d_pll, (T *)ctx->term.p);
        else
            PDT_LAUNCH(256, (k_pll_mix<T, false>), dim3((unsigned)lt_groups), dim3(256), 0, st, d_pcm, d_phi, N, Bp, PP,
                               d_info, d_pll, (T *)nullptr);
        L.end();
        }
        if (need_lock) {
            L.begin("lock_ema");
            if (ema_guess) {
                PDT_LAUNCH(64, k_lock_ema_zero<T>, dim3((unsigned)((nb_ema + 63) / 64)), dim3(64), 0, st, (const T *)ctx->term.p, N,
                                   PP.lock_alpha, d_info, Be, d_ema_zresp);
                PDT_LAUNCH(1024, k_lock_ema_guess<T>, dim3(1), dim3(1024), 0, st, (const double *)d_ema_zresp, N, PP.lock_alpha, d_info, Be,
                                   pow(1.0 - (double)PP.lock_alpha, (double)Be), d_ema_guess);
            }
            PDT_LAUNCH(64, k_lock_ema<T>, dim3((unsigned)((nb_ema + 63) / 64)), dim3(64), 0, st, (const T *)ctx->term.p, N, PP.lock_alpha,
                               d_info, Be, We, d_lock, (EmaSeam<T> *)ctx->seams_ema.p, (const double *)(ema_guess ? d_ema_guess : nullptr), 0ll);
            PDT_LAUNCH(64, k_lock_ema_fix<T>, dim3(1), dim3(64), 0, st, (const T *)ctx->term.p, N, PP.lock_alpha, d_info,
                               Be, d_lock, (EmaSeam<T> *)ctx->seams_ema.p, &d_sc->counters[1]);
            L.end();
        }
        if (quality) {
            // averagePhase after the lock: its input term from the phases (still in place), then the EMA like the lock detector's.
            // Nothing later in the chain needs it (only k_chunk_info, at the very end): whole captures run it on the side stream,
            // beside the filter, the AGC and the sampler; the chain only waits for the kernel that reads the phases (the AGC
            // output will take their place).
            quality_side = !inject && !ctx->tune.quality_inline;
            // Round 4: where mix and filter are one kernel (k_mix_fir, eight wavefronts) that kernel writes the EMA's input term
            // on its way; the walkers below are then launched behind it (quality_walkers, further down)
            quality_term_fused = quality_side && fuse_mix && std::is_same<T, float>::value;
            }
        quality_walkers_late = [&](bool term_here) {
            hipStream_t sq = quality_side ? ctx->stream2 : st;
            if (quality_side) PL.simple(OP_FORK);
            L.begin("quality", sq);
            T *d_tap = (T *)ctx->term_ap.p;
            if (term_here)
            PDT_LAUNCH(256, (k_pll_mix<T, false, true>), dim3((unsigned)lt_groups), dim3(256), 0, sq, d_pcm, d_phi, N, Bp, PP,
                               d_info, (T *)nullptr, d_tap);
            if (quality_side && term_here) PL.simple(OP_JOIN_RECORD);
            PDT_LAUNCH(64, k_lock_ema_zero_wave<T>, dim3((unsigned)nb_q), dim3(64), 0, sq, (const T *)d_tap, N,
                               avg_alpha, d_info, Bq, d_q_zresp);
            PDT_LAUNCH(1024, (k_lock_ema_guess<T, true>), dim3(1), dim3(1024), 0, sq, (const double *)d_q_zresp, N, avg_alpha, d_info, Bq,
                               pow(1.0 - (double)avg_alpha, (double)Bq), d_q_guess);
            PDT_LAUNCH(64, (k_lock_ema<T, true>), dim3((unsigned)((nb_q + 63) / 64)), dim3(64), 0, sq, (const T *)d_tap, N, avg_alpha,
                               d_info, Bq, Wq, d_avgph, (EmaSeam<T> *)ctx->seams_q.p, (const double *)d_q_guess, (long long)chunk);
            PDT_LAUNCH(64, k_lock_ema_fix<T>, dim3(1), dim3(64), 0, sq, (const T *)d_tap, N, avg_alpha, d_info,
                               Bq, d_avgph, (EmaSeam<T> *)ctx->seams_q.p, &d_sc->pad0_);
            L.end();
        };
        if (quality && !quality_term_fused) quality_walkers_late(true);
        quality_after_fir = quality && quality_term_fused;
        if (live) {                                            // twin main.c:370, DSP_SQLCH_THRESH 0.05 (:55)
            L.begin("squelch");
            PDT_LAUNCH(256, k_squelch<T>, dim3((unsigned)((N + 1023) / 1024)), dim3(256), 0, st, d_pll, (const T *)d_lock, N, (T)0.05);
            L.end();
        }
    }

    const size_t ops_after_pll = PL.ops.size();
    ctx->last_pll_block = Bp;

    // ---- FIR
    if (n_out > 0) {
        const int opt = 8;
        const long long tile = (long long)PDT_FIR_THREADS * opt;
        const long long grid = (n_out + tile - 1) / tile;
        L.begin(fuse_mix ? "mix_fir" : "fir");
        if (argos) {
            const size_t sh = (size_t)(ntaps + tile + ntaps + 8) * sizeof(T);
            PDT_LAUNCH(PDT_FIR_THREADS, k_fir_plain<T>, dim3((unsigned)grid), dim3(PDT_FIR_THREADS), sh, st, d_pll, N, ntaps, d_taps,
                               d_fir, opt);
        } else {
            const int K = ntaps / interp;
            const size_t sh_rt = (size_t)(65 * (K + 1) + 3 + 64 * K * interp) * sizeof(T);
            const long long tiles_rt = (N + 64ll * K - 1) / (64ll * K);
            // the AGC's affine tile maps are folded into this kernel when the AGC blocks are whole FIR tiles
            AgcMap *fir_tile_maps = nullptr;
            if (agc_tiles_per_block > 0) {
                if ((rc = ctx->agc_maps.ensure((size_t)(tiles_rt + 1) * sizeof(AgcMap) + (size_t)(nb_agc + 2) * (sizeof(double) + sizeof(AgcMap))))) return rc;
                fir_tile_maps = (AgcMap *)ctx->agc_maps.p;
                fused_tiles = tiles_rt;
            }
            // workgroups take tiles round robin; 32 per CU measured best (c3: 3.9 / 3.7 / 3.2 / 3.1 / 3.1 ms at 4 / 8 / 16 / 32 / 64):
            // the taps come by scalar loads per residue, nothing is kept across tiles, and a finer grain balances the tail
            const long long fir_tpb = ctx->tune.fir_wg_per_cu > 0 ? ctx->tune.fir_wg_per_cu : 32;
            const unsigned grid_rt = (unsigned)std::min<long long>(tiles_rt, 256ll * fir_tpb);
            bool done = false;
            if constexpr (std::is_same<T, float>::value) if (fuse_mix) {
                // one map per run of 208 outputs instead of one per FIR tile of 1 664 (AGC blocks are whole tiles = 8 runs each)
                const long long runs_nat = (N + PDT_MF_RUN - 1) / PDT_MF_RUN;
                AgcMap *run_maps = nullptr;
                if (agc_tiles_per_block > 0) {
                    if ((rc = ctx->agc_maps.ensure((size_t)(runs_nat + 1) * sizeof(AgcMap) + (size_t)(nb_agc + 2) * (sizeof(double) + sizeof(AgcMap))))) return rc;
                    run_maps = (AgcMap *)ctx->agc_maps.p;
                    fused_tiles = runs_nat;
                    agc_maps_per_block = agc_tiles_per_block * (64 * 26 / PDT_MF_RUN);
                }
                // eight wavefronts per workgroup (two workgroups per CU = four wavefronts per SIMD)
                const unsigned mf_grid = (unsigned)(lt_tiles * (Bp / PDT_MF_RUN));
                // (a stream segment keeps the PLL output's tail: the next segment's filter starts from its last 25 samples)
                float *mf_pll = (ctx->keep_pll || seg) ? (float *)d_pll : (float *)nullptr;
                const long long mf_pll_from = (seg && !ctx->keep_pll) ? std::max<long long>(0, N - 256) : 0ll;
#define PDT_MF_ARGS d_pcm, (const float *)d_phi, (const float *)d_pll, N, Bp, (const PllLockInfo<float> *)d_info, (const float *)ctx->taps_rot.p, (float *)d_fir, mf_pll, run_maps, (float)AP.decay
                if (quality_after_fir) {
                    float *d_tap = (float *)ctx->term_ap.p;
                    if (d_pcm.fmt == 0) PDT_LAUNCH(512, (k_mix_fir<26, 0, 8, true>), dim3(mf_grid), dim3(512), 0, st, PDT_MF_ARGS, d_tap, mf_pll_from);
                    else PDT_LAUNCH(512, (k_mix_fir<26, 1, 8, true>), dim3(mf_grid), dim3(512), 0, st, PDT_MF_ARGS, d_tap, mf_pll_from);
                } else {
                    if (d_pcm.fmt == 0) PDT_LAUNCH(512, (k_mix_fir<26, 0, 8>), dim3(mf_grid), dim3(512), 0, st, PDT_MF_ARGS, (float *)nullptr, mf_pll_from);
                    else PDT_LAUNCH(512, (k_mix_fir<26, 1, 8>), dim3(mf_grid), dim3(512), 0, st, PDT_MF_ARGS, (float *)nullptr, mf_pll_from);
                }
#undef PDT_MF_ARGS
                done = true;
            }
            if (!done && K == 26 && sh_rt <= 64000 && ctx->taps_rot.p && !ctx->tune.fir_generic) {
                done = true;
                switch (interp) {
#define PDT_FIR_CASE(I)                                                                                                       \
    case I:                                                                                                                   \
        PDT_LAUNCH(PDT_FIR_THREADS, (k_fir_interp_rt<T, I, 26>), dim3(grid_rt), dim3(PDT_FIR_THREADS), sh_rt, st, d_pll, N, (const T *)ctx->taps_rot.p, d_fir, \
                           fir_tile_maps, AP.decay);                                                                                    \
        break;
                    PDT_FIR_CASE(1) PDT_FIR_CASE(2) PDT_FIR_CASE(3) PDT_FIR_CASE(4) PDT_FIR_CASE(5) PDT_FIR_CASE(6) PDT_FIR_CASE(7) PDT_FIR_CASE(8)
#undef PDT_FIR_CASE
                default: done = false;
                }
            }
            if (!done) fused_tiles = 0;
            if (!done) {                                       // any other interpolation factor: generic form
                const size_t sh = (size_t)(ntaps + tile / interp + K + 8) * sizeof(T);
                PDT_LAUNCH(PDT_FIR_THREADS, k_fir_interp<T>, dim3((unsigned)grid), dim3(PDT_FIR_THREADS), sh, st, d_pll, N, interp, K,
                                   d_taps, d_fir, opt);
            }
        }
        L.end();
    }

    // the averagePhase walkers, behind the kernel that wrote their input (side stream: nothing in the chain waits for them)
    if (quality_after_fir) quality_walkers_late(false);

    // ---- AGC (+Squelch); in a stream segment over the new outputs only, from the carried gain (*d_norm)
    long long agc_last_block = -1;
    if (n_out - first_out > 0) {
        const long long na = n_out - first_out;
        const T *a_in = d_fir + first_out;
        T *a_out = d_agc + first_out;
        const T *a_lock = d_lock ? d_lock + first_out : nullptr;       // (indexed like the outputs: interp is 1 where it is read)
        AgcParams<T> APs = AP;
        if (APs.raw_out) APs.raw_out += first_out;
        const long long nb = (na + Ba - 1) / Ba;
        agc_last_block = nb - 1;
        const long long grid = (nb + 63) / 64;
        const bool fused = fused_tiles > 0;                    // tile maps already written by the FIR kernel
        if (!fused && (rc = ctx->agc_maps.ensure((size_t)(nb + 1) * (sizeof(AgcMap) + sizeof(double))))) return rc;
        AgcMap *d_maps = (AgcMap *)ctx->agc_maps.p;
        double *d_guess = (double *)(d_maps + (fused ? fused_tiles : nb) + 1);
        // (a stream segment: the FIR kernel's maps are indexed from the window's first output, the AGC's blocks from the first
        // NEW one -- a whole number of maps further on, seg_fast)
        const long long map_len_fused = fuse_mix ? (long long)PDT_MF_RUN : 64ll * 26 * interp;
        const long long map_first = fused ? first_out / map_len_fused : 0;
        const AgcMap *d_maps_in = d_maps + map_first;
        const long long n_maps_in = fused ? fused_tiles - map_first : nb;
        // warm-up length in gain time constants: the affine guess is off by the accumulated float rounding of
        // the true recurrence only, so a few time constants make the trajectories agree to the last bit
        double agc_K = (sizeof(T) == 4) ? 11.0 : 34.0;
        if (ctx->tune.agc_k > 0) agc_K = ctx->tune.agc_k;
        if (quality_side && !quality_term_fused) PL.simple(OP_JOIN_WAIT);          // the phases (in the AGC output's buffer) have been read
        L.begin("agc_block");
        if (!fused) PDT_LAUNCH(256, k_agc_affine<T>, dim3((unsigned)nb), dim3(256), 0, st, a_in, na, APs.decay, Ba, d_maps);
        if (agc_maps_per_block == 0) agc_maps_per_block = agc_tiles_per_block;        // (the FIR kernel's maps: one per tile)
        if (fused && agc_maps_per_block > 1) {
            AgcMap *d_bmaps = (AgcMap *)(d_guess + nb + 1);
            PDT_LAUNCH(256, k_agc_blockmaps, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, d_maps_in, nb,
                               (int)agc_maps_per_block, n_maps_in, d_bmaps);
            PDT_LAUNCH(1024, k_agc_guess<T>, dim3(1), dim3(1024), 0, st, (const AgcMap *)d_bmaps, nb, (const T *)d_norm, d_guess, 1, nb);
        } else
        PDT_LAUNCH(1024, k_agc_guess<T>, dim3(1), dim3(1024), 0, st, d_maps_in, nb, (const T *)d_norm, d_guess,
                           fused ? (int)agc_maps_per_block : 1, n_maps_in);
        // (Round 3: walkers that store only the gain in front of every 16-sample batch + a streaming kernel that applies them were
        // slower, 4.2 against 3.3 ms at an hour of 250 ksps: the walkers are not held up by their output stores.)
        // Round 4: the walkers move whole lines (k_agc_block_tr) wherever the FIR kernel delivered its tile maps -- float chain,
        // no Squelch, no raw copy; the double build, stream segments and explicit block sizes keep the per-lane form.
        bool agc_tr = false;
        float *agc_ckpt = nullptr;
        if constexpr (std::is_same<T, float>::value) {
            agc_tr = fused && !ctx->tune.agc_lanes && !APs.squelch && !APs.raw_out && first_out % 32 == 0 && Ba % 32 == 0 &&
                     agc_maps_per_block > 0 && Ba % agc_maps_per_block == 0;
            if (agc_tr) {
                const long long tile_len = 64ll * 26 * interp;                         // a FIR tile (= 8 runs of k_mix_fir)
                const long long map_len = Ba / agc_maps_per_block;                     // samples per map
                agc_tr = tile_len % 32 == 0 && tile_len % map_len == 0 && Ba % tile_len == 0;
                if (agc_tr) {
                    // (no slack of a whole block behind the time constants any more.  An hour at 250 ksps: K = 10 / 11 / 12 / 13 / 14 / 16
                    // -> 4 / 3 / 0 / 0 / 0 / 0 open seams, walkers 1.76 / 1.80 / 1.84 / 1.86 / 1.91 / 2.06 ms; a minute of noise in front:
                    // 3 / 1 / 2 / 1 / 1 / 0 and 12.1 ... 19.1 ms.  A seam that stays open is cheap since the repairs stop at the first
                    // checkpoint they reproduce.)
                    if (ctx->tune.agc_k <= 0) agc_K = 12.0;
                    const size_t ck_bytes = (size_t)(nb + 64) * (size_t)(Ba / PDT_AGC_CKPT + 1) * sizeof(float);
                    if ((rc = ctx->agc_ckpt.ensure(ck_bytes))) return rc;
                    agc_ckpt = (float *)ctx->agc_ckpt.p;
                    PDT_LAUNCH(64, (k_agc_block_tr<PDT_AGC_TR_R>), dim3((unsigned)grid), dim3(64), 0, st, (const float *)a_in, na, APs,
                               (const float *)d_norm, Ba, Wa, (const double *)d_guess, d_maps_in, (int)agc_maps_per_block,
                               (int)(tile_len / map_len), n_maps_in, tile_len, (float *)a_out, (AgcSeam<float> *)ctx->seams_agc.p, agc_K, agc_ckpt);
                }
            }
        }
        T *agc_ckpt_any = (T *)agc_ckpt;
        if (!agc_tr) {
            // the per-lane walkers of the double-precision build leave checkpoints too (blocks of at least two of them)
            if (sizeof(T) == 8 && Ba >= 2 * PDT_AGC_CKPT) {
                if ((rc = ctx->agc_ckpt.ensure((size_t)(nb + 64) * (size_t)(Ba / PDT_AGC_CKPT) * sizeof(T)))) return rc;
                agc_ckpt_any = (T *)ctx->agc_ckpt.p;
            }
            PDT_LAUNCH(64, k_agc_block<T>, dim3((unsigned)grid), dim3(64), 0, st, a_in, na, APs, d_norm, Ba, Wa,
                               (const double *)d_guess, a_lock, a_out, (AgcSeam<T> *)ctx->seams_agc.p, agc_K, agc_ckpt_any);
        }
        L.end();
        L.begin("agc_fix");
        PDT_LAUNCH(1024, k_agc_scan<T>, dim3(1), dim3(1024), 0, st, na, Ba, (const AgcSeam<T> *)ctx->seams_agc.p, &d_sc->agc_first_bad);
        PDT_LAUNCH(64, k_agc_fix<T>, dim3(1), dim3(64), 0, st, a_in, na, APs, Ba, a_lock, a_out,
                           (AgcSeam<T> *)ctx->seams_agc.p, d_sc->counters, (const long long *)&d_sc->agc_first_bad, agc_ckpt_any);
        L.end();
    }

    // ---- Gardner: exact parallel evaluation through boundary-state tables when the chunk geometry
    // allows it (float build, chunk fits the LDS window, boundary states in one binade), otherwise the
    // single-wavefront sequential chain.
    const bool use_mm = ctx->cfg.sampler == PDT_SAMPLER_MM;
    bool use_table = false;
    GardnerDomain GD;
    GD.q_min = 0; GD.u = 0; GD.n_q = 0; GD.n_cand = 0; GD.pad_q = 0; GD.idx_bits = 20; GD.span = 1;
    if constexpr (std::is_same<T, float>::value) {
        const int table_len = 1 << 22;     // the table kernel walks the chunk in LDS windows: no size limit of its own
        const float nT = (float)chunk_out, stepf = (float)GP.step;
        const long long seg_c_first = (seg && chunk > 0) ? first / chunk : 0;
        if (!argos && !use_mm && !ctx->tune.gardner_sequential && n_chunks - seg_c_first >= 4 && chunk_out >= 256 &&
            !(seg && ctx->tune.seg_sequential) &&
            chunk_out + 2 * (long long)stepf + 24 <= table_len && 8 * (long long)stepf + 256 < PDT_GTAB_WIN && chunk_out < (1 << 22)) {
            int e;
            (void)frexpf(nT - stepf - 1.0f, &e);                              // value in [2^(e-1), 2^e)
            const float u = ldexpf(1.0f, e - 24);
            const float q_min = u * floorf((nT - stepf - 1.0f) / u);
            const int n_q = (int)ceilf((stepf + 1.8f) / u);
            const float q_max = q_min + (float)n_q * u;
            const double max_count = (double)chunk_out / ((double)stepf - 0.11) + 2.0;
            int idx_bits = 1;
            while ((1ll << idx_bits) < 2 * (long long)n_q) idx_bits++;
            if (q_min >= ldexpf(1.0f, e - 1) && q_max < ldexpf(1.0f, e) && idx_bits <= 20 &&
                max_count < (double)((1u << (32 - idx_bits)) - 2u) && max_count < 65000.0) {
                use_table = true;
                GD.q_min = q_min;
                GD.u = u;
                GD.n_q = n_q;
                GD.idx_bits = idx_bits;
                const double pad = ctx->cfg.gardner_band_pad > 0 ? ctx->cfg.gardner_band_pad : (ctx->tune.band_pad > 0 ? ctx->tune.band_pad : 1.0 / 8.0);
                GD.pad_q = std::max(2, (int)(pad / (double)u));
                // Chunks per table row.  Many short chunks (an hour at 250 ksps: 90 000 of 666 symbols): the candidates of a chunk
                // merge onto a handful of trajectories within it, so boundary states are tabulated in front of every span-th chunk
                // only and the few distinct exits of a row's first chunk are walked on through the others (k_gardner_span):
                // scouts, candidate walks and chain hops per span chunks instead of per chunk.  A row's symbol count must fit
                // the table cell.  Stream segments keep one chunk per row (they are short, and enter with a carried state).
                // (Round 5: a stream segment too, when its first new chunk is where a row begins -- the rows are counted from the
                // window's chunk 0, the few in front of the first new chunk are history nobody asks for.)
                int span = ctx->tune.gspan > 0 ? ctx->tune.gspan : ((n_chunks - seg_c_first >= 4096 && (!seg || seg_fast)) ? 16 : 1);
                if (2 * n_q > 32 * PDT_GSPAN_BITMAP_WORDS || 8 * (long long)stepf + 256 >= PDT_GSUB_WIN) span = 1;
                while (span > 1 && ((double)span * max_count >= (double)((1u << (32 - idx_bits)) - 2u) || (n_chunks - 1) / span - seg_c_first / span < 4))
                    span /= 2;
                auto row_aligned = [&](int sp) { return !seg || seg_c_first % sp == 0; };
                if (span >= 8 && !ctx->tune.gspan) {
                    // the group behind the last row is walked by one wavefront, chunk after chunk (0.8 ms for 16 chunks): take the span
                    // near the wanted one that leaves the fewest chunks there
                    int best = span;
                    long long best_left = row_aligned(span) ? n_chunks - ((n_chunks - 1) / span) * span : (1ll << 62);
                    for (int sp = span - span / 4; sp <= span + span / 4; sp++) {
                        const long long left = n_chunks - ((n_chunks - 1) / sp) * sp;
                        if (left < best_left && row_aligned(sp) && (double)sp * max_count < (double)((1u << (32 - idx_bits)) - 2u)) { best = sp; best_left = left; }
                    }
                    span = best;
                }
                if (!row_aligned(span)) span = 1;
                GD.span = span;
            }
        }
    }
    ctx->gardner_mode = use_table ? 1 : 0;
    // stream segment: the sequential samplers go on from the carried state at chunk first / chunk; the symbol buffer starts
    // with the two symbols the Manchester stage looks back on, placed so that local and global symbol parities agree
    SamplerCarry<T> carry_in;
    carry_in.a = 0; carry_in.b = 0; carry_in.c = 0; carry_in.c_first = 0; carry_in.count0 = 0;
    SegTail<T> *d_tail = seg ? (SegTail<T> *)ctx->seg_dev.p : nullptr;
    SamplerCarry<T> *d_carry_out = seg ? &d_tail->sampler : nullptr;
    long long sym_pad = 0;
    unsigned clock0 = 0;
    unsigned long long bit0 = 0;
    long long sync_min_pos = 0;
    if (seg) {
        sym_pad = 2 + (long long)(seg->nsym_total & 1u);
        carry_in.c_first = chunk > 0 ? first / chunk : 0;
        carry_in.count0 = sym_pad;
        if (seg->have_sampler) { carry_in.a = (T)seg->sa; carry_in.b = (T)seg->sb; carry_in.c = (T)seg->sc; }
        T hist[4] = { 0, 0, 0, 0 };
        hist[sym_pad - 2] = (T)seg->sym_m2;
        hist[sym_pad - 1] = (T)seg->sym_m1;
        memcpy(spin + 256, hist, sizeof hist);
        PL.copy(OP_H2D, d_sym, spin + 256, (size_t)sym_pad * sizeof(T));
        if (quality) PL.memset_async(d_symidx, 0, (size_t)sym_pad * sizeof(long long));   // (k_chunk_info searches the picks' indices)
        clock0 = seg->clockmod;
        bit0 = seg->kept_bits.size();
        if (bit0) {
            memcpy(spin + 512, seg->kept_bits.data(), (size_t)bit0);
            PL.copy(OP_H2D, d_bits, spin + 512, (size_t)bit0);
            PL.memset_async(d_bitsym, 0, (size_t)bit0 * sizeof(unsigned));
        }
        sync_min_pos = std::max<long long>(0, seg->next_free - (long long)seg->bit_base);
        // The search treats the bits in front of local bit 0 as the zeros the reference's ring starts with -- true at the start
        // of the stream only.  Later the kept bits are the last len - 1 (or more) of the previous segment: a sync word that ends
        // inside them was looked at there with its real bits, and one that ends further on lies entirely inside this segment.
        if (seg->bit_base > 0) sync_min_pos = std::max<long long>(sync_min_pos, (long long)SP.len - 1);
    }
    if (use_table) {
        if constexpr (std::is_same<T, float>::value) {
            // consistent (q, last pick) combinations: the last pick is rint(q + err), |err| <= 0.1
            std::vector<unsigned> cand;
            std::vector<int> mfirst((size_t)GD.n_q + 1);
            cand.reserve((size_t)GD.n_q * 2);
            for (int m = 0; m < GD.n_q; m++) {
                mfirst[(size_t)m] = (int)cand.size();
                const float q = GD.q_min + (float)m * GD.u;
                const float fr = q - floorf(q);
                if (fr <= 0.61f) cand.push_back((unsigned)(2 * m));
                if (fr >= 0.39f) cand.push_back((unsigned)(2 * m + 1));
            }
            mfirst[(size_t)GD.n_q] = (int)cand.size();
            GD.n_cand = (int)cand.size();
            const long long key = chunk_out * 1000003ll + (long long)llround((double)GP.step * 4096.0);
            if ((rc = ctx->gcand.ensure(cand.size() * sizeof(unsigned)))) return rc;
            if ((rc = ctx->gmfirst.ensure(mfirst.size() * sizeof(int)))) return rc;
            if (ctx->gcand_key != key) {
                HIP_TRY(hipMemcpyAsync(ctx->gcand.p, cand.data(), cand.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemcpyAsync(ctx->gmfirst.p, mfirst.data(), mfirst.size() * sizeof(int), hipMemcpyHostToDevice, st));
                HIP_TRY(hipStreamSynchronize(st));              // the host vectors die at the end of this scope
                ctx->gcand_key = key;
            }
            const long long n_tab = (n_chunks - 1) / GD.span;        // table rows: groups of span full chunks that have a successor
            const long long n_groups = n_tab + 1;                    // ... and the group behind the last row (<= span chunks, the last may be short)
            SamplerCarry<float> tab_carry;                   // where the chain starts: chunk, state, symbols already in the buffer
            tab_carry.a = (float)carry_in.a; tab_carry.b = (float)carry_in.b; tab_carry.c = (float)carry_in.c;
            tab_carry.c_first = carry_in.c_first; tab_carry.count0 = carry_in.count0;
            SamplerCarry<float> chain_carry = tab_carry;     // the chain counts in table rows (groups of GD.span chunks)
            chain_carry.c_first = tab_carry.c_first / GD.span;
            const long long g_first = chain_carry.c_first;
            if ((rc = ctx->gtable.ensure((size_t)n_tab * (size_t)(2 * GD.n_q) * sizeof(unsigned)))) return rc;
            if ((rc = ctx->gentries.ensure((size_t)n_chunks * sizeof(GardnerEntry<float>)))) return rc;
            if ((rc = ctx->gbands.ensure((size_t)n_tab * sizeof(GardnerBand)))) return rc;
            L.begin("gardner_table");
            if ((rc = ctx->gclist.ensure((size_t)n_tab * PDT_GTAB_LIST * sizeof(unsigned)))) return rc;
            // (the table is never initialised as a whole -- 12 GB for an hour at 250 ksps: every look-up checks the chunk's band)
            // the scouts need symbols, not samples, to settle on the chunk's timing: the same number of them at every rate
            const int scout_syms = ctx->tune.scout_syms > 0 ? ctx->tune.scout_syms : 455;
            const int scout_tail = (int)std::min<long long>(PDT_GTAB_TAIL, std::max<long long>(256, (long long)((double)scout_syms * (double)GP.step) + 16));
            PDT_LAUNCH(64, k_gardner_scout, dim3((unsigned)n_tab), dim3(64), 0, st, (const float *)d_agc, GP, GD, n_tab,
                               (const int *)ctx->gmfirst.p, (const unsigned *)ctx->gcand.p, (unsigned *)ctx->gtable.p,
                               (GardnerBand *)ctx->gbands.p, (unsigned *)ctx->gclist.p, d_sc->gstats, scout_tail);
            {
                // locked chunks carry 100-300 candidates that merge quickly: see k_gardner_table_merge
                {
                    const unsigned parts = (unsigned)((GD.n_cand + PDT_GTM_SLOTS - 1) / PDT_GTM_SLOTS);
                    PDT_LAUNCH(PDT_GTM_THREADS, (k_gardner_table_merge<PDT_GTAB_WIN>), dim3((unsigned)n_tab, parts), dim3(PDT_GTM_THREADS), 0, st,
                                       (const float *)d_agc, GP, GD, n_tab, (const unsigned *)ctx->gcand.p,
                                       (const GardnerBand *)ctx->gbands.p, (const unsigned *)ctx->gclist.p,
                                       (unsigned *)ctx->gtable.p, d_sc->gstats);
                }
                if (GD.span > 1) {
                    // the distinct exits of every row's first chunk, walked on through the row's other chunks (see k_gardner_span_keys)
                    const unsigned cap_keys = ctx->tune.gspan_cap ? (unsigned)ctx->tune.gspan_cap
                                                                  : (unsigned)std::min<long long>(std::max<long long>(n_tab * 128, 1ll << 20), 1ll << 28);
                    const size_t cap_items = (size_t)cap_keys / PDT_GSUB_KEYS + (size_t)n_tab + 1;
                    if ((rc = ctx->gspan_keys.ensure((size_t)cap_keys * sizeof(unsigned)))) return rc;
                    if ((rc = ctx->gspan_tails.ensure((size_t)cap_keys * sizeof(unsigned)))) return rc;
                    if ((rc = ctx->gspan_recs.ensure((size_t)cap_keys * (size_t)(GD.span - 1) * sizeof(GardnerSpanRec)))) return rc;
                    if ((rc = ctx->gspan_rows.ensure((size_t)n_tab * sizeof(GardnerSpanRow)))) return rc;
                    if ((rc = ctx->gspan_items.ensure(cap_items * sizeof(GardnerSpanItem)))) return rc;
                    if ((rc = ctx->gspan_ctl.ensure(sizeof(GardnerSpanCtl)))) return rc;
                    PL.memset_async(ctx->gspan_ctl.p, 0, sizeof(GardnerSpanCtl));
                    PDT_LAUNCH(256, k_gardner_span_keys, dim3((unsigned)n_tab), dim3(256), 0, st, GD, n_tab, (const unsigned *)ctx->gcand.p,
                                       (GardnerBand *)ctx->gbands.p, (const unsigned *)ctx->gclist.p, (const unsigned *)ctx->gtable.p,
                                       (unsigned *)ctx->gspan_keys.p, cap_keys, (GardnerSpanRow *)ctx->gspan_rows.p,
                                       (GardnerSpanItem *)ctx->gspan_items.p, (GardnerSpanCtl *)ctx->gspan_ctl.p);
                    const unsigned walkers = (unsigned)std::min<long long>(n_tab / PDT_GSUB + 64, 256ll * 16);     // persistent wavefronts (8 KiB of LDS each)
                    PDT_LAUNCH(64, (k_gardner_span_walk<PDT_GSUB_WIN>), dim3(walkers), dim3(64), 0, st, (const float *)d_agc, GP, GD,
                                       (const unsigned *)ctx->gspan_keys.p, (const GardnerSpanItem *)ctx->gspan_items.p,
                                       (GardnerSpanCtl *)ctx->gspan_ctl.p, (unsigned *)ctx->gspan_tails.p, (GardnerSpanRec *)ctx->gspan_recs.p);
                    PDT_LAUNCH(128, k_gardner_span_join, dim3((unsigned)n_tab), dim3(128), 0, st, GD, n_tab, (const unsigned *)ctx->gcand.p,
                                       (const GardnerBand *)ctx->gbands.p, (const unsigned *)ctx->gclist.p, (unsigned *)ctx->gtable.p,
                                       (const unsigned *)ctx->gspan_keys.p, (const unsigned *)ctx->gspan_tails.p,
                                       (const GardnerSpanRow *)ctx->gspan_rows.p, d_sc->gstats);
                }
            }
            L.end();
            // chunks per chain segment: the chain hops one segment per ~1 us of dependent L2 look-ups, the composite maps and
            // the re-trace of a segment take G such look-ups each but run in parallel over the segments
            // (at most 64: k_gardner_segfill re-traces a segment with one lane per chunk)
            int G = (n_groups < 8000) ? 32 : 64;
            if (ctx->tune.gseg) G = ctx->tune.gseg;
            const long long n_seg = (n_groups + G - 1) / G;
            if ((rc = ctx->gsegmap.ensure((size_t)n_seg * (size_t)(2 * GD.n_q) * sizeof(GardnerSegCell)))) return rc;
            if ((rc = ctx->gsegstart.ensure((size_t)n_seg * sizeof(GardnerSegStart)))) return rc;
            L.begin("gardner_chain");
            PL.memset_async(ctx->gsegstart.p, 0, (size_t)n_seg * sizeof(GardnerSegStart));
            PDT_LAUNCH(1024, k_gardner_segmap, dim3((unsigned)n_seg), dim3(1024), 0, st, (const unsigned *)ctx->gtable.p, GD, n_groups,
                               G, (GardnerSegCell *)ctx->gsegmap.p, (const GardnerBand *)ctx->gbands.p);
            // Long captures: the chain runs range by range; as soon as a range is through, its entry states and symbols are produced
            // on the side stream (segfill, emission -- chip-wide kernels) while the single workgroup of the chain hops on
            // (an hour at 250 ksps: chain 1.6 ms + emission 1.8 ms one after the other -> the emission behind the chain).
            const int n_ranges = (!seg && n_groups >= 8192 && GD.span == 1 && !ctx->tune.chain_one_range) ? 4 : 1;
            if ((rc = ctx->gchain.ensure(sizeof(GardnerChainState)))) return rc;
            const bool side = n_ranges > 1;
            hipStream_t st_emit = side ? ctx->stream2 : st;
            L.end();
            long long c_lo = 0;
            for (int r = 0; r < n_ranges; r++) {
                const long long c_hi = (r == n_ranges - 1) ? n_groups : (n_groups * (r + 1) / n_ranges) / G * G;      // (groups)
                L.begin("gardner_chain");
                PDT_LAUNCH(256, k_gardner_chain, dim3(1), dim3(256), 0, st,      // 4 wavefronts: a walked chunk is staged 4x faster
                                   (const float *)d_agc, GP, GD, n_groups,
                                   (const unsigned *)ctx->gtable.p, (const GardnerSegCell *)ctx->gsegmap.p, G,
                                   (GardnerSegStart *)ctx->gsegstart.p, (GardnerEntry<float> *)ctx->gentries.p, d_sc->gstats,
                                   (const GardnerBand *)ctx->gbands.p, n_tab, chain_carry, (seg && seg->have_sampler) ? 1 : 0, c_hi,
                                   (GardnerChainState *)ctx->gchain.p, r == 0 ? 1 : 0,
                                   (const unsigned *)(GD.span > 1 ? ctx->gspan_keys.p : nullptr), (const unsigned *)(GD.span > 1 ? ctx->gspan_tails.p : nullptr),
                                   (const GardnerSpanRow *)(GD.span > 1 ? ctx->gspan_rows.p : nullptr));
                L.end();
                if (side) PL.simple(OP_FORK);
                const long long s_lo = c_lo / G, s_hi = (c_hi + G - 1) / G;
                L.begin("gardner", st_emit);
                if (s_hi > s_lo)
                    PDT_LAUNCH(64, k_gardner_segfill, dim3((unsigned)(s_hi - s_lo)), dim3(64), 0, st_emit, (const float *)d_agc, GP, GD, n_groups,
                                       (const unsigned *)ctx->gtable.p, G, (const GardnerSegStart *)ctx->gsegstart.p,
                                       (GardnerEntry<float> *)ctx->gentries.p, s_lo);
                // per-chunk emission: small LDS windows so that every chunk of a 10-minute capture is resident at once
                const unsigned char *d_flags = nullptr;
                if (GD.span > 1 && 8 * (long long)GP.step + 256 < PDT_GEMIT_SUB_WIN && !ctx->tune.gemit_groups &&
                    (double)PDT_GEMIT_SUB_WIN / ((double)GP.step - 0.2) + 2.0 < (double)PDT_GSUB_OUT) {
                    // rows of several chunks: four CHUNKS per wavefront (k_gardner_emit_first / _rest); the groups these cannot resolve,
                    // and the one behind the last row, go through the wavefront-per-group kernel below
                    if ((rc = ctx->gcentries.ensure((size_t)n_chunks * sizeof(GardnerEntry<float>)))) return rc;
                    if ((rc = ctx->gflags.ensure((size_t)n_groups + 64))) return rc;
                    PL.memset_async(ctx->gflags.p, 1, (size_t)n_groups, PL.side_of(st_emit));
                    PDT_LAUNCH(64, (k_gardner_emit_first<PDT_GEMIT_SUB_WIN>), dim3((unsigned)((n_tab - g_first + 3) / 4)), dim3(64), 0, st_emit, (const float *)d_agc, GP, GD,
                                       n_tab, (const GardnerEntry<float> *)ctx->gentries.p, (const unsigned *)ctx->gspan_keys.p,
                                       (const GardnerSpanRow *)ctx->gspan_rows.p, (const GardnerSpanRec *)ctx->gspan_recs.p,
                                       (GardnerEntry<float> *)ctx->gcentries.p, (unsigned char *)ctx->gflags.p, (float *)d_sym, d_symidx, sym_cap, g_first);
                    PDT_LAUNCH(64, (k_gardner_emit_rest<PDT_GEMIT_SUB_WIN>), dim3((unsigned)(((n_tab - g_first) * (GD.span - 1) + 3) / 4)), dim3(64), 0, st_emit,
                                       (const float *)d_agc, GP, GD, n_tab, (const GardnerEntry<float> *)ctx->gcentries.p,
                                       (const unsigned char *)ctx->gflags.p, (float *)d_sym, d_symidx, sym_cap, g_first);
                    d_flags = (const unsigned char *)ctx->gflags.p;
                }
                if (c_hi > c_lo)
                    PDT_LAUNCH(256, (k_gardner<float, PDT_GEMIT_LEN, PDT_GEMIT_OUT>), dim3((unsigned)(c_hi - c_lo)), dim3(PDT_GARDNER_THREADS), 0, st_emit,
                                       (const float *)d_agc, (const float *)d_lock, GP, (float *)d_sym, d_symidx, &d_sc->nsym, sym_cap,
                                       (const GardnerEntry<float> *)ctx->gentries.p, tab_carry, (SamplerCarry<float> *)d_carry_out, c_lo, GD.span, d_flags);
                L.end();
                c_lo = c_hi;
            }
            if (side) {
                PL.simple(OP_JOIN_RECORD);
                PL.simple(OP_JOIN_WAIT);
            }
        }
    } else if (use_mm) {
        // MMClockRecovery at the sampler's call site: sequential, one wavefront
        MmParams<T> MP;
        const T rangeT = (T)(ctx->cfg.mm_step_range != 0 ? ctx->cfg.mm_step_range : 3.0);     // ARGOSdemod/main.c:277
        MP.kp = (T)(ctx->cfg.mm_kp != 0 ? ctx->cfg.mm_kp : 0.15);
        MP.step0 = (T)(int)fsi / baud;                                                        // MMClockRecovery.c:20
        MP.step_max = (T)(int)fsi / (baud - rangeT);                                          // :9
        MP.step_min = (T)(int)fsi / (baud + rangeT);                                          // :10
        MP.n_total = n_out;
        MP.chunk_out = chunk_out;
        L.begin("gardner");
        if (seg && !seg->have_sampler) carry_in.b = MP.step0;                 // (the M&M state starts at stepSize = Fs / baud, MMClockRecovery.c:20)
        PDT_LAUNCH(PDT_GARDNER_THREADS, (k_mm<T, 8192, 1024>), dim3(1), dim3(PDT_GARDNER_THREADS), 0, st, (const T *)d_agc, MP, d_sym, d_symidx,
                           &d_sc->nsym, sym_cap, carry_in, seg ? 1 : 0, d_carry_out);
        L.end();
    } else {
        L.begin("gardner");
        constexpr int SMALL_LEN = 32768 / (int)sizeof(T), SMALL_OUT = 1024;    // two 32 KiB windows
        const long long small_need = chunk_out + 2 * (long long)GP.step + 24;
        const double small_syms = (double)chunk_out / ((double)GP.step - 0.25) + 4.0;
        constexpr int RING_LEN = 20480 / (int)sizeof(T), RING_NB = 6, RING_OUT = 256;   // six 20 KiB buffers
        if (small_need <= RING_LEN && small_syms < RING_OUT && chunk_out >= 4 * (long long)GP.step + 8 && n_chunks >= 8 && !seg &&
            !ctx->tune.gardner_onebuf && !ctx->tune.gardner_noring) {
            const size_t need_bytes = ((size_t)n_chunks + 64 + 15) & ~(size_t)15;
            if ((rc = ctx->gneed.ensure(need_bytes + (size_t)n_chunks * sizeof(CalmEntry<T>)))) return rc;
            CalmEntry<T> *d_calm = (CalmEntry<T> *)((char *)ctx->gneed.p + need_bytes);
            PDT_LAUNCH(256, k_chunk_need<T>, dim3((unsigned)n_chunks), dim3(256), 0, st, (const T *)d_agc, n_out, chunk_out, n_chunks,
                               (unsigned char *)ctx->gneed.p);
            // chunks without a walk in one stride: where sampling instants stay multiples of 2^(E - p) below 2^E (k_gardner_ring)
            T gran_scale = 0;
            if (!ctx->tune.gardner_nostride) {
                const double top = (double)chunk_out + 2.0 * (double)GP.step + 2.0;
                const int E = ilogb(top) + 1, p = sizeof(T) == 8 ? 53 : 24;
                const T sc = (T)ldexp(1.0, p - E);
                const T gstep = GP.step * sc;
                if (E < p && gstep == (T)rint((double)gstep) && (double)gstep < ldexp(1.0, p - 1)) gran_scale = sc;
            }
            PDT_LAUNCH(64 * (RING_NB + 1), (k_gardner_ring<T, RING_LEN, RING_NB, RING_OUT>), dim3(1), dim3(64 * (RING_NB + 1)), 0, st,
                               (const T *)d_agc, (const T *)d_lock, GP, (const unsigned char *)ctx->gneed.p, d_sym, d_symidx, &d_sc->nsym,
                               sym_cap, d_calm, gran_scale);
            PDT_LAUNCH(256, k_calm_emit<T>, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, st, (const unsigned char *)ctx->gneed.p,
                               (const CalmEntry<T> *)d_calm, GP, d_sym, d_symidx, sym_cap);
        } else if (small_need <= SMALL_LEN && small_syms < SMALL_OUT && !ctx->tune.gardner_onebuf && !seg)
            PDT_LAUNCH(256, (k_gardner_small<T, SMALL_LEN, SMALL_OUT>), dim3(1), dim3(256), 0, st, (const T *)d_agc, (const T *)d_lock, GP,
                               d_sym, d_symidx, &d_sc->nsym, sym_cap);
        else
            PDT_LAUNCH(256, (k_gardner<T, GardnerLds<T>::LEN, GardnerLds<T>::OUT>), dim3(1), dim3(256), 0, st, d_agc, d_lock, GP, d_sym, d_symidx,
                               &d_sc->nsym, sym_cap, (const GardnerEntry<T> *)nullptr, carry_in, d_carry_out, 0ll, 1, (const unsigned char *)nullptr);
        L.end();
    }

    // ---- Manchester
    L.begin("manchester");
    PDT_LAUNCH(PDT_TILE_THREADS, k_manch_tile<T>, dim3((unsigned)n_tiles), dim3(PDT_TILE_THREADS), 0, st, d_sym, &d_sc->nsym, manch_thr,
                       d_tiles, sym_pad);
    PDT_LAUNCH(1024, k_manch_scan, dim3(1), dim3(1024), 0, st, d_tiles, &d_sc->nsym, &d_sc->nbits, sym_pad, clock0, bit0,
                       seg ? &d_tail->clock : (unsigned *)nullptr);
    PDT_LAUNCH(PDT_TILE_THREADS, k_manch_emit<T>, dim3((unsigned)n_tiles), dim3(PDT_TILE_THREADS), 0, st, d_sym, &d_sc->nsym, manch_thr,
                       d_tiles, d_bits, d_bitsym, bit_cap, sym_pad);
    L.end();

    // ---- byte sync
    L.begin("bytesync");
    launch_bytesync(ctx, PL, st, SP, d_sc, bit_cap, hit_cap, frame_cap, sync_min_pos);
    L.end();
    if (seg) {
        // everything the next segment starts from, in one record
        const long long last_pll = (N > 0) ? (N - 1) / Bp : -1;
        PDT_LAUNCH(256, (k_seg_tail<T, PllSeam<T>, AgcSeam<T>>), dim3(1), dim3(256), 0, st, (const PllSeam<T> *)ctx->seams_pll.p, last_pll,
                   (const T *)d_lock, N, (const AgcSeam<T> *)ctx->seams_agc.p, agc_last_block, (const T *)d_sym, &d_sc->nsym,
                   (const unsigned char *)d_bits, (const unsigned *)d_bitsym, (const long long *)d_symidx, &d_sc->nbits,
                   (long long)bit0, d_tail);
        PL.copy(OP_D2H, spin + 4096, d_tail, sizeof(SegTail<T>));
    }
    ctx->pend_chunks = 0;
    if (inject) {
        // the stage call ends behind the PLL: drop what was recorded for the later stages (their counters stay zero)
        PL.ops.resize(ops_after_pll);
    } else if (quality && n_chunks > 0) {
        // (N > 0 here; a context without a PLL run -- never -- would leave avg_phase 0)
        if (quality_side) {                                  // the averagePhase stream is complete
            PL.simple(OP_JOIN_RECORD);
            PL.simple(OP_JOIN_WAIT);
        }
        PDT_LAUNCH(256, k_chunk_info<T>, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, st, (const T *)d_avgph, N, chunk, n_chunks,
                           interp, (const long long *)d_symidx, (const unsigned long long *)&d_sc->nsym, (const unsigned *)d_bitsym,
                           (const unsigned long long *)&d_sc->nbits, (ChunkInfo *)ctx->chunkinfo.p);
        // (a segment: the new chunks' records; those of the window's history were reported by the segments before)
        const long long rep_first = seg ? first / chunk : 0;
        PL.copy(OP_D2H, ctx->qual_pin, (const ChunkInfo *)ctx->chunkinfo.p + rep_first, (size_t)(n_chunks - rep_first) * sizeof(ChunkInfo));
        ctx->pend_chunks = (uint64_t)(n_chunks - rep_first);
    }
    PL.simple(OP_EV1);

    // ---- results back to the host
    // one synchronisation: the scalars, the lock record and (speculatively, into pinned memory) as many frame
    // records as the previous call of this context produced, plus a margin
    // (the overlapped ingest's segments differ in length: the whole capacity -- about twice what a segment yields, a few MB into
    // pinned memory -- rather than a second, blocking copy between two segments)
    const uint32_t spec_frames = (seg && seg->in_place) ? frame_cap : std::min<uint32_t>(frame_cap, ctx->last_nframes + ctx->last_nframes / 8 + 64);
    if ((size_t)spec_frames * sizeof(FrameRec) > ctx->pinned_cap) {
        if (ctx->pinned) (void)hipHostFree(ctx->pinned);
        ctx->pinned = nullptr;
        ctx->pinned_cap = 0;
        const size_t want = (size_t)spec_frames * sizeof(FrameRec) * 2 + 4096;
        if (timed_host_malloc((void **)&ctx->pinned, want) == hipSuccess) ctx->pinned_cap = want;
        else (void)hipGetLastError();
    }
    const uint32_t got_frames = (ctx->pinned_cap >= (size_t)spec_frames * sizeof(FrameRec)) ? spec_frames : 0u;
    static_assert(sizeof(PllLockInfo<T>) <= 96, "lock record staging");
    PL.copy(OP_D2H, ctx->pend_sc, d_sc, sizeof(DevScalars));
    PL.copy(OP_D2H, ctx->pend_info, d_info, sizeof(PllLockInfo<T>));
    if (got_frames) PL.copy(OP_D2H, ctx->pinned, d_frames, (size_t)got_frames * sizeof(FrameRec));
    ctx->pend_got_frames = got_frames;
    ctx->pend_n = n;
    ctx->pending = true;
    if (phase == RUN_ALL) {
        pdt_ctx *self = ctx;
        if ((rc = execute_plans(&self, 1))) return rc;
    }
    }   // phase != RUN_FINISH
    if (phase == RUN_ENQUEUE) return PDT_OK;
    FinishArgs FA;
    FA.argos = argos; FA.need_lock = need_lock; FA.fuse_mix = fuse_mix;
    FA.N = N; FA.n_out = n_out; FA.chunk = chunk; FA.chunk_out = chunk_out; FA.first = first; FA.first_out = first_out; FA.sym_cap = sym_cap;
    FA.interp = interp; FA.ntaps = ntaps; FA.hit_cap = hit_cap; FA.frame_cap = frame_cap; FA.SP = SP;
    return finish_capture<T>(ctx, n, FA);
}


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
    const size_t need = (size_t)nslots * PDT_INGEST_SPAN;
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
        if (timed_host_malloc((void **)&ctx->ingest_pin, need) != hipSuccess) { (void)hipGetLastError(); return PDT_ERR_NOMEM; }
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
    auto reader = [J, src_c, bytes, nspans, T, PDT_INGEST_SPAN, pin_base](int t) {
        std::atomic<int> &failed = J->failed;
        const IngestSrc &src = src_c;
        int round = 0;
        for (size_t k = (size_t)t; k < nspans && !failed; k += (size_t)T, round++) {
            const int slot = t * PDT_INGEST_SLOTS + (round % PDT_INGEST_SLOTS);
            std::atomic<int> &st = J->slot_state[(size_t)slot];
            while (st.load(std::memory_order_acquire) != 0) {          // its previous copy is still on its way
                if (failed) return;
                std::this_thread::sleep_for(std::chrono::microseconds(10));
            }
            unsigned char *pin = pin_base + (size_t)slot * PDT_INGEST_SPAN;
            const size_t at = k * PDT_INGEST_SPAN;
            const size_t len = std::min(PDT_INGEST_SPAN, bytes - at);
            if (src.mem) {
                memcpy(pin, src.mem + at, len);
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
            st.store(1, std::memory_order_release);
        }
    };
    const bool background = job != nullptr;
    auto submitter = [ctx, J, bytes, dst, nspans, NS, nslots, PDT_INGEST_SPAN, pin_base, background]() {
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
                if (hipMemcpyAsync((unsigned char *)dst + at, pin_base + (size_t)slot * PDT_INGEST_SPAN, len, hipMemcpyHostToDevice, J->cs[q]) != hipSuccess ||
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

// ---------------------------------------------------------------- stage-level entry points
// One stage of the chain on caller data, the reference function's hidden statics as an explicit state record, so that the
// per-chunk dumps of the reference (or of the oracle) can be replayed stage by stage through the kernels the whole-capture
// path uses.
extern "C++" {
template <typename T> static int stage_manchester(pdt_ctx *ctx, const void *sym_host, uint64_t nsym, double thr_d,
                                                  pdt_manchester_state *state, uint8_t *bits_out, uint32_t *bit_symbol_out,
                                                  uint64_t *nbits_out)
{
    pdt_manchester_state fresh;
    memset(&fresh, 0, sizeof fresh);
    if (!state) state = &fresh;
    // the two symbols the decisions look back on go in front, placed so that local and stream symbol parities agree
    const long long pad = 2 + (long long)(state->even_odd & 1u);
    const long long total = pad + (long long)nsym;
    const long long sym_cap = total + 64, bit_cap = sym_cap;
    const long long n_tiles = (sym_cap + PDT_TILE - 1) / PDT_TILE;
    int rc;
    if ((rc = ctx->sym.ensure((size_t)sym_cap * sizeof(T)))) return rc;
    if ((rc = ctx->bits.ensure((size_t)bit_cap))) return rc;
    if ((rc = ctx->bitsym.ensure((size_t)bit_cap * sizeof(unsigned)))) return rc;
    if ((rc = ctx->hits.ensure((size_t)n_tiles * sizeof(ManchTile)))) return rc;
    if ((rc = ctx->scal.ensure(sizeof(DevScalars)))) return rc;
    hipStream_t st = ctx->stream;
    T *d_sym = (T *)ctx->sym.p;
    ManchTile *d_tiles = (ManchTile *)ctx->hits.p;
    DevScalars *d_sc = (DevScalars *)ctx->scal.p;
    DevScalars sc;
    memset(&sc, 0, sizeof sc);
    sc.nsym = (unsigned long long)total;
    T hist[4] = { 0, 0, 0, 0 };
    hist[pad - 2] = (T)state->previous;
    hist[pad - 1] = (T)state->current;
    const T thr = (T)thr_d;
    Plan &PL = ctx->plan;
    PL.clear();
    PL.side_stream = ctx->stream2;
    PL.copy(OP_H2D, d_sc, &sc, sizeof sc);
    PL.copy(OP_H2D, d_sym, hist, (size_t)pad * sizeof(T));
    if (nsym) PL.copy(OP_H2D, d_sym + pad, sym_host, (size_t)nsym * sizeof(T));
    PDT_LAUNCH(PDT_TILE_THREADS, k_manch_tile<T>, dim3((unsigned)n_tiles), dim3(PDT_TILE_THREADS), 0, st, (const T *)d_sym,
               (const unsigned long long *)&d_sc->nsym, thr, d_tiles, pad);
    PDT_LAUNCH(1024, k_manch_scan, dim3(1), dim3(1024), 0, st, d_tiles, (const unsigned long long *)&d_sc->nsym, &d_sc->nbits, pad,
               (unsigned)(state->clockmod & 1u), 0ull, &d_sc->pad0_);
    PDT_LAUNCH(PDT_TILE_THREADS, k_manch_emit<T>, dim3((unsigned)n_tiles), dim3(PDT_TILE_THREADS), 0, st, (const T *)d_sym,
               (const unsigned long long *)&d_sc->nsym, thr, d_tiles, (unsigned char *)ctx->bits.p, (unsigned *)ctx->bitsym.p, bit_cap, pad);
    DevScalars *back = ctx->pend_sc;                                  // pinned
    PL.copy(OP_D2H, back, d_sc, sizeof sc);
    {
        pdt_ctx *self = ctx;
        if ((rc = execute_plans(&self, 1))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(st));
    sc = *back;
    if ((long long)sc.nbits > bit_cap) return PDT_ERR_STATE;
    if (sc.nbits && bits_out) HIP_TRY(hipMemcpy(bits_out, ctx->bits.p, (size_t)sc.nbits, hipMemcpyDeviceToHost));
    if (sc.nbits && bit_symbol_out) {
        HIP_TRY(hipMemcpy(bit_symbol_out, ctx->bitsym.p, (size_t)sc.nbits * sizeof(unsigned), hipMemcpyDeviceToHost));
        for (uint64_t b = 0; b < sc.nbits; b++) bit_symbol_out[b] -= (uint32_t)pad;       // index into this call's symbols
    }
    if (nbits_out) *nbits_out = sc.nbits;
    // ManchesterDecode.c:16-20: the statics after the call
    const T *sy = (const T *)sym_host;
    if (nsym >= 2) { state->previous = (double)sy[nsym - 2]; state->current = (double)sy[nsym - 1]; }
    else if (nsym == 1) { state->previous = state->current; state->current = (double)sy[0]; }
    state->clockmod = sc.pad0_;
    state->even_odd = (uint32_t)((state->even_odd + nsym) & 0xffu);                       // unsigned char evenOddCounter
    memset(ctx->stage_len, 0, sizeof ctx->stage_len);
    return PDT_OK;
}

template <typename T> static int stage_fir(pdt_ctx *ctx, const void *in_host, uint64_t n, pdt_fir_state *state, void *out_host)
{
    pdt_fir_state fresh;
    memset(&fresh, 0, sizeof fresh);
    if (!state) state = &fresh;
    const bool argos = ctx->cfg.mode == PDT_MODE_ARGOS;
    const int interp = (int)ctx->interp, ntaps = (int)ctx->ntaps;
    const int K = argos ? ntaps : ntaps / interp;                      // inputs an output looks back on
    if (K < 1 || K > 64) return PDT_ERR_ARG;
    // POES: the reference's ring keeps input m in slot m mod K and sums the slots in ascending order (LowPassFilter.c:43-70),
    // so the local index of every input must equal its stream index modulo K: p zeros, the K last inputs, the new ones
    const long long p = argos ? 0 : (long long)(state->count % (uint64_t)K);
    const long long lead = p + K;
    const long long N = lead + (long long)n, n_out = N * interp;
    int rc;
    if ((rc = ctx->pll.ensure((size_t)(N + 1) * sizeof(T)))) return rc;
    if ((rc = ctx->fir.ensure((size_t)(n_out + 1) * sizeof(T)))) return rc;
    hipStream_t st = ctx->stream;
    T *d_in = (T *)ctx->pll.p, *d_out = (T *)ctx->fir.p, *d_taps = (T *)ctx->taps.p;
    std::vector<T> head((size_t)lead, (T)0);
    for (int i = 0; i < K; i++) head[(size_t)(p + i)] = (T)state->history[i];
    Plan &PL = ctx->plan;
    PL.clear();
    PL.side_stream = ctx->stream2;
    PL.copy(OP_H2D, d_in, head.data(), (size_t)lead * sizeof(T));
    if (n) PL.copy(OP_H2D, d_in + lead, in_host, (size_t)n * sizeof(T));
    const int opt = 8;
    const long long tile = (long long)PDT_FIR_THREADS * opt;
    const long long grid = (n_out + tile - 1) / tile;
    if (argos) {
        const size_t sh = (size_t)(ntaps + tile + ntaps + 8) * sizeof(T);
        PDT_LAUNCH(PDT_FIR_THREADS, k_fir_plain<T>, dim3((unsigned)grid), dim3(PDT_FIR_THREADS), sh, st, (const T *)d_in, N, ntaps,
                   (const T *)d_taps, d_out, opt);
    } else {
        const size_t sh_rt = (size_t)(65 * (K + 1) + 3 + 64 * K * interp) * sizeof(T);
        const long long tiles_rt = (N + 64ll * K - 1) / (64ll * K);
        const unsigned grid_rt = (unsigned)std::min<long long>(tiles_rt, 256ll * 32);
        bool done = false;
        if (K == 26 && sh_rt <= 64000 && ctx->taps_rot.p && !ctx->tune.fir_generic) {     // the kernel of the whole-capture path
            done = true;
            switch (interp) {
#define PDT_FIR_CASE(I)                                                                                                       \
    case I:                                                                                                                   \
        PDT_LAUNCH(PDT_FIR_THREADS, (k_fir_interp_rt<T, I, 26>), dim3(grid_rt), dim3(PDT_FIR_THREADS), sh_rt, st, (const T *)d_in, N, (const T *)ctx->taps_rot.p, d_out, \
                           (AgcMap *)nullptr, (T)0);                                                                          \
        break;
                PDT_FIR_CASE(1) PDT_FIR_CASE(2) PDT_FIR_CASE(3) PDT_FIR_CASE(4) PDT_FIR_CASE(5) PDT_FIR_CASE(6) PDT_FIR_CASE(7) PDT_FIR_CASE(8)
#undef PDT_FIR_CASE
            default: done = false;
            }
        }
        if (!done) {
            const size_t sh = (size_t)(ntaps + tile / interp + K + 8) * sizeof(T);
            PDT_LAUNCH(PDT_FIR_THREADS, k_fir_interp<T>, dim3((unsigned)grid), dim3(PDT_FIR_THREADS), sh, st, (const T *)d_in, N, interp, K,
                       (const T *)d_taps, d_out, opt);
        }
    }
    {
        pdt_ctx *self = ctx;
        if ((rc = execute_plans(&self, 1))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (n && out_host)
        HIP_TRY(hipMemcpy(out_host, d_out + lead * interp, (size_t)n * (size_t)interp * sizeof(T), hipMemcpyDeviceToHost));
    // the ring after the call: the last K inputs, oldest first
    const T *x = (const T *)in_host;
    double nh[64];
    for (int i = 0; i < K; i++) {
        const long long src = (long long)n - K + i;                   // index into the new inputs, negative = older history
        nh[i] = src >= 0 ? (double)x[src] : state->history[K + src];
    }
    memcpy(state->history, nh, sizeof(double) * (size_t)K);
    state->count += n;
    memset(ctx->stage_len, 0, sizeof ctx->stage_len);
    return PDT_OK;
}

template <typename T> static int stage_pll(pdt_ctx *ctx, const void *iq_host, uint64_t n, int fmt, pdt_pll_state *state, void *out_host,
                                           void *lock_out_host, double *avg_phase_ret)
{
    const size_t fb = fmt == PDT_FMT_F32 ? 8 : 4;                     // bytes per I,Q pair
    pdt_pll_state fresh;
    memset(&fresh, 0, sizeof fresh);
    if (!state) state = &fresh;
    if (n == 0) {                                                     // the loop body never runs; :277 returns the static
        if (avg_phase_ret) *avg_phase_ret = state->started ? state->avg_phase : 0.0;
        return PDT_OK;
    }
    const bool was_locked = state->started && state->locked;
    const uint64_t lead = was_locked ? 1 : 0;                         // a dummy sample in front stands for "locked before sample 0"
    const uint64_t N = n + lead;
    int rc = ctx->pcm.ensure((size_t)N * fb + 16);
    if (rc) return rc;
    HIP_TRY(hipMemset(ctx->pcm.p, 0, 8));
    HIP_TRY(hipMemcpy((char *)ctx->pcm.p + lead * fb, iq_host, (size_t)n * fb, hipMemcpyHostToDevice));
    ctx->pcm_dev = ctx->pcm.p;
    ctx->pcm_fmt = fmt == PDT_FMT_F32 ? 1 : 0;                        // float pairs = `float complex` as they are / int16 pairs, wave.c:127-172
    ctx->inj.active = true;
    ctx->inj.started = state->started != 0;
    ctx->inj.locked = was_locked;
    ctx->inj.phase = state->phase; ctx->inj.freq = state->freq; ctx->inj.avg = state->avg_phase;
    ctx->inj.locksig = state->locksig; ctx->inj.sweep = state->sweep;
    ctx->batch_hint = 1;
    ctx->n_samples = N;
    ctx->n_out = N * ctx->interp;
    rc = run_capture<T>(ctx, N, RUN_ALL);
    ctx->inj.active = false;
    if (rc) return rc;
    // outputs and the statics after the call, from the streams the run left on the device
    if (out_host) HIP_TRY(hipMemcpy(out_host, (const T *)ctx->pll.p + lead, (size_t)n * sizeof(T), hipMemcpyDeviceToHost));
    if (lock_out_host) HIP_TRY(hipMemcpy(lock_out_host, (const T *)ctx->lock.p + lead, (size_t)n * sizeof(T), hipMemcpyDeviceToHost));
    T last_lock = 0, last_avg = 0;
    HIP_TRY(hipMemcpy(&last_lock, (const T *)ctx->lock.p + (N - 1), sizeof(T), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&last_avg, (const T *)ctx->avgph.p + (N - 1), sizeof(T), hipMemcpyDeviceToHost));
    PllLockInfo<T> info;
    HIP_TRY(hipMemcpy(&info, ctx->lockinfo.p, sizeof info, hipMemcpyDeviceToHost));
    state->started = 1;
    if (info.lock_sample < 0) {                                       // still searching: the acquisition's state after the last sample
        state->phase = (double)info.st.phase; state->freq = (double)info.st.freq; state->sweep = (double)info.st.sweep;
    } else {
        if (!was_locked) {
            state->locked = 1;
            state->lock_index = info.lock_sample;
            state->lock_freq_hz = (double)(info.freq_at_lock * (T)ctx->cfg.sample_rate) / (2.0 * M_PI);   // CarrierTrackingPLL.c:269
        }
        state->sweep = (double)info.st.sweep;
        if (info.lock_sample == (long long)N - 1) {                   // locked on the very last sample: nothing walked behind it
            state->phase = (double)info.st.phase; state->freq = (double)info.st.freq;
        } else {
            PllSeam<T> sm;
            const long long Bp = ctx->last_pll_block > 0 ? ctx->last_pll_block : 1;
            HIP_TRY(hipMemcpy(&sm, (const PllSeam<T> *)ctx->seams_pll.p + ((long long)N - 1) / Bp, sizeof sm, hipMemcpyDeviceToHost));
            state->phase = (double)sm.phase1; state->freq = (double)sm.freq1;
        }
    }
    state->locksig = (double)last_lock;
    state->avg_phase = (double)last_avg;
    if (avg_phase_ret) *avg_phase_ret = (double)last_avg;
    memset(ctx->stage_len, 0, sizeof ctx->stage_len);
    ctx->frames_host.clear();
    return PDT_OK;
}

template <typename T> static int stage_gardner(pdt_ctx *ctx, const void *in_host, uint64_t n, uint64_t capacity,
                                               const void *neighbour_host, pdt_gardner_state *state, void *out_host,
                                               uint64_t *pick_out, uint64_t *nsym_out)
{
    pdt_gardner_state fresh;
    memset(&fresh, 0, sizeof fresh);
    if (!state) state = &fresh;
    const bool argos = ctx->cfg.mode == PDT_MODE_ARGOS;
    const int interp = (int)ctx->interp;
    const long long C = (long long)capacity, na = (long long)n;
    // The sampler's reads past the chunk (Q3: the mid-point index is not rolled over with nextSample) see what the caller's
    // buffer still holds behind element n -- for the kernels that is "the previous chunk of the stream": a stream of two
    // chunks of C, the buffer as it stands and then its first n elements, walked from chunk 1 on.
    const long long total = C + na;
    const T Fs = (T)ctx->cfg.sample_rate;
    const T fsi = Fs * (T)interp;
    GardnerParams<T> GP;
    const T baud = ctx->lp.gardner_baud != 0 ? (T)ctx->lp.gardner_baud : argos ? (T)(400 * 2.0) : (T)(8320 * 2 + 0.3);   // main.c:90 / ARGOS main.c:64
    GP.step = (T)(int)fsi / baud;                                              // GardenerClockRecovery.c:19
    GP.kp = (ctx->lp.gardner_kp != 0 || (ctx->lp.zero_mask & PDT_LP_ZERO_GARDNER_KP)) ? (T)ctx->lp.gardner_kp : (T)3.0;
    GP.lim = (ctx->lp.gardner_step_range != 0 || (ctx->lp.zero_mask & PDT_LP_ZERO_GARDNER_STEP_RANGE)) ? (T)ctx->lp.gardner_step_range : (T)0.1;
    GP.n_total = total;
    GP.chunk_out = C;
    GP.argos_heap = 0;
    GP.argos_field_bits = 0;
    GP.argos_even = 0;
    if (argos && neighbour_host && (size_t)C * 8 < 128 * 1024) {              // Q16: the array malloc'd behind this one
        const unsigned long long req = 8ull * (unsigned long long)C;
        GP.argos_heap = 1;
        GP.argos_field_bits = ((req + 8 + 15) & ~15ull) | 1ull;
        GP.argos_even = ((req + 8) % 16) != 0;
    }
    const long long sym_cap = (long long)((double)na / ((double)GP.step - 0.25)) + 64;
    int rc;
    if ((rc = ctx->agc.ensure((size_t)(total + 1) * sizeof(T)))) return rc;
    if ((rc = ctx->lock.ensure((size_t)(total + 1) * sizeof(T)))) return rc;
    if ((rc = ctx->sym.ensure((size_t)sym_cap * sizeof(T)))) return rc;
    if ((rc = ctx->symidx.ensure((size_t)sym_cap * sizeof(long long)))) return rc;
    if ((rc = ctx->scal.ensure(sizeof(DevScalars)))) return rc;
    if ((rc = ctx->seg_dev.ensure(sizeof(SegTail<T>) + 256))) return rc;
    hipStream_t st = ctx->stream;
    T *d_in = (T *)ctx->agc.p, *d_nb = (T *)ctx->lock.p;
    DevScalars *d_sc = (DevScalars *)ctx->scal.p;
    SegTail<T> *d_tail = (SegTail<T> *)ctx->seg_dev.p;
    SamplerCarry<T> carry_in;
    carry_in.a = (T)state->next_sample; carry_in.b = (T)state->prev_bit; carry_in.c = (T)state->half_sample;
    carry_in.c_first = 1; carry_in.count0 = 0;
    Plan &PL = ctx->plan;
    PL.clear();
    PL.side_stream = ctx->stream2;
    PL.memset_async(d_sc, 0, sizeof(DevScalars));
    PL.copy(OP_H2D, d_in, in_host, (size_t)C * sizeof(T));
    if (na) PL.copy(OP_H2D, d_in + C, in_host, (size_t)na * sizeof(T));
    if (GP.argos_heap) {
        PL.copy(OP_H2D, d_nb, neighbour_host, (size_t)C * sizeof(T));
        if (na) PL.copy(OP_H2D, d_nb + C, neighbour_host, (size_t)na * sizeof(T));
    }
    PDT_LAUNCH(256, (k_gardner<T, GardnerLds<T>::LEN, GardnerLds<T>::OUT>), dim3(1), dim3(256), 0, st, (const T *)d_in,
               (const T *)(GP.argos_heap ? d_nb : nullptr), GP, (T *)ctx->sym.p, (long long *)ctx->symidx.p, &d_sc->nsym, sym_cap,
               (const GardnerEntry<T> *)nullptr, carry_in, &d_tail->sampler, 0ll, 1, (const unsigned char *)nullptr);
    DevScalars *back = ctx->pend_sc;                                  // pinned
    PL.copy(OP_D2H, back, d_sc, sizeof(DevScalars));
    {
        pdt_ctx *self = ctx;
        if ((rc = execute_plans(&self, 1))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(st));
    const uint64_t nsym = back->nsym;
    if ((long long)nsym > sym_cap) return PDT_ERR_STATE;
    if (nsym && out_host) HIP_TRY(hipMemcpy(out_host, ctx->sym.p, (size_t)nsym * sizeof(T), hipMemcpyDeviceToHost));
    if (nsym && pick_out) {
        HIP_TRY(hipMemcpy(pick_out, ctx->symidx.p, (size_t)nsym * sizeof(long long), hipMemcpyDeviceToHost));
        for (uint64_t k = 0; k < nsym; k++) pick_out[k] -= (uint64_t)C;                      // index into this call's samples
    }
    if (nsym_out) *nsym_out = nsym;
    SamplerCarry<T> after;
    HIP_TRY(hipMemcpy(&after, &d_tail->sampler, sizeof after, hipMemcpyDeviceToHost));
    state->next_sample = (double)after.a; state->prev_bit = (double)after.b; state->half_sample = (double)after.c;
    memset(ctx->stage_len, 0, sizeof ctx->stage_len);
    return PDT_OK;
}

template <typename T> static int stage_static_gain(pdt_ctx *ctx, const void *iq_host, uint64_t n, int fmt, double level, double *gain_out)
{
    const size_t fb = fmt == PDT_FMT_F32 ? 8 : 4;
    int rc;
    if ((rc = ctx->pcm.ensure((size_t)n * fb + 16))) return rc;
    if ((rc = ctx->mag.ensure((size_t)(n + 1) * sizeof(T)))) return rc;
    if ((rc = ctx->scal.ensure(sizeof(DevScalars)))) return rc;
    hipStream_t st = ctx->stream;
    DevScalars *d_sc = (DevScalars *)ctx->scal.p;
    IqSrc src;
    src.p = ctx->pcm.p;
    src.fmt = fmt == PDT_FMT_F32 ? 1 : 0;
    Plan &PL = ctx->plan;
    PL.clear();
    PL.side_stream = ctx->stream2;
    PL.memset_async(d_sc, 0, sizeof(DevScalars));
    PL.copy(OP_H2D, ctx->pcm.p, iq_host, (size_t)n * fb);
    PDT_LAUNCH(256, k_static_gain<T>, dim3(1), dim3(256), 0, st, src, (long long)n, (T *)ctx->mag.p, (T)level, 0.0, (T *)&d_sc->norm);
    DevScalars *back = ctx->pend_sc;                                  // pinned
    PL.copy(OP_D2H, back, d_sc, sizeof(DevScalars));
    {
        pdt_ctx *self = ctx;
        if ((rc = execute_plans(&self, 1))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(st));
    T g;
    memcpy(&g, &back->norm, sizeof(T));
    if (gain_out) *gain_out = (double)g;
    return PDT_OK;
}

template <typename T> static int stage_mm(pdt_ctx *ctx, const void *in_host, uint64_t n, pdt_mm_state *state, void *out_host,
                                          uint64_t *pick_out, uint64_t *nsym_out)
{
    pdt_mm_state fresh;
    memset(&fresh, 0, sizeof fresh);
    if (!state) state = &fresh;
    if (nsym_out) *nsym_out = 0;
    const bool argos = ctx->cfg.mode == PDT_MODE_ARGOS;
    const T Fs = (T)ctx->cfg.sample_rate;
    const T fsi = Fs * (T)(int)ctx->interp;
    const T baud = ctx->lp.gardner_baud != 0 ? (T)ctx->lp.gardner_baud : argos ? (T)(400 * 2.0) : (T)(8320 * 2 + 0.3);
    MmParams<T> MP;
    const T rangeT = (T)(ctx->cfg.mm_step_range != 0 ? ctx->cfg.mm_step_range : 3.0);     // ARGOSdemod/main.c:277
    MP.kp = (T)(ctx->cfg.mm_kp != 0 ? ctx->cfg.mm_kp : 0.15);
    MP.step0 = (T)(int)fsi / baud;                                                        // MMClockRecovery.c:20
    MP.step_max = (T)(int)fsi / (baud - rangeT);                                          // :9
    MP.step_min = (T)(int)fsi / (baud + rangeT);                                          // :10
    MP.n_total = (long long)n;
    MP.chunk_out = (long long)n;                                                          // one call = one chunk
    if (!state->started) { state->started = 1; state->next_sample = 0; state->step_size = (double)MP.step0; state->sample_last = 0; }
    if (n == 0) return PDT_OK;
    const long long sym_cap = (long long)((double)n / ((double)MP.step_min * 0.999)) + 64;
    int rc;
    if ((rc = ctx->agc.ensure((size_t)(n + 1) * sizeof(T)))) return rc;
    if ((rc = ctx->sym.ensure((size_t)sym_cap * sizeof(T)))) return rc;
    if ((rc = ctx->symidx.ensure((size_t)sym_cap * sizeof(long long)))) return rc;
    if ((rc = ctx->scal.ensure(sizeof(DevScalars)))) return rc;
    if ((rc = ctx->seg_dev.ensure(sizeof(SegTail<T>) + 256))) return rc;
    hipStream_t st = ctx->stream;
    DevScalars *d_sc = (DevScalars *)ctx->scal.p;
    SegTail<T> *d_tail = (SegTail<T> *)ctx->seg_dev.p;
    SamplerCarry<T> carry_in;
    carry_in.a = (T)state->next_sample; carry_in.b = (T)state->step_size; carry_in.c = (T)state->sample_last;
    carry_in.c_first = 0; carry_in.count0 = 0;
    Plan &PL = ctx->plan;
    PL.clear();
    PL.side_stream = ctx->stream2;
    PL.memset_async(d_sc, 0, sizeof(DevScalars));
    PL.copy(OP_H2D, ctx->agc.p, in_host, (size_t)n * sizeof(T));
    PDT_LAUNCH(PDT_GARDNER_THREADS, (k_mm<T, 8192, 1024>), dim3(1), dim3(PDT_GARDNER_THREADS), 0, st, (const T *)ctx->agc.p, MP,
               (T *)ctx->sym.p, (long long *)ctx->symidx.p, &d_sc->nsym, sym_cap, carry_in, 1, &d_tail->sampler);
    DevScalars *back = ctx->pend_sc;                                  // pinned
    PL.copy(OP_D2H, back, d_sc, sizeof(DevScalars));
    {
        pdt_ctx *self = ctx;
        if ((rc = execute_plans(&self, 1))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(st));
    const uint64_t nsym = back->nsym;
    if ((long long)nsym > sym_cap) return PDT_ERR_STATE;
    if (nsym && out_host) HIP_TRY(hipMemcpy(out_host, ctx->sym.p, (size_t)nsym * sizeof(T), hipMemcpyDeviceToHost));
    if (nsym && pick_out) HIP_TRY(hipMemcpy(pick_out, ctx->symidx.p, (size_t)nsym * sizeof(long long), hipMemcpyDeviceToHost));
    if (nsym_out) *nsym_out = nsym;
    SamplerCarry<T> after;
    HIP_TRY(hipMemcpy(&after, &d_tail->sampler, sizeof after, hipMemcpyDeviceToHost));
    state->next_sample = (double)after.a; state->step_size = (double)after.b; state->sample_last = (double)after.c;
    memset(ctx->stage_len, 0, sizeof ctx->stage_len);
    return PDT_OK;
}

template <typename T> static int stage_agc(pdt_ctx *ctx, void *data_host, uint64_t n, double initial, double attack, double decay,
                                           pdt_agc_state *state)
{
    pdt_agc_state fresh;
    memset(&fresh, 0, sizeof fresh);
    if (!state) state = &fresh;
    const bool argos = ctx->cfg.mode == PDT_MODE_ARGOS;
    const int interp = (int)ctx->interp;
    const T Fs = (T)ctx->cfg.sample_rate;
    const T fsi = Fs * (T)interp;                                              // POESTIPdemod/main.c:429
    AgcParams<T> AP;
    AP.attack = attack != 0 ? (T)attack : ctx->lp.agc_attack != 0 ? (T)ctx->lp.agc_attack : (T)(79.5775 * (2.0 * M_PI / (double)fsi));
    AP.decay = decay != 0 ? (T)decay : ctx->lp.agc_decay != 0 ? (T)ctx->lp.agc_decay : (T)(159.1549 * (2.0 * M_PI / (double)fsi));
    AP.squelch = 0;
    AP.squelch_thr = 0;
    AP.raw_out = nullptr;
    const T gain0 = state->started ? (T)state->gain : (T)initial;              // AGC.c:91-95: `initial` counts on the first call only
    state->started = 1;
    if (n == 0) { state->gain = (double)gain0; return PDT_OK; }
    // the whole-capture path's block geometry (any values give the same output)
    const double fs_d = (double)ctx->cfg.sample_rate;
    auto round4 = [](long long v) { return (v + 3) / 4 * 4; };
    long long Ba = ctx->cfg.agc_block ? ctx->cfg.agc_block : (long long)((argos ? 0.125 : 0.0625) * fs_d * interp);
    long long Wa = ctx->cfg.agc_warm ? ctx->cfg.agc_warm : (long long)((argos ? 2.0 : 1.0) * fs_d * interp);
    Ba = std::max<long long>(64, round4(Ba));
    Wa = round4(Wa);
    const long long na = (long long)n, nb = (na + Ba - 1) / Ba, grid = (nb + 63) / 64;
    int rc;
    if ((rc = ctx->fir.ensure((size_t)(na + 1) * sizeof(T)))) return rc;
    if ((rc = ctx->agc.ensure((size_t)(na + 1) * sizeof(T)))) return rc;
    if ((rc = ctx->seams_agc.ensure((size_t)(nb + 1) * sizeof(AgcSeam<T>)))) return rc;
    if ((rc = ctx->agc_maps.ensure((size_t)(nb + 1) * (sizeof(AgcMap) + sizeof(double))))) return rc;
    if ((rc = ctx->scal.ensure(sizeof(DevScalars)))) return rc;
    hipStream_t st = ctx->stream;
    const T *a_in = (const T *)ctx->fir.p;
    T *a_out = (T *)ctx->agc.p;
    DevScalars *d_sc = (DevScalars *)ctx->scal.p;
    T *d_norm = (T *)&d_sc->norm;
    AgcMap *d_maps = (AgcMap *)ctx->agc_maps.p;
    double *d_guess = (double *)(d_maps + nb + 1);
    DevScalars sc;
    memset(&sc, 0, sizeof sc);
    memcpy(&sc.norm, &gain0, sizeof(T));
    double agc_K = (sizeof(T) == 4) ? 11.0 : 34.0;
    if (ctx->tune.agc_k > 0) agc_K = ctx->tune.agc_k;
    Plan &PL = ctx->plan;
    PL.clear();
    PL.side_stream = ctx->stream2;
    PL.copy(OP_H2D, d_sc, &sc, sizeof sc);
    PL.copy(OP_H2D, ctx->fir.p, data_host, (size_t)na * sizeof(T));
    PDT_LAUNCH(256, k_agc_affine<T>, dim3((unsigned)nb), dim3(256), 0, st, a_in, na, AP.decay, Ba, d_maps);
    PDT_LAUNCH(1024, k_agc_guess<T>, dim3(1), dim3(1024), 0, st, (const AgcMap *)d_maps, nb, (const T *)d_norm, d_guess, 1, nb);
    PDT_LAUNCH(64, k_agc_block<T>, dim3((unsigned)grid), dim3(64), 0, st, a_in, na, AP, d_norm, Ba, Wa, (const double *)d_guess,
               (const T *)nullptr, a_out, (AgcSeam<T> *)ctx->seams_agc.p, agc_K, (T *)nullptr);
    PDT_LAUNCH(1024, k_agc_scan<T>, dim3(1), dim3(1024), 0, st, na, Ba, (const AgcSeam<T> *)ctx->seams_agc.p, &d_sc->agc_first_bad);
    PDT_LAUNCH(64, k_agc_fix<T>, dim3(1), dim3(64), 0, st, a_in, na, AP, Ba, (const T *)nullptr, a_out, (AgcSeam<T> *)ctx->seams_agc.p,
               d_sc->counters, (const long long *)&d_sc->agc_first_bad, (T *)nullptr);
    {
        pdt_ctx *self = ctx;
        if ((rc = execute_plans(&self, 1))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(data_host, a_out, (size_t)na * sizeof(T), hipMemcpyDeviceToHost));      // in place, as the reference
    AgcSeam<T> last;
    HIP_TRY(hipMemcpy(&last, (const AgcSeam<T> *)ctx->seams_agc.p + (nb - 1), sizeof last, hipMemcpyDeviceToHost));
    state->gain = (double)last.g1;
    memset(ctx->stage_len, 0, sizeof ctx->stage_len);
    return PDT_OK;
}

template <typename T> static int stage_squelch(pdt_ctx *ctx, void *data_host, const void *lock_host, uint64_t n, double thr)
{
    if (n == 0) return PDT_OK;
    int rc;
    if ((rc = ctx->agc.ensure((size_t)(n + 4) * sizeof(T)))) return rc;
    if ((rc = ctx->lock.ensure((size_t)(n + 4) * sizeof(T)))) return rc;
    hipStream_t st = ctx->stream;
    Plan &PL = ctx->plan;
    PL.clear();
    PL.side_stream = ctx->stream2;
    PL.copy(OP_H2D, ctx->agc.p, data_host, (size_t)n * sizeof(T));
    PL.copy(OP_H2D, ctx->lock.p, lock_host, (size_t)n * sizeof(T));
    PDT_LAUNCH(256, k_squelch<T>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, (T *)ctx->agc.p, (const T *)ctx->lock.p, (long long)n,
               (T)thr);
    {
        pdt_ctx *self = ctx;
        if ((rc = execute_plans(&self, 1))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(data_host, ctx->agc.p, (size_t)n * sizeof(T), hipMemcpyDeviceToHost));
    memset(ctx->stage_len, 0, sizeof ctx->stage_len);
    return PDT_OK;
}
}  // extern "C++"

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
// alone.  Three segments of 64 / 22 / 14 % leave 8.6 ms there (two: 76 / 24 %, 9.7 ms; four equal ones, round 3: 12 ms of
// floor each, 49 ms in all).  (iii) A segment's launch plan is recorded BEFORE the host waits for its last span, and the text
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
