// pdt_rt.h -- internals of libpdt.so shared by its translation units (round 6: the library is four of them, so that the device
// code is compiled side by side -- pdt_api.hip: contexts, ingest, streaming, the C ABI; pdt_chain_f32.hip / pdt_chain_f64.hip: the
// chain's launch recording (run_capture, finish_capture, the stage entries) instantiated for float resp. double, with every
// kernel those launch; pdt_chain_wide_f32.hip / _f64.hip: the PLL kernels' slow-wrap variants).  Not installed, not part of the ABI.
#ifndef PDT_RT_H
#define PDT_RT_H
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

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "libpdt: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, \
                    __LINE__);                                                                         \
            return PDT_ERR_NOGPU;                                                                      \
        }                                                                                              \
    } while (0)

namespace pdtrt {

// host time this process has spent allocating device and pinned memory (pdt_stats.alloc_ms: the cold path's breakdown)
extern std::atomic<long long> g_alloc_ns;              // (defined in pdt_api.hip)
struct AllocTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~AllocTimer() { g_alloc_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
inline hipError_t timed_host_malloc(void **p, size_t bytes)
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
    PllTailList tail_list;       // k_pll_tail_scan -> k_pll_tail: the stretches of open seams behind closed runs
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

// the host side of a launch, one function per (body, block size): Plan::launch stores its address.  The slow-wrap variants of the
// PLL kernels are instantiated in translation units of their own (pdt_chain_wide_f32.hip / _f64.hip; `extern template` where they are used).
template <auto Body, int TB> void go_fn(dim3 g, dim3 b, size_t sh, hipStream_t st, const void *dp)
{
    hipLaunchKernelGGL((k_run<Body, TB>), g, b, sh, st, (const typename BodyTraits<decltype(Body)>::pack *)dp);
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
        o.go = &go_fn<Body, TB>;
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

}  // namespace pdtrt
using namespace pdtrt;

// Developer switches (A/B runs of older kernel variants, tuning sweeps).  The library never reads the environment: the
// switches come from a process-wide registry that only the TEST-ONLY entry pdt_dev_set fills (include/pdt_dev.h; the Python
// binding used by tests/ and bench.py mirrors the PDT_* environment variables into it), and a context takes its copy ONCE,
// when it is opened.
struct Tuning {
    double band_pad = 0.0, pll_warm_scale = 1.0, head_taus = 0.0, agc_k = 0.0, pll_warm_s = 0.0, agc_warm_s = 0.0;
    double overlap_split[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    long long hbm_limit_mb = 0, window_piece = 0;
    int ingest_direct = -1, ingest_numa = -1;        // -1 = decide by probing the file (ingest_capture), 0 = never, 1 = always try
    int scout_syms = 0, gspan = 0, gspan_cap = 0, ingest_threads = 0, ingest_span_mb = 0, ingest_streams = 0, overlap_segments = 0, overlap_min_mb = 0, fir_wg_per_cu = 0, agc_tpb = 0, gseg = 0, pll_block = 0, fix_passes = 2;
    bool fir_generic = false, mix_unfused = false, quality_inline = false, gemit_groups = false, agc_unfused = false, no_excl = false, gardner_onebuf = false, gardner_noring = false, gardner_sequential = false, seg_sequential = false, agc_lanes = false, overlap = true, debug_overlap = false, chain_one_range = false, ema_noguess = false, debug_sync = false, pll_noshort = false, pll_nockpt = false, pll_noconsensus = false, pll_notail = false, seg_plain = false, sync_block = false, gardner_nostride = false;
    void load();                 // (pdt_api.hip: from the registry pdt_dev_set fills)
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
    long long gspan_nrows = 0;         // table rows of several chunks in the last run (0: none) -- pdt_dev_span_rows
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
    int ingest_was_direct = 0, ingest_numa_node = -1;   // how the last ingest read its file (pdt_stats)
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

namespace pdtrt {

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

inline int poes_interp(uint32_t rate) { return (int)rint(150000.0 / (double)(float)rate); }   // POESTIPdemod/main.c:347

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


// (pdt_api.hip)
int execute_plans(pdt_ctx *const *ctxs, int M);

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

inline uint32_t next_pow2(uint32_t v)
{
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

inline SyncParams make_sync_params(bool argos, bool argos_twin = false)
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

// (pdt_api.hip)
void launch_bytesync(pdt_ctx *ctx, Plan &PL, hipStream_t st, const SyncParams &SP, DevScalars *d_sc, long long bit_cap, uint32_t hit_cap,
                     uint32_t frame_cap, long long min_pos = 0);
void chunk_reports_range(const pdt_ctx *ctx, uint64_t c0, uint64_t c1, uint64_t total, pdt_chunk_report *out, const pdt_frame *open_frame);

// What run_capture's recording phase hands to its finish phase (finish_capture): the capacities and flags the read-back needs.
struct FinishArgs {
    bool argos, need_lock, fuse_mix;
    long long N, n_out, chunk, chunk_out, first, first_out, sym_cap;
    int interp, ntaps;
    uint32_t hit_cap, frame_cap;
    SyncParams SP;
};

// phase: the whole call, or split for the batched entry point -- enqueue every kernel and the read-back copies of
// one capture (no host synchronisation), later wait for them and build the host-side results
enum { RUN_ALL = 0, RUN_ENQUEUE = 1, RUN_FINISH = 2 };

// ---- the chain (pdt_chain.inc), instantiated for float in pdt_chain_f32.hip and for double in pdt_chain_f64.hip
template <typename T> int finish_capture(pdt_ctx *ctx, uint64_t n, const FinishArgs &FA);
template <typename T> int run_capture(pdt_ctx *ctx, uint64_t n, int phase = RUN_ALL);
template <typename T> int stage_manchester(pdt_ctx *ctx, const void *sym_host, uint64_t nsym, double thr_d, pdt_manchester_state *state,
                                           uint8_t *bits_out, uint32_t *bit_symbol_out, uint64_t *nbits_out);
template <typename T> int stage_fir(pdt_ctx *ctx, const void *in_host, uint64_t n, pdt_fir_state *state, void *out_host);
template <typename T> int stage_pll(pdt_ctx *ctx, const void *iq_host, uint64_t n, int fmt, pdt_pll_state *state, void *out_host,
                                    void *lock_out_host, double *avg_phase_ret);
template <typename T> int stage_gardner(pdt_ctx *ctx, const void *in_host, uint64_t n, uint64_t capacity, const void *neighbour_host,
                                        pdt_gardner_state *state, void *out_host, uint64_t *pick_out, uint64_t *nsym_out);
template <typename T> int stage_static_gain(pdt_ctx *ctx, const void *iq_host, uint64_t n, int fmt, double level, double *gain_out);
template <typename T> int stage_mm(pdt_ctx *ctx, const void *in_host, uint64_t n, pdt_mm_state *state, void *out_host, uint64_t *pick_out,
                                   uint64_t *nsym_out);
template <typename T> int stage_agc(pdt_ctx *ctx, void *data_host, uint64_t n, double initial, double attack, double decay,
                                    pdt_agc_state *state);
template <typename T> int stage_squelch(pdt_ctx *ctx, void *data_host, const void *lock_host, uint64_t n, double thr);
// every instantiation of the chain, as `template` (the unit that defines them) or `extern template` (everybody else)
#define PDT_CHAIN_INSTANCES(KW, T)                                                                                                          \
    KW template int finish_capture<T>(pdt_ctx *, uint64_t, const FinishArgs &);                                                             \
    KW template int run_capture<T>(pdt_ctx *, uint64_t, int);                                                                               \
    KW template int stage_manchester<T>(pdt_ctx *, const void *, uint64_t, double, pdt_manchester_state *, uint8_t *, uint32_t *, uint64_t *); \
    KW template int stage_fir<T>(pdt_ctx *, const void *, uint64_t, pdt_fir_state *, void *);                                               \
    KW template int stage_pll<T>(pdt_ctx *, const void *, uint64_t, int, pdt_pll_state *, void *, void *, double *);                        \
    KW template int stage_gardner<T>(pdt_ctx *, const void *, uint64_t, uint64_t, const void *, pdt_gardner_state *, void *, uint64_t *, uint64_t *); \
    KW template int stage_static_gain<T>(pdt_ctx *, const void *, uint64_t, int, double, double *);                                         \
    KW template int stage_mm<T>(pdt_ctx *, const void *, uint64_t, pdt_mm_state *, void *, uint64_t *, uint64_t *);                         \
    KW template int stage_agc<T>(pdt_ctx *, void *, uint64_t, double, double, double, pdt_agc_state *);                                     \
    KW template int stage_squelch<T>(pdt_ctx *, void *, const void *, uint64_t, double);
// the slow-wrap variants of the PLL kernels (loop gains so large that one step may move the phase by 2 pi: a caller's own loop
// constants) are the largest kernels of the library by far and compile for as long as everything else of a precision together
#define PDT_WIDE_INSTANCES(KW, T)                                                                                  \
    KW template void go_fn<&k_pll_phase<T, true>, 256>(dim3, dim3, size_t, hipStream_t, const void *);             \
    KW template void go_fn<&k_pll_acquire_pipe<T, true>, 128>(dim3, dim3, size_t, hipStream_t, const void *);      \
    KW template void go_fn<&k_pll_head<T, true>, 64>(dim3, dim3, size_t, hipStream_t, const void *);               \
    KW template void go_fn<&k_pll_fix<T, true>, PDT_FIX_THREADS>(dim3, dim3, size_t, hipStream_t, const void *);     \
    KW template void go_fn<&k_pll_tail<T, true>, 64>(dim3, dim3, size_t, hipStream_t, const void *);
}  // namespace pdtrt
#endif
