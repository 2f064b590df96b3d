/*
 * include/pdt.h -- C ABI of libpdt.so, the MI355X (gfx950) implementation of the
 * POES-TIP / ARGOS IQ demodulation chain of nebarnix/Project-Desert-Tortoise.
 *
 * The reference has no FFI layer: its boundary is (i) the demodPOES/demodARGOS
 * command lines with their minorFrames_*.txt / packets_*.txt files and (ii) the
 * C functions of common/ that POESTIPdemod/main.c and ARGOSdemod/main.c call
 * once per 10 000-sample chunk.  A GPU cannot usefully be entered once per
 * chunk per stage, so this ABI sits one level up: a re-entrant context receives
 * a whole capture (or a large span of it), emulates the reference's chunked
 * semantics internally (chunk size is observable in the output)
 * and hands back decoded frame records; the host program formats the text
 * exactly like POESTIPdemod/ByteSync.c:62-69,96-101 / ARGOSdemod/ByteSync.c:62-70,99-103.
 * Per-stage read-back (pdt_read_stage) exposes every intermediate stream that
 * the reference passes between its stage functions so each stage can be
 * compared with the reference's own output.
 *
 * Plain C types only; caller-owned memory; integer error codes; no globals:
 * one context per capture, contexts are independent (one per GPU / stream).
 *
 * Reference interface replaced by each entry point:
 *   pdt_open            the lazy first-call initialisation of every stage
 *                       (CarrierTrackingPLL.c:88-100, LowPassFilter.c:30-40, AGC.c:92-96,
 *                       GardenerClockRecovery.c:17-21, ByteSync.c:28-39) plus the
 *                       constants at the call sites POESTIPdemod/main.c:346-369,413-454,
 *                       ARGOSdemod/main.c:248-284
 *   pdt_demod_pcm16     the while(!feof) chunk loop POESTIPdemod/main.c:373-492 /
 *                       ARGOSdemod/main.c:250-306 over GetComplexWaveChunk (wave.c:59-175),
 *                       StaticGain (AGC.c:48-75), CarrierTrackPLL (CarrierTrackingPLL.c:54-278),
 *                       LowPassFilterInterp / LowPassFilter (LowPassFilter.c:13-71,76-125),
 *                       NormalizingAGC (AGC.c:78-132), Squelch (AGC.c:24-46),
 *                       GardenerClockRecovery (GardenerClockRecovery.c:5-114) or, on request,
 *                       MMClockRecovery (MMClockRecovery.c:5-83),
 *                       ManchesterDecode (ManchesterDecode.c:10-100),
 *                       ByteSyncOnSyncword (POESTIPdemod/ByteSync.c:16-150) /
 *                       FindSyncWords (ARGOSdemod/ByteSync.c:17-150)
 *   pdt_demod_fd        same, reading the capture file itself: the fread loops of GetComplexWaveChunk / GetComplexRawChunk
 *                       (wave.c:126-172, 483-537) become a threaded read into pinned memory overlapped with the copy to HBM
 *   pdt_demod_device    same, input already resident in HBM (bench / multi-capture)
 *   pdt_demod_f32       the same loop over GetComplexRawChunk (wave.c:413-540): RAW float32 captures
 *   pdt_stream_*        the same loop fed block by block, as POESTIPdemodPortAudio/main.c:324-393 is by
 *                       Pa_ReadStream (the sound-card side itself is out of scope); the function-local statics that carry
 *                       each stage from chunk to chunk there (CarrierTrackingPLL.c:60-75, LowPassFilter.c:21-27, AGC.c:83-84,
 *                       GardenerClockRecovery.c:11-15, ManchesterDecode.c:16-21, ByteSync.c:18-22) are the carried state here
 *   pdt_frames          the fprintf stream of ByteSync.c, as records
 *   pdt_format_frames   the text ByteSync.c writes to the output file
 *   pdt_tip_check       the downstream frame validation the reference keeps in MATLAB:
 *                       standalone_matlab/Functionized/checkParity.m:1-92 (five even-parity checks per TIP
 *                       minor frame) and daytimeDecode.m:1-40 (minor-frame counter, spacecraft id, day,
 *                       millisecond of day, T0) -- SURVEY 8f #2
 *   pdt_make_lpf        MakeLPFIR (LowPassFilter.c:127-175)
 *   pdt_wav_parse_header ReadWavHeader (wave.c:303-378)
 */
#ifndef PDT_H
#define PDT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4 (round 6): a capture that does not fit the device's free memory is demodulated through a bounded window (the streaming
 * path, fed from the file or the caller's memory piece by piece) instead of failing with PDT_ERR_NOMEM -- the reference's chunk
 * loop needs O(chunk) memory whatever the file's length (POESTIPdemod/main.c:373); pdt_stats.segments / .windowed say how a
 * capture was taken; pdt_loop_params.zero_mask; a context whose caller asked for the PLL stream (pdt_keep_pll(ctx, 1)) is
 * never demodulated in overlapped segments.
 * 3 (round 5): + pdt_demod_file (capture file in, frame text out, in one call); pdt_demod_fd / pdt_demod_file demodulate a
 * large POES file in segments while it is still being read BY DEFAULT again (round 4: only on request), after which
 * pdt_read_stage / pdt_stage_len describe the last segment only.
 * 2 (round 4): + pdt_write_frames / pdt_write_records.  Behaviour a client of version 1 should know about, all of it
 * introduced under version 1 in round 3 without a bump: pdt_build_tag, pdt_keep_pll, pdt_stage_bytesync_from and the error
 * code PDT_ERR_IO exist; every pdt_demod_* and pdt_stage_* entry returns PDT_ERR_STATE while a stream is open, and the first
 * pdt_stream_push_* opens one by itself (resetting frames and statistics).                                              */
#define PDT_ABI_VERSION 4

enum { PDT_MODE_POES = 0, PDT_MODE_ARGOS = 1 };
enum { PDT_SAMPLER_GARDNER = 0, PDT_SAMPLER_MM = 1 };

enum {
    PDT_OK = 0,
    PDT_ERR_ARG = -1,       /* bad argument                                  */
    PDT_ERR_NOGPU = -2,     /* no HIP device / HIP runtime failure           */
    PDT_ERR_NOMEM = -3,
    PDT_ERR_FORMAT = -4,    /* unsupported WAV format                        */
    PDT_ERR_RATE = -5,      /* sample rate gives interpolation factor 0 (Fs > 300 kHz, POESTIPdemod/main.c:347) */
    PDT_ERR_STATE = -6,     /* call sequence error                           */
    PDT_ERR_IO = -7         /* read error on the capture file (pdt_demod_fd) */
};

/* intermediate streams, in the order the reference produces them */
enum {
    PDT_ST_PLL = 0,     /* realDataOut of CarrierTrackPLL            DT per input sample        */
    PDT_ST_LOCK,        /* lockSignalStreamOut (ARGOS only)          DT per input sample        */
    PDT_ST_FIR,         /* low-pass output                           DT per interpolated sample */
    PDT_ST_AGC,         /* NormalizingAGC (+Squelch, ARGOS) output   DT per interpolated sample */
    PDT_ST_SYM,         /* Gardner symbols                           DT per symbol              */
    PDT_ST_SYMIDX,      /* global interpolated-sample index each symbol was taken at, int64    */
    PDT_ST_BITS,        /* Manchester bits, '0'/'1'                  uint8 per bit              */
    PDT_ST_BITSYM,      /* global symbol index each bit's time stamp comes from, uint32         */
    PDT_ST_AGC_RAW,     /* NormalizingAGC output BEFORE Squelch (only after pdt_keep_presquelch): what ARGOSdemod -r
                           writes to output.raw (ARGOSdemod/main.c:273-274); equal to PDT_ST_AGC for POES     */
    PDT_ST_COUNT
};

enum { PDT_CHAIN_FILE = 0, PDT_CHAIN_LIVE = 1 };
enum { PDT_FMT_PCM16 = 0, PDT_FMT_F32 = 1 };   /* interleaved little-endian int16 I,Q pairs / IEEE float32 I,Q pairs */

typedef struct pdt_config {
    int32_t  mode;            /* PDT_MODE_POES / PDT_MODE_ARGOS                                   */
    uint32_t sample_rate;     /* Hz, the WAV header value (after the -s override, Q6)             */
    uint64_t chunk;           /* reference chunk size in input samples; 0 = 10000 / 2400          */
    double   norm_override;   /* -n option; 0 = StaticGain of the first chunk                     */
    int32_t  device;          /* HIP device ordinal                                               */
    int32_t  profile;         /* 1 = bracket every kernel with HIP events (pdt_kernel_times)      */
    /* Block-parallel evaluation of the PLL / AGC recurrences: each block replays
     * `warm` samples before its own `block` samples; seams are validated bitwise
     * and re-run sequentially on mismatch, so results never depend on these.
     * 0 = defaults derived from the sample rate.                                                  */
    uint32_t pll_block, pll_warm, agc_block, agc_warm;
    /* Gardner boundary-state tables: candidate entry states are tabulated within this many samples of
     * the end point of each of 64 scout trajectories (0 = default 1/8; the true state was further
     * from every scout in ~1 chunk per 9000 of the test captures; candidates merge within a few hundred
     * symbols, so a wider pad costs little).  Smaller = less work, more chunks
     * walked serially; the result never depends on it.                                            */
    double   gardner_band_pad;
    /* Symbol sampler: 0 = GardenerClockRecovery (what the reference runs); 1 = MMClockRecovery
     * (common/MMClockRecovery.c:5-83) at the same call site -- the switch the reference keeps commented out
     * (ARGOSdemod/main.c:277) -- with stepRange / kp below (0 = that call's values, 3 and 0.15).          */
    int32_t  sampler;
    /* Chain variant: 0 = the file programs (POESTIPdemod / ARGOSdemod, what BASELINE measures); 1 = the sound-card
     * twin POESTIPdemodPortAudio (main.c:41-65,324-393): same stage functions, acquisition gain 198.9437, lock threshold
     * 0.10, the lock signal kept and Squelch(0.05) applied between PLL and FIR, Manchester threshold 0.75; default
     * chunk 2400 (its block size).  Feed it float32 frames at 48 kHz to be the twin.  POES only (the ARGOS twin is
     * the float build of the ARGOS chain): PDT_ERR_ARG otherwise.                                                     */
    int32_t  chain;
    double   mm_step_range, mm_kp;
} pdt_config;

typedef struct pdt_frame {
    double   time;        /* time stamp the reference prints with "%.5f"                          */
    int64_t  bit_index;   /* global index of the bit that completed the sync word                 */
    int64_t  time_src;    /* global interpolated-sample index the time stamp derives from         */
    uint8_t  inverted;    /* found through the inverse sync word ("i" suffix, POES only)          */
    uint8_t  nbytes;      /* bytes present: 104 (POES) / 7 (ARGOS) when complete                  */
    uint8_t  complete;    /* all bytes seen (newline emitted)                                     */
    uint8_t  pad;
    uint8_t  bytes[104];
} pdt_frame;

typedef struct pdt_stats {
    uint64_t samples, out_samples, symbols, bits, frames;
    int64_t  lock_sample;         /* global sample index of the one-time PLL lock, -1 = never     */
    double   lock_freq_hz;        /* " : PLL locked at %0.2fHz"                                   */
    double   norm_factor;         /* "Normalization Factor: %f"                                   */
    double   avg_phase;           /* CarrierTrackPLL return value at the lock sample (quality)    */
    uint32_t interp, ntaps;
    uint32_t pll_blocks, pll_seam_fixes, agc_blocks, agc_seam_fixes;
    double   gpu_ms;              /* device time of the last pdt_demod_* call (HIP events)        */
    uint32_t gardner_parallel;    /* 1 = symbol sampler ran through the parallel boundary-state tables,
                                     0 = single-wavefront sequential chain                         */
    uint32_t gardner_walked;      /* chunks whose entry state was outside the tabulated band (walked by the chain) */
    uint32_t gardner_full_domain; /* chunks tabulated over the full boundary-state domain (scouts not locked)       */
    uint32_t sync_overflow;       /* 4096-bit tiles with more than 31 sync hits (generic path used)                */
    uint64_t gardner_candidates;  /* boundary states evaluated by the table kernel                                  */
    double   ingest_ms;           /* (ABI 3) host wall time from the call's start until the capture's last byte had been queued
                                     for the copy to HBM (pdt_demod_fd / _file / _pcm16 / _f32; 0: input was resident)     */
    double   alloc_ms;            /* (ABI 3) host time this PROCESS has spent in device / pinned allocations so far (the cold
                                     path's breakdown: a context's first capture of a size pays for its buffers)          */
    uint32_t segments;            /* (ABI 4) pieces the last capture was demodulated in: 1 = one piece; more (the overlapped
                                     ingest of a large file, the bounded window): pdt_read_stage / pdt_stage_len, the seam
                                     counters and pdt_kernel_times describe the LAST piece only                            */
    uint32_t windowed;            /* (ABI 4) 1 = the capture did not fit the device's free memory and went through the
                                     bounded window (same frames, text, counts and per-chunk reports)                      */
    uint32_t ingest_direct;       /* (ABI 4) 1 = the capture file was read with O_DIRECT straight into the pinned staging (its
                                     pages were not in the page cache, the file system offers it): two passes over host memory
                                     per byte instead of three                                                              */
    int32_t  ingest_numa_node;    /* (ABI 4) NUMA node the reader threads and the staging memory were bound to (the GPU's, when
                                     the file's cached pages lie there or it was read directly); -1 = not bound             */
} pdt_stats;

typedef struct pdt_kernel_time {
    char     name[32];
    uint32_t launches;
    double   total_ms;
} pdt_kernel_time;

typedef struct pdt_ctx pdt_ctx;

int  pdt_abi_version(void);
/* Identifies the build: the first 12 hex digits of the SHA-1 over the library's sources (csrc/, include/pdt.h), set by the
 * Makefile.  Profiles committed under profiles/ record it, so that a figure measured with another build is never quoted. */
const char *pdt_build_tag(void);
const char *pdt_strerror(int code);
int  pdt_device_count(void);

int  pdt_open(const pdt_config *cfg, pdt_ctx **out);

/* The loop constants of the chain (ABI 3).  The reference's stage functions take them as arguments -- CarrierTrackPLL(...,
 * Fs, freqRange, d_lock_threshold, lockSigAlpha, loopbw_acq, loopbw_track) (CarrierTrackPLL.h:11), NormalizingAGC(..., attack_rate,
 * decay_rate) (AGC.h:7), GardenerClockRecovery(..., baud, stepRange, kp) (GardenerClockRecovery.h:3), ManchesterDecode(...,
 * resyncThreshold) (ManchesterDecode.h:3) -- and its mains pass the values a context uses by default (POESTIPdemod/main.c:413,429,
 * 438,445; ARGOSdemod/main.c:265,276-282).  A field left 0 keeps that default; the others replace it in every later
 * pdt_demod_* / pdt_stream_* / pdt_stage_* call of the context, as the value of the context's DECIMAL_TYPE the reference's
 * function would have received.  Nothing else changes: the block-parallel evaluation derives its warm-up lengths from the loop
 * gains it is given and validates every seam bitwise, so the result is the sequential loop's for any constants (only the time
 * moves: loops that contract slowly mean longer warm-ups and more repairs).  pdt_stage_agc's own rate arguments still win
 * over agc_attack / agc_decay.  PDT_ERR_ARG: a negative or non-finite value, a baud rate that leaves fewer than two samples per
 * symbol, a frequency range of half the sample rate or more, a step range above the mains' 0.1 (the samplers' windows and symbol
 * capacities are sized for it).  PDT_ERR_STATE while a stream is open.                                                                          */
typedef struct pdt_loop_params {
    double pll_freq_range_hz;      /* frequency limit of the loop, Hz                                  4500 / 550        */
    double pll_lock_threshold;     /* lock detector threshold                                         0.08 (twin 0.10) / 0.1 */
    double pll_lock_alpha;         /* lockSigAlpha, per sample                                        0.3979 w / 3.1831 w, w = 2 pi / Fs */
    double pll_loopbw_acq;         /* loop bandwidth before the lock, radians per sample              127.3240 w (twin 198.9437 w) / 16 w */
    double pll_loopbw_track;       /* ... after it                                                    10.3451 w / 16 w  */
    double agc_attack, agc_decay;  /* NormalizingAGC rates, per output sample                         79.5775 w', 159.1549 w', w' = 2 pi / (Fs interp) */
    double gardner_baud;           /* symbols per second (Manchester half bits)                       16640.3 / 800     */
    double gardner_step_range;     /* clip of the timing error, at most 0.1                           0.1               */
    double gardner_kp;             /* timing loop gain                                                3.0               */
    double manchester_threshold;   /* resyncThreshold                                                 1.0 (twin 0.75) / 0.5 */
    uint32_t zero_mask;            /* (ABI 4) PDT_LP_ZERO_* bits: the field IS zero (not "keep the default") -- a lock threshold of 0
                                      (lock on the first positive detector value), a timing gain or clip of 0 (open-loop
                                      sampler), a resync threshold of 0.  The other constants must be positive to mean anything */
    uint32_t reserved_;
} pdt_loop_params;
enum { PDT_LP_ZERO_LOCK_THRESHOLD = 1, PDT_LP_ZERO_GARDNER_KP = 2, PDT_LP_ZERO_GARDNER_STEP_RANGE = 4, PDT_LP_ZERO_MANCHESTER_THRESHOLD = 8 };
int  pdt_set_loop_params(pdt_ctx *ctx, const pdt_loop_params *params);
int  pdt_get_device(const pdt_ctx *ctx);          /* the HIP device ordinal the context lives on */
void pdt_close(pdt_ctx *ctx);

/* Use an existing HIP stream (hipStream_t passed as void*) for all work; NULL = own stream. */
int  pdt_set_stream(pdt_ctx *ctx, void *hip_stream);
/* Also keep the AGC output before Squelch (stage PDT_ST_AGC_RAW) in the following pdt_demod_* calls: the stream the
 * reference's `-r` option dumps (ARGOSdemod/main.c:171-180,273-274).  Costs one more stream-sized buffer.            */
int  pdt_keep_presquelch(pdt_ctx *ctx, int enable);

/* Whether the PLL output stream (stage PDT_ST_PLL = CarrierTrackPLL's realDataOut, CarrierTrackingPLL.c:113) is written out in
 * the following pdt_demod_* calls.  At INTERP 1 (sample rates from 150 ksps up) the float chain mixes and filters in one kernel
 * (k_mix_fir): nothing but the filter reads that stream, and with enable = 0 it never goes through HBM -- pdt_stage_len /
 * pdt_read_stage of PDT_ST_PLL then report 0 samples / PDT_ERR_ARG.  On by default (every stage readable after a call); the
 * host programs and the benchmark switch it off.  No effect on the other chains, which always keep the stream.           */
int  pdt_keep_pll(pdt_ctx *ctx, int enable);

/* Also keep what the reference's chunk loop knows after every chunk, in the following pdt_demod_* calls (whole captures; not
 * the streaming entry points): the value CarrierTrackPLL returns for the chunk -- averagePhase, the running mean of
 * |arg(PLL output)| with alpha 0.00005, after the chunk's last sample (CarrierTrackingPLL.c:80,124,152,277), from which
 * POESTIPdemod/main.c:461-481 prints its quality figure 10 log10((pi/2 - averagePhase)^2) -- and the return values of the
 * sampler, the Manchester decoder and the byte synchroniser for that chunk (main.c:438,445,454-460), i.e. everything the
 * "\r" progress line shows.  Before the lock the acquisition kernel delivers averagePhase as it goes; after it, one more
 * block-parallel EMA over the whole capture (same scheme and the same bit-for-bit guarantee as the lock detector's stream).
 * Costs two stream-sized buffers and the EMA walkers' time (16 time constants of 20 000 samples per block: ~6 ms whatever the
 * capture's length).  The overlapped ingest of large captures (pdt_demod_fd / pdt_demod_file) keeps the reports too: its
 * segments end on chunk boundaries, averagePhase and the counts are carried from one to the next.  Off by default.        */
int  pdt_keep_quality(pdt_ctx *ctx, int enable);
typedef struct pdt_chunk_report {
    uint64_t samples;     /* nSamples of the chunk (the last one may be short)                                          */
    double   avg_phase;   /* CarrierTrackPLL's return value for this chunk, exactly (float widened for POES)            */
    uint64_t symbols;     /* GardenerClockRecovery / MMClockRecovery return value                                       */
    uint64_t bits;        /* ManchesterDecode return value                                                              */
    uint64_t frames;      /* ByteSyncOnSyncword / FindSyncWords return value (sync words found in this chunk's bits)    */
    double   time0;       /* waveDataTime[0] when the progress line is printed (ARGOS: after the in-place compactions)  */
} pdt_chunk_report;
/* Per-chunk reports of the last pdt_demod_* call, in chunk order; returns the number copied (out == NULL: the number
 * available; 0 unless pdt_keep_quality was on).                                                                          */
uint64_t pdt_chunk_reports(const pdt_ctx *ctx, pdt_chunk_report *out, uint64_t max_chunks);
/* The same reports WHILE a call runs -- the reference prints its progress line after every chunk (POESTIPdemod/main.c:457-481).
 * With pdt_keep_quality on, `fn` is handed the reports of chunks [first_chunk, first_chunk + n) as soon as they are final:
 * once per completed segment of an overlapped pdt_demod_fd / pdt_demod_file (from a thread of the library, in chunk order,
 * never two calls at a time, the last one before the demod call returns), once at the end of any other whole-capture call
 * (from the caller's thread).  `so_far`: pdt_get_stats as of the last of these chunks (norm_factor from the first chunk on,
 * lock_sample / lock_freq_hz once the PLL has locked: what the reference prints between its progress lines, main.c:420,
 * CarrierTrackingPLL.c:269).  Both pointers are only valid during the call; the callback must not call into the context.
 * fn == NULL switches it off.                                                                                            */
typedef void (*pdt_progress_fn)(void *user, uint64_t first_chunk, const pdt_chunk_report *reports, uint64_t n, const pdt_stats *so_far);
int  pdt_set_progress(pdt_ctx *ctx, pdt_progress_fn fn, void *user);

/* Demodulate one whole capture: nframes interleaved little-endian int16 I,Q pairs
 * in host memory (copied to the GPU) ...                                                        */
int  pdt_demod_pcm16(pdt_ctx *ctx, const int16_t *iq_host, uint64_t nframes);
/* ... or straight from the file the caller opened (what GetComplexWaveChunk / GetComplexRawChunk do with their FILE*,
 * wave.c:59-175,413-540, once per chunk): nframes I,Q pairs of `sample_format` starting at byte_offset (44 for the
 * canonical WAV header ReadWavHeader accepts, 0 for RAW).  The library reads the file in 2 MiB spans with a few host
 * threads into pinned memory and copies them to the GPU while the next spans are being read, so a capture is in HBM about
 * as soon as the page cache and the PCIe link allow.  PDT_ERR_FORMAT when the file ends early, PDT_ERR_IO on a read error
 * (EINTR is retried).
 * Large POES captures (2.5 GiB and more -- below that the segments' latency floors cost more than the overlap hides --, Gardner sampler): the chain starts before the last span has
 * arrived and runs in three unequal segments with carried state (the streaming path over the resident capture; 55 / 28 /
 * 17 % of the capture, cut where a segment can use the whole-capture kernels, so that only the last, small one is left to run
 * when the last byte has arrived).  Frames, text and pdt_get_stats' counts then describe the whole capture as ever; but
 * pdt_read_stage / pdt_stage_len describe the LAST SEGMENT only (window-local indices), the pll / agc seam counters and
 * gpu_ms are the last segment's, and no stream is left open behind the call (pdt_stats.segments says so; a context
 * whose caller switched the PLL stream on, pdt_keep_pll(ctx, 1), is demodulated in one piece).
 * A capture that does not fit (ABI 4): when the buffers of a whole capture -- about 8 x the file for POES at 250 ksps --
 * exceed the device's free memory, the capture goes through a bounded window instead (pieces of the file pushed through the
 * streaming path with carried state: same frames, text, counts and reports; pdt_stats.windowed = 1).  The reference's loop
 * takes a file of any length (POESTIPdemod/main.c:373, while(!feof)); so do pdt_demod_fd / _file / _pcm16 / _f32.            */
int  pdt_demod_fd(pdt_ctx *ctx, int fd, uint64_t byte_offset, uint64_t nframes, int sample_format);
/* The whole job of POESTIPdemod/main.c:373-492 / ARGOSdemod/main.c:250-306 in one call: capture file in (as pdt_demod_fd), the
 * minor-frame / packet text out to the descriptor text_fd, from its position on -- the reference's ByteSync.c:62-101 writes
 * that text with fprintf WHILE it demodulates, and so does this: where the capture is demodulated in segments (above) the
 * text of a finished segment is formatted and written while the next segment runs, so that what is left behind the last
 * kernel is the last segment's text alone.  Otherwise it is pdt_demod_fd followed by pdt_write_frames.  *text_bytes (may be
 * NULL) = bytes written.  pdt_frames / pdt_get_stats / pdt_format_frames afterwards as after pdt_demod_fd.  (ABI 3)     */
int  pdt_demod_file(pdt_ctx *ctx, int fd, uint64_t byte_offset, uint64_t nframes, int sample_format, int text_fd,
                    uint64_t *text_bytes);
/* ... or already resident in device memory (no copy; buffer is only read).                      */
int  pdt_demod_device(pdt_ctx *ctx, const void *iq_device, uint64_t nframes);

/* Batched many-capture mode: `count` independent captures, one context each, inputs resident in
 * device memory.  The kernels of all captures are enqueued back to back on the contexts' own streams -- they
 * overlap on the GPU, whose serial recurrences leave most of it idle for a single capture -- and the results are
 * collected afterwards; each context then holds exactly what pdt_demod_device would have produced.
 * (Measured: 1.5x the single-capture throughput at 8 captures of 30 M samples.)                             */
int  pdt_demod_batch_device(pdt_ctx *const *ctxs, const void *const *iq_device, const uint64_t *nframes, int count);

/* RAW input of demodPOES (".raw": interleaved IEEE float32 I,Q used as they are, no normalisation;
 * GetComplexRawChunk, wave.c:413-540, POESTIPdemod/main.c:313-339; the sample rate comes from -s).
 * POES only -- ARGOSdemod/main.c:238-241 refuses RAW files.                                       */
int  pdt_demod_f32(pdt_ctx *ctx, const float *iq_host, uint64_t nframes);
int  pdt_demod_device_f32(pdt_ctx *ctx, const void *iq_device, uint64_t nframes);

/* Streaming front end: feed a capture piece by piece, as the reference's real-time twin does with
 * 2 400-frame sound-card blocks (POESTIPdemodPortAudio/main.c:324-393), and collect minor frames as they become final.
 * Like the reference's chunk loop, every stage carries its state from one piece of work to the next: whenever a push
 * completes one or more reference chunks, exactly those new chunks are demodulated -- PLL phase / frequency / lock state,
 * the FIR's last inputs, AGC gain, sampler state (nextSample, halfSample, prev), Manchester history and clock phase, the
 * byte synchroniser's last bits and open frame all continue from where the previous segment ended -- and the frames
 * completed by these chunks are reported.  Cost per push depends on the push, not on the length of the stream; the device
 * keeps a bounded window of the input (the PLL's warm-up history, see pdt_stream_retained) and nothing else of the past.
 * Guarantee: the frames reported by the pushes followed by those of pdt_stream_end are exactly the frames of one
 * pdt_demod_* call on the whole capture.
 *   pdt_stream_begin      forget any stream in progress (the sample format is fixed by the first push); optional: the first
 *                         push after pdt_open / pdt_stream_end opens a new stream by itself.  While a stream is open (first
 *                         push .. pdt_stream_end / pdt_stream_begin) the stage buffers hold the tails the next push continues
 *                         from: every pdt_demod_* and pdt_stage_* entry returns PDT_ERR_STATE
 *   pdt_stream_push_*     append nframes I,Q pairs; *new_frames = frames that became final with this push
 *   pdt_stream_end        the capture is over: demodulate the short last chunk, report the remaining frames (a frame cut
 *                         by the end stays partial, Q11); afterwards pdt_frames / pdt_get_stats / pdt_format_frames
 *                         describe the whole stream
 *   pdt_stream_frames     the frames reported by the last push / end, in order
 *   pdt_stream_retained   input samples the device currently holds (history + not yet demodulated)                  */
int      pdt_stream_begin(pdt_ctx *ctx);
int      pdt_stream_push_pcm16(pdt_ctx *ctx, const int16_t *iq_host, uint64_t nframes, uint64_t *new_frames);
int      pdt_stream_push_f32(pdt_ctx *ctx, const float *iq_host, uint64_t nframes, uint64_t *new_frames);
int      pdt_stream_end(pdt_ctx *ctx, uint64_t *new_frames);
uint64_t pdt_stream_frames(const pdt_ctx *ctx, pdt_frame *out, uint64_t max_frames);
uint64_t pdt_stream_retained(const pdt_ctx *ctx);

/* Results of the last pdt_demod_* call. */
uint64_t pdt_num_frames(const pdt_ctx *ctx);
uint64_t pdt_frames(const pdt_ctx *ctx, pdt_frame *out, uint64_t max_frames);
int      pdt_get_stats(const pdt_ctx *ctx, pdt_stats *out);
/* Text exactly as the reference writes it to minorFrames_*.txt / packets_*.txt.
 * Returns the number of bytes needed; writes at most `cap` bytes.                               */
uint64_t pdt_format_frames(const pdt_ctx *ctx, char *buf, uint64_t cap);
/* The same text for any array of frame records (e.g. the records of several captures gathered on one rank); host only, no
 * context and no GPU needed.  POESTIPdemod/ByteSync.c:62-69,96-101, ARGOSdemod/ByteSync.c:62-70,99-103.                 */
uint64_t pdt_format_records(const pdt_frame *frames, uint64_t nframes, char *buf, uint64_t cap);
/* The same text written to an open file descriptor, from the descriptor's position on (the reference's fprintf calls,
 * ByteSync.c:62-101, all at once): slices of the frames are formatted and written side by side (pwrite) when the
 * descriptor can seek, in order otherwise; the position ends behind the text.  *bytes_written may be NULL.
 * PDT_ERR_IO when a write fails.  (ABI 2)                                                                         */
int      pdt_write_frames(const pdt_ctx *ctx, int fd, uint64_t *bytes_written);
int      pdt_write_records(const pdt_frame *frames, uint64_t nframes, int fd, uint64_t *bytes_written);

/* Copy an intermediate stream back to the host (elements [first, first+count)); returns the
 * number of elements copied, or a negative error.  Element type per the PDT_ST_* table;
 * DT = float (POES) / double (ARGOS).                                                           */
int64_t  pdt_read_stage(const pdt_ctx *ctx, int stage, uint64_t first, uint64_t count, void *out);
uint64_t pdt_stage_len(const pdt_ctx *ctx, int stage);

/* Stage-level entry: run only the sync-word search + frame extraction kernels
 * (ByteSyncOnSyncword, POESTIPdemod/ByteSync.c:16-150 / FindSyncWords, ARGOSdemod/ByteSync.c:17-150)
 * on a string of '0'/'1' characters; the time stamp of bit k is k.  Results through pdt_frames /
 * pdt_format_frames.  (The reference ships exactly such a bit-string harness, commented out, at
 * POESTIPdemod/ByteSync.c:6-14.)                                                                  */
int      pdt_stage_bytesync(pdt_ctx *ctx, const uint8_t *bits_host, uint64_t nbits);
/* The same when the string continues an earlier one (the synchronisers keep their last bits and an open frame in statics
 * between calls, ByteSync.c:18-22): no sync word may COMPLETE before bit first_sync_end -- one that ends inside the bits the
 * caller kept from the previous call was seen there, with the real bits in front of it instead of the zeros the ring starts
 * with.  pdt_compat.c continues the reference's call-by-call state with this.                                            */
int      pdt_stage_bytesync_from(pdt_ctx *ctx, const uint8_t *bits_host, uint64_t nbits, uint64_t first_sync_end);

/* More stage-level entries: one stage of the chain on caller data in host memory (DT = float for POES
 * contexts, double for ARGOS), through the kernels the whole-capture path uses, with the reference function's hidden
 * `static` variables as an explicit state record the caller keeps -- so that per-chunk dumps of the reference can be
 * replayed stage by stage, chunk after chunk.  state == NULL: a fresh stage, nothing carried out.  A zeroed record is the
 * reference's state before its first call.                                                                           */
typedef struct pdt_manchester_state {   /* ManchesterDecode.c:16-20 */
    double   current, previous;         /* currentSample, prevSample after the last call (exact for float and double)  */
    uint32_t clockmod;                  /* which symbol parity ends a bit                                              */
    uint32_t even_odd;                  /* evenOddCounter (an unsigned char there: counted modulo 256)                 */
} pdt_manchester_state;
/* unsigned long ManchesterDecode(DT *dataStreamIn, DT *dataStreamTime, unsigned long nSymbols, unsigned char *bitStream,
 * DT resyncThreshold) (ManchesterDecode.h:3): bits_out receives the '0'/'1' characters (room for nsymbols: after a
 * resynchronisation consecutive symbols can both end a bit),
 * *nbits_out their number (the return value); bit_symbol_out[j] (optional) = index idxi of the symbol whose time stamp the
 * reference's in-place compaction gives bit j (:86).                                                                   */
int      pdt_stage_manchester(pdt_ctx *ctx, const void *symbols_host, uint64_t nsymbols, double resync_threshold,
                              pdt_manchester_state *state, uint8_t *bits_out, uint32_t *bit_symbol_out, uint64_t *nbits_out);
typedef struct pdt_fir_state {          /* LowPassFilter.c:13-41 (interpolating form) / :76-100 (in place)               */
    uint64_t count;                     /* inputs filtered so far: the interpolating form's ring position is count mod K */
    double   history[64];               /* the last K inputs, oldest first; K = ntaps / interp (POES, 26) or ntaps (ARGOS, 50) */
} pdt_fir_state;
/* void LowPassFilterInterp(DT *inTime, DT *in, DT *out, DT *outTime, unsigned long n, DT *h, int N, int interp) with the
 * context's taps and interpolation factor (LowPassFilter.h:4; POES), void LowPassFilter(DT *data, unsigned long n, DT *h,
 * int N) (:6; ARGOS): n inputs -> n * interp outputs in out_host.  The time streams are closed-form here
 * (pdt_time_axis).                                                                                                    */
int      pdt_stage_fir(pdt_ctx *ctx, const void *in_host, uint64_t n, pdt_fir_state *state, void *out_host);

typedef struct pdt_pll_state {          /* CarrierTrackingPLL.c:60-75 */
    int32_t  started;                   /* 0 = the next call is the first one (firstLock == -2: the statics get their start values) */
    int32_t  locked;                    /* firstLock >= 0: the one-time lock has happened, the tracking gains are in force   */
    int64_t  lock_index;                /* firstLock: index, within the call that saw it, of the sample that declared the lock */
    double   lock_freq_hz;              /* " : PLL locked at %0.2fHz" (:269)                                                  */
    double   phase, freq;               /* d_phase, d_freq after the last sample                                              */
    double   avg_phase;                 /* averagePhase (the return value)                                                    */
    double   locksig;                   /* d_locksig                                                                          */
    double   sweep;                     /* sweep (signed, :232-246); frozen at the lock                                       */
} pdt_pll_state;
/* DT CarrierTrackPLL(DT complex *complexDataIn, DT *realDataOut, DT *lockSignalStreamOut, unsigned int nSamples, DT Fs, ...)
 * (CarrierTrackPLL.h:11) with the constants the context's main passes (POESTIPdemod/main.c:413-420, ARGOSdemod/main.c:265; the
 * twin's with PDT_CHAIN_LIVE).  iq_host = n samples as the capture file holds them: PDT_FMT_PCM16 (int16 pairs, converted to
 * `DT complex` as wave.c:127-172 does) or PDT_FMT_F32 (pairs of float = `float complex` as they are; float contexts only,
 * PDT_ERR_FORMAT otherwise); out_host = realDataOut, lock_out_host (optional) = lockSignalStreamOut, *avg_phase_ret (optional)
 * = the return value.  Runs the whole-capture path's PLL kernels (sequential acquisition, block-parallel tracking with
 * validated seams, lock-detector and averagePhase EMAs) from the record's state and stops behind the PLL.  Not with
 * cfg.profile (PDT_ERR_STATE; like every stage entry, not while a stream is open).                                                                          */
int      pdt_stage_pll(pdt_ctx *ctx, const void *iq_host, uint64_t n, int sample_format, pdt_pll_state *state, void *out_host,
                       void *lock_out_host, double *avg_phase_ret);
/* DT StaticGain(DT complex *complexData, unsigned int nSamples, DT desiredLevel) (AGC.h:4, AGC.c:48-74): the gain the mains
 * take from their first chunk (main.c:384-389, level 1.0).  The samples as the capture file holds them (PDT_FMT_PCM16: int16
 * pairs, converted as wave.c:127-172 does; PDT_FMT_F32: float pairs as they are); stateless.                              */
int      pdt_stage_static_gain(pdt_ctx *ctx, const void *iq_host, uint64_t n, int sample_format, double level, double *gain_out);
typedef struct pdt_gardner_state {      /* GardenerClockRecovery.c:12-15 */
    double   next_sample;               /* nextSample, rolled over by the chunk length at the end of the call (:113)          */
    double   prev_bit;                  /* prevBit                                                                            */
    double   half_sample;               /* halfSample: the mid-point INDEX of the next symbol, not rolled over (Q3)           */
} pdt_gardner_state;
/* unsigned long GardenerClockRecovery(DT *dataStreamIn, DT *dataStreamInTime, unsigned long numSamples, DT *dataStreamOut,
 * int Fs, DT baud, DT stepRange, DT kp) (GardenerClockRecovery.h:3) with the context's rate / baud / limits (main.c:438,
 * ARGOSdemod/main.c:278).  in_host is the caller's BUFFER: `capacity` elements of which the first n are this call's samples --
 * the function reads a few elements past n (the stale mid-point index of the first symbol, the look-ahead of the last), i.e.
 * whatever the buffer still holds there from earlier calls, zeros behind `capacity` (the mains over-allocate; fresh pages).
 * ARGOS: neighbour_host (optional) = the `capacity` elements of the array the main allocated right behind the buffer
 * (lockSignalStream, ARGOSdemod/main.c:169-176) -- with glibc's heap layout the reads past the buffer land there (Q16).
 * out_host: the symbols (room for n / (step - 0.25) + 2), pick_out[k] (optional) = index of the sample symbol k was taken
 * at (dataStreamInTime[k] = dataStreamInTime[pick_out[k]] is the in-place compaction), *nsym_out = the return value.
 * Runs the sequential sampler kernel of the streaming path (one wavefront).                                              */
int      pdt_stage_gardner(pdt_ctx *ctx, const void *in_host, uint64_t n, uint64_t capacity, const void *neighbour_host,
                           pdt_gardner_state *state, void *out_host, uint64_t *pick_out, uint64_t *nsym_out);
typedef struct pdt_mm_state {           /* MMClockRecovery.c:12-22 */
    int32_t  started;                   /* 0 = the next call is the first one (stepSize = Fs / baud, :20)                     */
    int32_t  pad;
    double   next_sample, step_size, sample_last;   /* nextSample (rolled over, :80), stepSize, sampleLast                    */
} pdt_mm_state;
/* unsigned long MMClockRecovery(DT *dataStreamIn, DT *dataStreamInTime, unsigned long numSamples, DT *dataStreamOut, int Fs,
 * DT baud, DT stepRange, DT kp) (MMClockRecovery.h:3) with the context's rate / baud and cfg.mm_step_range / mm_kp (0 = 3 and
 * 0.15, the commented-out call of ARGOSdemod/main.c:277).  Outputs as pdt_stage_gardner's; this sampler never reads past n.  */
int      pdt_stage_mm(pdt_ctx *ctx, const void *in_host, uint64_t n, pdt_mm_state *state, void *out_host, uint64_t *pick_out,
                      uint64_t *nsym_out);
typedef struct pdt_agc_state {          /* AGC.c:84-95 */
    int32_t  started;                   /* 0 = the next call is the first one: its `initial` becomes the gain           */
    int32_t  pad;
    double   gain;                      /* the static gain after the last call                                          */
} pdt_agc_state;
/* void NormalizingAGC(DT *dataStreamIn, unsigned long nSamples, DT initial, DT attack_rate, DT decay_rate) (AGC.h:6): in place
 * on data_host.  attack / decay 0 = the values the mains pass for this context's rate (POESTIPdemod/main.c:429-430,
 * ARGOSdemod/main.c:270).  Evaluated block-parallel with validated seams like the whole-capture path (k_agc_affine / _guess /
 * _block / _scan / _fix).                                                                                               */
int      pdt_stage_agc(pdt_ctx *ctx, void *data_host, uint64_t n, double initial, double attack, double decay,
                       pdt_agc_state *state);
/* void Squelch(DT *dataStream, DT *squelchStreamIn, unsigned long nSamples, DT squelchThreshold) (AGC.h:8): in place; its only
 * static (lastSquelch) feeds commented-out console output.                                                               */
int      pdt_stage_squelch(pdt_ctx *ctx, void *data_host, const void *lock_host, uint64_t n, double threshold);

/* Frame validation of the POES minor frames of the last pdt_demod_* call (a per-frame kernel and a
 * reduction on the GPU; the reference does this offline in MATLAB from the text file).  MATLAB is
 * 1-indexed: its minorFrames(frame, w) is bytes[w-1] here.                                          */
typedef struct pdt_tip_frame {
    uint16_t minor_id;     /* 9-bit minor-frame counter ((bytes[4] & 1) << 8) | bytes[5]   (daytimeDecode.m:4)      */
    uint8_t  spacecraft;   /* bytes[2]: 8 = NOAA-15, 13 = NOAA-18, 15 = NOAA-19            (daytimeDecode.m:16,82-93) */
    uint8_t  parity;       /* bit g set = even-parity check g failed; g = 0..4 covers bytes 2-18, 19-35, 36-52,
                              53-69, 70-86 against bits 5,4,3,2,1 of bytes[103]             (checkParity.m:20-86)  */
    uint8_t  checked;      /* 1 = complete frame (only whole frames are rows of the reference's matrix)            */
    uint8_t  has_time;     /* 1 = minor_id == 0: day / day_ms are valid                     (daytimeDecode.m:18)    */
    uint16_t day;          /* (bytes[8] << 1) + ((bytes[9] | 128) >> 7), as the reference computes it (:19)         */
    int32_t  day_ms;       /* ((bytes[9] & 7) << 24) + (bytes[10] << 16) + (bytes[11] << 8) + bytes[12];
                              -1 when not below 86 400 000                                  (:22-29)               */
} pdt_tip_frame;

typedef struct pdt_tip_summary {
    uint64_t frames_checked, good_frames;   /* "<good> out of <checked> Error Free Frames"  (checkParity.m:91)      */
    uint64_t good_chunks, bad_chunks;       /* zeros / ones of the parity matrix            (checkParity.m:92)      */
    int32_t  spacecraft;                    /* mode of the spacecraft ids, -1 = no frame    (daytimeDecode.m:82)    */
    int32_t  day;                           /* mode of the day numbers of the major-frame starts, -1 = none (:95)   */
    int64_t  t0_ms;                         /* mode(round(day_ms - 1000 * frame time)) over the positive ones,
                                               -1 = none (capture start in spacecraft ms of day, :34-36)            */
    uint64_t time_frames;                   /* frames with minor_id == 0                                             */
} pdt_tip_summary;

/* Runs the validation kernels on the frames of the last pdt_demod_* call (POES contexts only). */
int      pdt_tip_check(pdt_ctx *ctx, pdt_tip_summary *out);
/* Per-frame records of the last pdt_tip_check, in frame order; returns the number copied. */
uint64_t pdt_tip_frames(const pdt_ctx *ctx, pdt_tip_frame *out, uint64_t max_frames);

/* Per-kernel device times of the last call (cfg.profile = 1). Returns the entry count. */
int      pdt_kernel_times(const pdt_ctx *ctx, pdt_kernel_time *out, int max_entries);

/* Host-side helpers */
int  pdt_make_lpf(int mode, uint32_t sample_rate, void *taps_out, int *ntaps, int *interp);
int  pdt_wav_parse_header(const uint8_t hdr[44], uint32_t *sample_rate, uint32_t *channels,
                          uint32_t *bits_per_sample, uint32_t *format, uint32_t *data_bytes);
/* Test hook, host only: the library's own restatements of the C-library functions the reference calls (sincos / sin / cos /
 * sincosf / hypot / hypotf; CarrierTrackingPLL.c:106-107,134-135, LowPassFilter.c:148,163, AGC.c:57-67) evaluated on the host
 * over an array.  fn: 0 sincos -> out0 = sin, out1 = cos; 1 sin; 2 cos; 3 sincosf of (float)x, widened; 4 hypot of the pairs
 * (x[2i], x[2i+1]) -> out0[i]; 5 hypotf of the pairs; 6 the branch-free sincosf of the fused mix + FIR kernel; 7 / 8 the error
 * and phase wraps of one float PLL step (CarrierTrackingPLL.c:168-188) in their fused form.  The kernels run the same code.     */
int  pdt_host_math(int fn, const double *x, uint64_t n, double *out0, double *out1);
/* The reference's running-sum time axis (wave.c:91,96-97,167-168): value after m additions of Ts. */
double pdt_time_axis(int mode, uint32_t sample_rate, uint64_t m);

#ifdef __cplusplus
}
#endif
#endif
