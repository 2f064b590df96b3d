/*
 * pdt_compat.c -- the reference's stage functions, with their own prototypes, over libpdt.so.
 *
 * The second form of the drop-in boundary: a program written against the headers of common/ -- POESTIPdemod/main.c,
 * ARGOSdemod/main.c, or a harness such as oracle/ref_driver.c -- links this library instead of the reference's objects and
 * keeps its chunk loop, its buffers and its time arrays; every call runs the stage's HIP kernels through the pdt_stage_* entries
 * of include/pdt.h.  Like the reference's functions these keep their state in statics: one stream per process.
 *
 *   libpdt_compat_poes.so   DECIMAL_TYPE float   (POESTIPdemod/config.h: USE_FLOATS 1)
 *   libpdt_compat_argos.so  DECIMAL_TYPE double  (-DPDT_COMPAT_ARGOS; ARGOSdemod/config.h: USE_FLOATS 0)
 *
 * Prototypes replaced, one for one:
 *   CarrierTrackPLL.h:11       CarrierTrackPLL
 *   LowPassFilter.h:4-6        LowPassFilter, LowPassFilterInterp, MakeLPFIR
 *   AGC.h:5-7                  Squelch, StaticGain, NormalizingAGC
 *   GardenerClockRecovery.h:3  GardenerClockRecovery
 *   MMClockRecovery.h:3        MMClockRecovery
 *   ManchesterDecode.h:3       ManchesterDecode
 *   POESTIPdemod/ByteSync.h:4  ByteSyncOnSyncword        ARGOSdemod/ByteSync.h:3  FindSyncWords
 *   wave.h:27-29               ReadWavHeader, GetComplexRawChunk, GetComplexWaveChunk (round 5; host only: file reads and the
 *                              running-sum time axis, SURVEY Q1 -- a program linked against this library needs no object of
 *                              the reference any more)
 * What the kernels do not carry -- the time arrays -- is index bookkeeping and is done here exactly as the reference does it
 * (LowPassFilter.c:67, GardenerClockRecovery.c:30,111, ManchesterDecode.c:86, ByteSync.c:96).  The loop constants a call passes
 * (frequency range, lock threshold and rate, loop bandwidths; AGC rates; baud rate, timing gain and clip; resync threshold) are
 * handed on to the context (pdt_set_loop_params, round 5): any values, call by call, as with the reference.  What must still be
 * the mains': the filter (MakeLPFIR's design for the chain: the FIR kernels are built around its 26 taps per phase), a timing
 * clip of at most 0.1, M&M's limits, the sync words -- anything else there ends the program with a message.
 *
 * No CPU path: without a GPU the first call ends the program ("pdt_compat: ... no HIP device").
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pdt.h"

#ifdef PDT_COMPAT_ARGOS
#define DT double
#define COMPAT_MODE PDT_MODE_ARGOS
#else
#define DT float
#define COMPAT_MODE PDT_MODE_POES
#endif

static pdt_ctx *g_ctx;
static uint32_t g_rate;
static int g_used;                 /* a stage has run on g_ctx: its rate is final */

static void die(const char *what, int rc)
{
    fprintf(stderr, "pdt_compat: %s: %s\n", what, rc ? pdt_strerror(rc) : "not what the reference's mains pass");
    exit(3);
}
#define TRY(call) do { int rc_ = (call); if (rc_ != PDT_OK) die(#call, rc_); } while (0)

/* the context of this process; the sample rate is known from the first call that carries it (MakeLPFIR, CarrierTrackPLL) */
static pdt_ctx *ctx_for(uint32_t rate)
{
    if (g_ctx && (rate == 0 || rate == g_rate)) return g_ctx;
    if (g_ctx && g_used) die("sample rate changed in mid-stream", 0);
    if (g_ctx) pdt_close(g_ctx);
    pdt_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.mode = COMPAT_MODE;
    cfg.sample_rate = rate ? rate : (COMPAT_MODE == PDT_MODE_ARGOS ? 32000u : 50000u);     /* (StaticGain does not depend on it) */
    g_rate = cfg.sample_rate;
    TRY(pdt_open(&cfg, &g_ctx));
    return g_ctx;
}

/* the loop constants the stage functions were last called with, as the context holds them (0 = the mains') */
static pdt_loop_params g_lp;
static void set_params(pdt_ctx *c, const pdt_loop_params *want)
{
    if (memcmp(&g_lp, want, sizeof g_lp) == 0) return;
    TRY(pdt_set_loop_params(c, want));
    g_lp = *want;
}

/* `DT complex` samples as something the stage entries take: float pairs as they are; doubles are the WAV's int16 / 32768
 * (wave.c:127-172), converted back exactly */
static const void *iq_arg(DT complex *x, unsigned long n, int *fmt, void **scratch)
{
#ifdef PDT_COMPAT_ARGOS
    int16_t *p = malloc(4 * (n ? n : 1));
    if (!p) die("malloc", PDT_ERR_NOMEM);
    const double *d = (const double *)x;
    for (unsigned long i = 0; i < 2 * n; i++) {
        const double v = d[i] * 32768.0;
        if (v != rint(v) || v < -32768.0 || v > 32767.0) die("double samples that are not int16 / 32768", 0);
        p[i] = (int16_t)v;
    }
    *fmt = PDT_FMT_PCM16;
    *scratch = p;
    return p;
#else
    (void)n;
    *fmt = PDT_FMT_F32;
    *scratch = NULL;
    return x;
#endif
}

/* ---- AGC.h */
DT StaticGain(DT complex *complexData, unsigned int nSamples, DT desiredLevel)
{
    int fmt;
    void *tmp;
    const void *iq = iq_arg(complexData, nSamples, &fmt, &tmp);
    double gain = 0;
    TRY(pdt_stage_static_gain(ctx_for(0), iq, nSamples, fmt, (double)desiredLevel, &gain));
    free(tmp);
    return (DT)gain;
}

static pdt_agc_state g_agc;
void NormalizingAGC(DT *dataStreamIn, unsigned long nSamples, DT initial, DT attack_rate, DT decay_rate)
{
    g_used = 1;
    if (attack_rate == 0 || decay_rate == 0) die("NormalizingAGC rate of 0 (the stage entry reads 0 as \"the mains' constant\")", 0);
    TRY(pdt_stage_agc(ctx_for(0), dataStreamIn, nSamples, (double)initial, (double)attack_rate, (double)decay_rate, &g_agc));
}

void Squelch(DT *dataStream, DT *squelchStreamIn, unsigned long nSamples, DT squelchThreshold)
{
    g_used = 1;
    TRY(pdt_stage_squelch(ctx_for(0), dataStream, squelchStreamIn, nSamples, (double)squelchThreshold));
}

/* ---- CarrierTrackPLL.h */
static pdt_pll_state g_pll;
DT CarrierTrackPLL(DT complex *complexDataIn, DT *realDataOut, DT *lockSignalStreamOut, unsigned int nSamples, DT Fs, DT freqRange,
                   DT d_lock_threshold, DT lockSigAlpha, DT loopbw_acq, DT loopbw_track)
{
    pdt_ctx *c = ctx_for((uint32_t)Fs);
    g_used = 1;
    {   /* its constants, whatever they are (CarrierTrackPLL.h:11) */
        pdt_loop_params lp = g_lp;
        lp.pll_freq_range_hz = (double)freqRange;
        lp.pll_lock_threshold = (double)d_lock_threshold;
        lp.pll_lock_alpha = (double)lockSigAlpha;
        lp.pll_loopbw_acq = (double)loopbw_acq;
        lp.pll_loopbw_track = (double)loopbw_track;
        /* a field left 0 means "the mains' constant" to the context: a caller's own 0 must not turn into that.  A lock threshold
         * of 0 is a value (zero_mask); a range, detector rate or bandwidth of 0 is no loop at all */
        if (freqRange == 0 || lockSigAlpha == 0 || loopbw_acq == 0 || loopbw_track == 0) die("CarrierTrackPLL constant of 0", 0);
        lp.zero_mask = (lp.zero_mask & ~(uint32_t)PDT_LP_ZERO_LOCK_THRESHOLD) | (d_lock_threshold == 0 ? PDT_LP_ZERO_LOCK_THRESHOLD : 0u);
        set_params(c, &lp);
    }
    int fmt;
    void *tmp;
    const void *iq = iq_arg(complexDataIn, nSamples, &fmt, &tmp);
    const int was_locked = g_pll.locked;
    double avg = 0;
    TRY(pdt_stage_pll(c, iq, nSamples, fmt, &g_pll, realDataOut, lockSignalStreamOut, &avg));
    free(tmp);
    if (!was_locked && g_pll.locked) printf(" : PLL locked at %0.2fHz\n", g_pll.lock_freq_hz);     /* CarrierTrackingPLL.c:269 */
    return (DT)avg;
}

/* ---- LowPassFilter.h */
int MakeLPFIR(DT *h, int N, DT Fc, DT Fs, int interpFactor)
{
    if (interpFactor < 1) die("MakeLPFIR interpolation factor", 0);
    const uint32_t rate = (uint32_t)lrint((double)Fs / interpFactor);
    int ntaps = 0, interp = 0;
    ctx_for(rate);
    TRY(pdt_make_lpf(COMPAT_MODE, rate, NULL, &ntaps, &interp));
    if (ntaps != N || interp != interpFactor || Fc != (DT)(COMPAT_MODE == PDT_MODE_ARGOS ? 700.0 : 11000.0)) die("MakeLPFIR design", 0);
    TRY(pdt_make_lpf(COMPAT_MODE, rate, h, &ntaps, &interp));
    return N;                                                      /* LowPassFilter.c:174 */
}

static pdt_fir_state g_fir;
static void check_taps(const DT *h, int N, int interpFactor)
{
    static int checked;
    if (checked) return;
    DT mine[256];
    int ntaps = 0, interp = 0;
    TRY(pdt_make_lpf(COMPAT_MODE, g_rate, NULL, &ntaps, &interp));
    if (ntaps != N || interp != interpFactor || N > 256) die("filter length / interpolation factor", 0);
    TRY(pdt_make_lpf(COMPAT_MODE, g_rate, mine, &ntaps, &interp));
    if (memcmp(mine, h, sizeof(DT) * (size_t)N) != 0) die("filter taps other than MakeLPFIR's", 0);
    checked = 1;
}

void LowPassFilterInterp(DT *dataStreamInTime, DT *dataStreamIn, DT *dataStreamOut, DT *dataStreamOutTime, unsigned long nSamples,
                         DT *filterCoeffs, int N, int interpFactor)
{
    pdt_ctx *c = ctx_for(0);
    g_used = 1;
    check_taps(filterCoeffs, N, interpFactor);
    TRY(pdt_stage_fir(c, dataStreamIn, nSamples, &g_fir, dataStreamOut));
    /* LowPassFilter.c:67: an output carries the time of the input BEHIND the one it was computed from -- for the chunk's last
     * input that is the array's element nSamples, whatever the caller's buffer holds there */
    for (unsigned long o = 0; o < nSamples * (unsigned long)interpFactor; o++) dataStreamOutTime[o] = dataStreamInTime[o / interpFactor + 1];
}

void LowPassFilter(DT *dataStream, unsigned long nSamples, DT *filterCoeffs, int N)
{
    pdt_ctx *c = ctx_for(0);
    g_used = 1;
    check_taps(filterCoeffs, N, 1);
    DT *out = malloc(sizeof(DT) * (nSamples ? nSamples : 1));
    if (!out) die("malloc", PDT_ERR_NOMEM);
    TRY(pdt_stage_fir(c, dataStream, nSamples, &g_fir, out));
    memcpy(dataStream, out, sizeof(DT) * nSamples);
    free(out);
}

/* ---- GardenerClockRecovery.h / MMClockRecovery.h */
static void check_sampler(int Fs, DT baud)
{
    int interp = 1;
    TRY(pdt_make_lpf(COMPAT_MODE, g_rate, NULL, NULL, &interp));
    if ((uint32_t)Fs != g_rate * (uint32_t)interp) die("sampler rate other than the filter's output rate", 0);
    (void)baud;
}

static pdt_gardner_state g_gardner;
unsigned long GardenerClockRecovery(DT *dataStreamIn, DT *dataStreamInTime, unsigned long numSamples, DT *dataStreamOut, int Fs, DT baud,
                                    DT stepRange, DT kp)
{
    pdt_ctx *c = ctx_for(0);
    g_used = 1;
    check_sampler(Fs, baud);
    {   /* baud, clip and gain of this call (GardenerClockRecovery.h:3; the mains: main.c:438, ARGOSdemod/main.c:278) */
        pdt_loop_params lp = g_lp;
        lp.gardner_baud = (double)baud;
        lp.gardner_step_range = (double)stepRange;
        lp.gardner_kp = (double)kp;
        if (stepRange > (DT)0.1) die("GardenerClockRecovery stepRange above 0.1", 0);
        lp.zero_mask &= ~(uint32_t)(PDT_LP_ZERO_GARDNER_KP | PDT_LP_ZERO_GARDNER_STEP_RANGE);       /* (an open-loop sampler: 0 is 0) */
        if (kp == 0) lp.zero_mask |= PDT_LP_ZERO_GARDNER_KP;
        if (stepRange == 0) lp.zero_mask |= PDT_LP_ZERO_GARDNER_STEP_RANGE;
        set_params(c, &lp);
    }
    /* The function reads a little past numSamples (the first symbol's stale mid-point index, Q3; the last symbol's look-ahead):
     * whatever lies behind the caller's samples, in its buffer or behind it (Q16).  So does this one: the kernel is handed the
     * caller's memory up to the furthest index the sampler can form. */
    const double step = (double)Fs / (double)baud;
    /* ... which is index numSamples + step + 1 at most (rint(nextSample) of the symbol that ends the loop, :59,:111; the stale
     * mid-point index lies below it): this many elements are taken from the caller, as the reference takes them.  The kernel
     * stages a little more than the sampler can ask for (its window margin): that rest is zeros, not the caller's heap. */
    const unsigned long reach = numSamples + 2 * (unsigned long)step + 24;
    const unsigned long touch = numSamples + (unsigned long)step + 2;
    const unsigned long cap_sym = (unsigned long)((double)numSamples / (step - 0.25)) + 4;
    uint64_t *pick = malloc(sizeof(uint64_t) * cap_sym);
    DT *sym = malloc(sizeof(DT) * cap_sym);
    DT *win = calloc(reach, sizeof(DT));
    if (!pick || !sym || !win) die("malloc", PDT_ERR_NOMEM);
    memcpy(win, dataStreamIn, sizeof(DT) * (touch < reach ? touch : reach));
    uint64_t nsym = 0;
    TRY(pdt_stage_gardner(c, win, numSamples, reach, NULL, &g_gardner, sym, pick, &nsym));
    free(win);
    for (uint64_t k = 0; k < nsym; k++) {
        dataStreamOut[k] = sym[k];
        dataStreamInTime[k] = dataStreamInTime[pick[k]];                                     /* :30 (in place: pick[k] >= k) */
    }
    /* :111 -- nextSample has been rolled over by numSamples (:113), an exact subtraction: adding it back is exact too */
    dataStreamInTime[nsym] = dataStreamInTime[(unsigned int)rint((DT)g_gardner.next_sample + (DT)numSamples)];
    free(pick);
    free(sym);
    return (unsigned long)nsym;
}

static pdt_mm_state g_mm;
unsigned long MMClockRecovery(DT *dataStreamIn, DT *dataStreamInTime, unsigned long numSamples, DT *dataStreamOut, int Fs, DT baud,
                              DT stepRange, DT kp)
{
    pdt_ctx *c = ctx_for(0);
    g_used = 1;
    check_sampler(Fs, baud);
    {
        pdt_loop_params lp = g_lp;
        lp.gardner_baud = (double)baud;
        set_params(c, &lp);
    }
    if (stepRange != (DT)3.0 || kp != (DT)0.15) die("MMClockRecovery limits (the context's are 3 and 0.15, ARGOSdemod/main.c:277)", 0);
    const double step_min = (double)Fs / ((double)baud + (double)stepRange);
    const unsigned long cap_sym = (unsigned long)((double)numSamples / step_min) + 4;
    uint64_t *pick = malloc(sizeof(uint64_t) * cap_sym);
    DT *sym = malloc(sizeof(DT) * cap_sym);
    if (!pick || !sym) die("malloc", PDT_ERR_NOMEM);
    uint64_t nsym = 0;
    TRY(pdt_stage_mm(c, dataStreamIn, numSamples, &g_mm, sym, pick, &nsym));
    for (uint64_t k = 0; k < nsym; k++) {
        dataStreamOut[k] = sym[k];
        dataStreamInTime[k] = dataStreamInTime[pick[k]];                                     /* MMClockRecovery.c:29 / :59 */
    }
    free(pick);
    free(sym);
    return (unsigned long)nsym;
}
int sign(DT x) { return (x > 0) - (x < 0); }                                                 /* MMClockRecovery.c:85-88 */

/* ---- ManchesterDecode.h */
static pdt_manchester_state g_manch;
unsigned long ManchesterDecode(DT *dataStreamIn, DT *dataStreamInTime, unsigned long nSymbols, unsigned char *bitStream,
                               DT resyncThreshold)
{
    pdt_ctx *c = ctx_for(0);
    g_used = 1;
    uint8_t *bits = malloc(nSymbols + 8);
    uint32_t *bsym = malloc(sizeof(uint32_t) * (nSymbols + 8));
    if (!bits || !bsym) die("malloc", PDT_ERR_NOMEM);
    uint64_t nbits = 0;
    TRY(pdt_stage_manchester(c, dataStreamIn, nSymbols, (double)resyncThreshold, &g_manch, bits, bsym, &nbits));
    for (uint64_t j = 0; j < nbits; j++) {
        bitStream[j] = bits[j];
        dataStreamInTime[j] = dataStreamInTime[bsym[j]];                                     /* :86 (in place: bsym[j] >= j) */
    }
    free(bits);
    free(bsym);
    return (unsigned long)nbits;
}

/* ---- ByteSync.h.  The synchronisers keep the last syncWordLength bits and the frame being shifted in between calls
 * (ByteSync.c:18-22).  Here: the bits from the sync word of the open frame on -- or the last syncWordLength - 1 -- are kept and
 * handed to the search again in front of the new ones; what the file receives is the text of the frames minus what earlier
 * calls already wrote. */
static unsigned char *g_kept;
static size_t g_kept_n;
static int g_open;                 /* a frame is being shifted in: g_kept starts at its sync word */
static size_t g_open_written;      /* characters of its line already in the file */
static double g_open_time;
static int g_open_inverted;

static int sync_common(unsigned char *bitStreamIn, DT *bitStreamInTime, unsigned long nSamples, const char *syncWord,
                       unsigned int syncWordLength, FILE *f, const char *want, unsigned int want_len)
{
    pdt_ctx *c = ctx_for(0);
    g_used = 1;
    if (syncWordLength != want_len || memcmp(syncWord, want, want_len) != 0) die("sync word", 0);
    const size_t n = g_kept_n + nSamples;
    unsigned char *all = malloc(n ? n : 1);
    if (!all) die("malloc", PDT_ERR_NOMEM);
    memcpy(all, g_kept, g_kept_n);
    memcpy(all + g_kept_n, bitStreamIn, nSamples);
    TRY(pdt_stage_bytesync_from(c, all, n, g_open ? (uint64_t)(syncWordLength - 1) : (uint64_t)g_kept_n));
    const uint64_t nf = pdt_num_frames(c);
    pdt_frame *fr = malloc(sizeof(pdt_frame) * (nf ? nf : 1));
    if (!fr) die("malloc", PDT_ERR_NOMEM);
    pdt_frames(c, fr, nf);
    int found = 0;
    int still_open = 0;
    size_t open_from = 0;
    for (uint64_t k = 0; k < nf; k++) {
        const int carried = g_open && k == 0;
        if (carried && fr[k].bit_index != (int64_t)(syncWordLength - 1)) die("the open frame was not found again", PDT_ERR_STATE);
        if (carried) {
            fr[k].time = g_open_time;
            fr[k].inverted = (uint8_t)g_open_inverted;
        } else {
            fr[k].time = (double)bitStreamInTime[(size_t)fr[k].bit_index - g_kept_n];        /* ByteSync.c:96 / :130 */
            found++;
        }
        char line[512];
        const uint64_t len = pdt_format_records(&fr[k], 1, line, sizeof line);
        const size_t skip = carried ? g_open_written : 0;
        if (len > skip) fwrite(line + skip, 1, (size_t)len - skip, f);
        if (!fr[k].complete) {                                   /* (only the last one can be) */
            still_open = 1;
            open_from = (size_t)fr[k].bit_index - (syncWordLength - 1);
            g_open_written = (size_t)len;
            g_open_time = fr[k].time;
            g_open_inverted = fr[k].inverted;
        }
    }
    if (g_open && nf == 0) die("the open frame was not found again", PDT_ERR_STATE);
    g_open = still_open;
    const size_t from = still_open ? open_from : (n > syncWordLength - 1 ? n - (syncWordLength - 1) : 0);
    unsigned char *kept = malloc(n - from ? n - from : 1);
    if (!kept) die("malloc", PDT_ERR_NOMEM);
    memcpy(kept, all + from, n - from);
    free(g_kept);
    g_kept = kept;
    g_kept_n = n - from;
    free(all);
    free(fr);
    return found;
}

#ifdef PDT_COMPAT_ARGOS
int FindSyncWords(unsigned char *bitStreamIn, DT *bitStreamInTime, unsigned long nSamples, char *syncWord, unsigned int syncWordLength,
                  FILE *minorFrameFile)
{
    return sync_common(bitStreamIn, bitStreamInTime, nSamples, syncWord, syncWordLength, minorFrameFile, "0001011110000", 13);
}
#else
int ByteSyncOnSyncword(unsigned char *bitStreamIn, DT *bitStreamInTime, unsigned long nSamples, char *syncWord,
                       unsigned int syncWordLength, FILE *minorFrameFile)
{
    return sync_common(bitStreamIn, bitStreamInTime, nSamples, syncWord, syncWordLength, minorFrameFile, "1110110111100010000", 19);
}
#endif


/* ---- wave.h.  The capture readers: host-only, no GPU involved (they fill the caller's chunk buffers, which the stage functions
 * above then take).  What they must reproduce besides the samples is the TIME AXIS: a running sum of Ts = 1 / sample_rate in
 * DECIMAL_TYPE arithmetic, carried in function statics from chunk to chunk -- and from file to file --, one addition per sample
 * (wave.c:91,96-97,167-168; SURVEY Q1: in float it runs fast or slow per binade and stalls at 2^24 Ts).
 *
 * HEADER is passed by value: its layout is part of the link contract (wave.h:8-24). */
typedef struct HEADER {
    unsigned char riff[4];
    unsigned int overall_size;
    unsigned char wave[4];
    unsigned char fmt_chunk_marker[4];
    unsigned int length_of_fmt;
    unsigned int format_type;
    unsigned int channels;
    unsigned int sample_rate;
    unsigned int byterate;
    unsigned int block_align;
    unsigned int bits_per_sample;
    unsigned char data_chunk_header[4];
    unsigned int data_size;
    unsigned char type;
} HEADER;

static unsigned int le_field(const unsigned char *b, size_t got, size_t at, int nbytes)
{
    unsigned int v = 0;
    for (int k = 0; k < nbytes; k++)
        if (at + (size_t)k < got) v |= (unsigned int)b[at + (size_t)k] << (8 * k);
    return v;
}

/* wave.c:303-378: the canonical 44-byte header, field by field, no validation (Q15: a `fmt ` chunk of 16 bytes followed at once
 * by `data`); the file position ends behind it; what a short file does not hold stays zero */
HEADER ReadWavHeader(FILE *waveFilePtr)
{
    HEADER h;
    unsigned char b[44];
    if (waveFilePtr == NULL) {
        printf("Error opening file\n");
        exit(1);
    }
    memset(&h, 0, sizeof h);
    memset(b, 0, sizeof b);
    const size_t got = fread(b, 1, sizeof b, waveFilePtr);
    memcpy(h.riff, b, 4);
    h.overall_size = le_field(b, got, 4, 4);
    memcpy(h.wave, b + 8, 4);
    memcpy(h.fmt_chunk_marker, b + 12, 4);
    h.length_of_fmt = le_field(b, got, 16, 4);
    h.format_type = le_field(b, got, 20, 2);
    h.channels = le_field(b, got, 22, 2);
    h.sample_rate = le_field(b, got, 24, 4);
    h.byterate = le_field(b, got, 28, 4);
    h.block_align = le_field(b, got, 32, 2);
    h.bits_per_sample = le_field(b, got, 34, 2);
    memcpy(h.data_chunk_header, b + 36, 4);
    h.data_size = le_field(b, got, 40, 4);
    return h;
}

static void reader_checks(FILE *f, const HEADER *h, const void *a, const void *b)
{
    if (f == NULL) {
        printf("Error opening file\n");
        exit(1);
    }
    if (a == NULL || b == NULL) {
        printf("Dude, allocate your fracking memory already. UGH. \n");
        exit(1);
    }
    if (h->channels != 2) {
        printf("Complex read requires 2 channels (I and Q)\n");
        exit(1);
    }
}

/* wave.c:59-175.  PCM samples, I then Q, scaled by the full scale of the sample width; the value passes through an int16_t on
 * the way (Q5: of a 32-bit sample only the low half survives, an 8-bit file yields its first byte for both channels), so only
 * 16-bit files mean anything.  Returns the number of sample pairs the file still held. */
unsigned long int GetComplexWaveChunk(FILE *waveFilePtr, HEADER header, DT complex *waveData, DT *waveDataTime, unsigned long int nSamples)
{
    static DT running_time = 0, Ts = -1;
    reader_checks(waveFilePtr, &header, waveData, waveDataTime);
    if (header.format_type != 1) {
        printf("Only PCM is currently supported :(\n");
        exit(1);
    }
    if (Ts < 0) Ts = 1.0 / (DT)header.sample_rate;
    const long pair = (long)(header.channels * header.bits_per_sample) / 8;       /* bytes per I,Q pair */
    const long per = pair / (long)header.channels;
    if (per * (long)header.channels != pair) {
        printf("Error: %ld x %ud <> %ld\n", per, header.channels, pair);
        return nSamples;
    }
    const DT full = header.bits_per_sample == 8 ? (DT)128 : header.bits_per_sample == 16 ? (DT)32768
                    : header.bits_per_sample == 32 ? (DT)2147483648LL : (DT)0;
    unsigned char *raw = malloc((size_t)pair * (nSamples ? nSamples : 1));
    if (!raw) die("malloc", PDT_ERR_NOMEM);
    const unsigned long got = fread(raw, (size_t)pair, nSamples, waveFilePtr);
    for (unsigned long i = 0; i < got; i++) {
        const unsigned char *p = raw + (size_t)pair * i;
        DT v[2] = { 0, 0 };
        for (int ch = 0; ch < 2; ch++) {
            int16_t word = 0;
            if (per == 4) word = (int16_t)(p[4 * ch] | (p[4 * ch + 1] << 8) | (p[4 * ch + 2] << 16) | ((unsigned)p[4 * ch + 3] << 24));
            else if (per == 2) word = (int16_t)(p[2 * ch] | (p[2 * ch + 1] << 8));
            else if (per == 1) word = (int16_t)p[0];
            v[ch] = word / full;
        }
        waveData[i] = v[0] + v[1] * I;
        running_time += Ts;
        waveDataTime[i] = running_time;
    }
    free(raw);
    return got;
}

/* wave.c:413-540.  RAW captures: interleaved 32-bit IEEE floats as they are (no scaling); pair by pair, so a file that ends
 * inside a chunk returns the pairs it held; statics of its own for the time axis */
unsigned long int GetComplexRawChunk(FILE *rawFilePtr, HEADER header, DT complex *waveData, DT *waveDataTime, unsigned long int nSamples)
{
    static DT running_time = 0, Ts = 0;
    if (header.channels != 2) {                                    /* (this reader tests the channel count first) */
        printf("Complex read requires 2 channels (I and Q)\n");
        exit(1);
    }
    reader_checks(rawFilePtr, &header, waveData, waveDataTime);
    if (Ts == 0) Ts = 1.0 / (DT)header.sample_rate;
    const long pair = (long)(header.channels * header.bits_per_sample) / 8;
    const long per = pair / (long)header.channels;
    if (per * (long)header.channels != pair) {
        printf("Error: %ld x %ud <> %ld\n", per, header.channels, pair);
        return nSamples;
    }
    unsigned char *p = malloc((size_t)(pair > 0 ? pair : 1));
    if (!p) die("malloc", PDT_ERR_NOMEM);
    for (unsigned long i = 0; i < nSamples; i++) {
        if (fread(p, (size_t)pair, 1, rawFilePtr) != 1) {
            free(p);
            return i;
        }
        if (per != 4) exit(1);                                     /* anything but 32-bit floats ends the program there too */
        float f[2];
        memcpy(f, p, 8);
        waveData[i] = (DT)f[0] + (DT)f[1] * I;
        running_time += Ts;
        waveDataTime[i] = running_time;
    }
    free(p);
    return nSamples;
}
