/*
 * demod_main.c -- demodPOES / demodARGOS (build with -DPDT_ARGOS) on an MI355X.
 *
 * Host side of the drop-in: plain C, same command line, console messages and output-file
 * surface as the reference programs, with the DSP chain executed by libpdt.so (HIP kernels)
 * through the C ABI of include/pdt.h.  There is no CPU DSP path in this program: without a
 * GPU, pdt_open fails and the program exits with an error.
 *
 * Behaviour mirrored from the reference (file:line):
 *   options -s <kHz> -r -n <gain> -c <chunk>           POESTIPdemod/main.c:185-234
 *           -r -n -c                                   ARGOSdemod/main.c:121-164
 *   -s with a WAV overrides the rate with the kHz number taken as Hz (Q6)   main.c:343-344
 *   -r POES: opens/creates an empty output.raw (all writes are commented out)   main.c:299-307
 *      ARGOS: output.raw receives the AGC output before Squelch (doubles)      ARGOSdemod/main.c:171-180,273-274
 *   44-byte canonical header, no chunk walk            common/wave.c:303-378
 *   every byte after the header is sample data (while(!feof))               main.c:373
 *   output name minorFrames_YYYYMMDD_HHMMSS.txt / packets_YYYYMMDD_HHMMSS.txt   main.c:289 / ARGOS main.c:213
 *   "Normalization Factor: %f", " : PLL locked at %0.2fHz"                  main.c:388, CarrierTrackingPLL.c:269
 *   output removed when no frame was found             main.c:508-512
 * A capture of any length is taken (main.c:373 reads until end of file): one that does not fit the GPU's free memory goes through
 * the library's bounded window (pdt.h, ABI 4).
 * Additions: -o <file> chooses the output name (tests), -d <n> picks the GPU, -D NAME=VALUE sets a developer switch of the library, -P drops the progress lines, -q (POES) prints the frame
 * validation the reference keeps in MATLAB (checkParity.m:91-92, daytimeDecode.m:36,82-95) after decoding,
 * -m selects MMClockRecovery (the sampler the reference keeps commented out at ARGOSdemod/main.c:277),
 * -l (POES) runs the sound-card twin's chain (POESTIPdemodPortAudio/main.c:41-65,324-393: its PLL constants,
 * Squelch between PLL and FIR, Manchester threshold 0.75, blocks of 2400); with the file name "-" it is the twin's
 * loop itself, reading float32 I,Q blocks from standard input (e.g. a sound-card recorder's pipe, -s 48) until
 * end of file and appending every minor frame to the output as soon as it is final.
 * The per-chunk "\r" progress line (main.c:461-481 with its quality figure, ARGOSdemod/main.c:290-296) is printed after the
 * run, chunk by chunk from the library's per-chunk reports (pdt_keep_quality), with the reference's arithmetic and under the
 * reference's condition (progress of more than 0.15 %, or end of file); one summary line follows.  RAW float32 input (".raw", -s mandatory) is supported for POES
 * exactly as in POESTIPdemod/main.c:313-339.
 */
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>
#include <unistd.h>

#include "pdt.h"
#include "pdt_dev.h"                   /* -D NAME=VALUE: the library's developer switches (tests; the library never reads the environment) */

#ifdef PDT_ARGOS
#define MODE PDT_MODE_ARGOS
#define DEFAULT_CHUNKSIZE 2400
#define OPTS "s:rn:c:o:d:mlPTD:"      /* -l (round 4): the sound-card twin's chain, -s its rate in kHz when the samples come from a pipe */
#define BANNER "Project Desert Tortoise: Wave file ARGOS Demodulator (MI355X build)\n"
#define PREFIX "packets"
#define UNIT "Packets"
#else
#define MODE PDT_MODE_POES
#define DEFAULT_CHUNKSIZE 10000
#define OPTS "s:rn:c:o:d:qmlPTD:"
#define BANNER "Project Desert Tortoise: Wave file NOAA TIP Demodulator (MI355X build)\n"
#define PREFIX "minorFrames"
#define UNIT "Frames"
#endif

/* The reference's progress line, chunk by chunk (POESTIPdemod/main.c:457-481, ARGOSdemod/main.c:286-296).  The loop runs once
 * more with zero samples when the data end exactly at a chunk boundary (fread does not set the end-of-file flag before it
 * comes up short, wave.c:125; main.c:372): that pass prints the line again, feof() being true at last.                      */
#define ANSI_COLOR_RED "\x1b[31m"
#define ANSI_COLOR_GREEN "\x1b[32m"
#define ANSI_COLOR_YELLOW "\x1b[33m"
#define ANSI_COLOR_RESET "\x1b[0m"
/* The lines are printed from the library's progress function (pdt_set_progress): a large capture is demodulated in segments
 * while it is still being read, and every finished segment's chunks are reported while the next one runs.                      */
typedef struct progress_state {
    long num_samples;
    unsigned long chunkSize;
    uint64_t total_chunks;                 /* of the capture */
    int extra;                             /* the zero-sample pass at the end of the file */
    int norm_wanted, norm_printed, lock_printed, any;
    unsigned long i, totalSymbols, totalBits, totalSamples;
    int totalFrames;
#ifdef PDT_ARGOS
    double percentComplete;
#else
    float percentComplete;
#endif
} progress_state;

static void print_norm_and_lock(progress_state *P, const pdt_stats *st)
{
    if (P->norm_wanted && !P->norm_printed) {
        printf("Normalization Factor: %f\n", st->norm_factor);                  /* main.c:420 */
        P->norm_printed = 1;
    }
    if (!P->lock_printed && st->lock_sample >= 0) {
        printf(" : PLL locked at %0.2fHz\n", st->lock_freq_hz);                 /* CarrierTrackingPLL.c:269 */
        P->lock_printed = 1;
    }
}

static void progress_line(progress_state *P, const pdt_chunk_report *q, int last)
{
#ifdef PDT_ARGOS
    if ((((double)(P->i) / P->num_samples) * 100.0 - P->percentComplete > 0.15) || last) {
        P->percentComplete = ((double)(P->i) / P->num_samples) * 100.0;
        printf("\r");
        printf("%0.1f%% %0.3f Ks : %0.1f Sec: %ld Sym : %ld Bits : %d Packets", ((double)(P->i) / P->num_samples) * 100.0,
               (P->totalSamples) / 1000.0, q->time0, P->totalSymbols, P->totalBits, P->totalFrames);
    }
#else
    float averagePhase;
    char qualityString[20];
    if ((((float)(P->i) / P->num_samples) * 100.0 - P->percentComplete > 0.15) || last) {
        P->percentComplete = ((float)(P->i) / P->num_samples) * 100.0;
        averagePhase = (float)q->avg_phase;
        printf("\r");
        printf("%f\t", fabs(M_PI / 2.0 - averagePhase));
        averagePhase = 10.0 * log10f(powf(fabs(M_PI / 2.0 - averagePhase), 2));
        if (averagePhase > -4.3)
            snprintf(qualityString, 20, "%s%02.1fQ%s", ANSI_COLOR_GREEN, averagePhase, ANSI_COLOR_RESET);
        else if (averagePhase > -5)
            snprintf(qualityString, 20, "%s%02.1fQ%s", ANSI_COLOR_YELLOW, averagePhase, ANSI_COLOR_RESET);
        else if (averagePhase > -6)
            snprintf(qualityString, 20, "%s%02.1fQ%s", ANSI_COLOR_YELLOW, averagePhase, ANSI_COLOR_RESET);
        else
            snprintf(qualityString, 20, "%s%02.1fQ%s", ANSI_COLOR_RED, averagePhase, ANSI_COLOR_RESET);
        printf("%0.1f%% %0.3f Ks : %0.1f Sec: %ld Sym : %ld Bits : %d Frames : %s   ", ((float)(P->i) / P->num_samples) * 100.0,
               (P->totalSamples) / 1000.0, (float)q->time0, P->totalSymbols, P->totalBits, P->totalFrames, qualityString);
    }
#endif
}

static void on_progress(void *user, uint64_t first_chunk, const pdt_chunk_report *r, uint64_t n, const pdt_stats *so_far)
{
    progress_state *P = (progress_state *)user;
    print_norm_and_lock(P, so_far);
    for (uint64_t k = 0; k < n; k++) {
        const pdt_chunk_report *q = &r[k];
        const int final_chunk = first_chunk + k + 1 == P->total_chunks;
        P->i += q->samples;
        P->totalBits += q->bits;
        P->totalFrames += (int)q->frames;
        P->totalSymbols += q->symbols;
        P->totalSamples += q->samples;
        progress_line(P, q, final_chunk && !P->extra);                 /* last: feof(inFilePtr) */
        if (final_chunk && P->extra) progress_line(P, q, 1);
        if (final_chunk) printf("\n");
    }
    P->any = 1;
    fflush(stdout);
}

static const char *get_filename_ext(const char *filename)
{
    const char *dot = strrchr(filename, '.');
    if (!dot || dot == filename) return "";
    return dot + 1;
}

static void put_frames(FILE *out, const pdt_frame *f, uint64_t n)
{
    for (uint64_t k = 0; k < n; k++) {                                /* POESTIPdemod/ByteSync.c:62-69,96-101 */
        fprintf(out, f[k].inverted ? "%.5fi " : "%.5f ", f[k].time);
        for (unsigned b = 0; b < f[k].nbytes; b++) fprintf(out, "%.2X ", f[k].bytes[b]);
        if (f[k].complete) fprintf(out, "\n");
    }
    fflush(out);
}

/* The twin's loop (POESTIPdemodPortAudio/main.c:324-393, ARGOSdemodPortAudio/main.c:266-329) with standard input as the sound card: blocks of `chunk`
 * float32 I,Q frames until end of file (there: until a key is hit); frames are appended as they become final. */
static int live_loop(FILE *in, FILE *out, const char *outFileName, double sampleRate, unsigned long chunk, double normFactor,
                     int device, int sampler)
{
    if (sampleRate < 1) sampleRate = 48.0;                            /* twin: SAMPLE_RATE 48000 (main.c:27) */
    pdt_config cfg;
    memset(&cfg, 0, sizeof cfg);
#ifdef PDT_ARGOS
    cfg.mode = PDT_MODE_ARGOS;                                        /* + PDT_CHAIN_LIVE: the float build of the ARGOS chain */
#else
    cfg.mode = PDT_MODE_POES;
#endif
    cfg.sample_rate = (uint32_t)(sampleRate * 1000.0);
    cfg.chunk = chunk;
    cfg.norm_override = normFactor;
    cfg.device = device;
    cfg.sampler = sampler;
    cfg.chain = PDT_CHAIN_LIVE;
    pdt_ctx *ctx = NULL;
    int rc = pdt_open(&cfg, &ctx);
    if (rc != PDT_OK) {
        printf("GPU demodulator unavailable: %s\n", pdt_strerror(rc));
        fclose(out);
        remove(outFileName);
        return 1;
    }
    float *block = (float *)malloc(sizeof(float) * 2 * chunk);
    pdt_frame *fr = NULL;
    uint64_t cap = 0, total = 0, samples = 0, fresh = 0;
    if (!block || pdt_stream_begin(ctx) != PDT_OK) {
        printf("Error in malloc\n");
        return 1;
    }
    for (;;) {
        const size_t got = fread(block, 2 * sizeof(float), chunk, in);
        if (got) {
            rc = pdt_stream_push_f32(ctx, block, got, &fresh);
        } else {
            rc = pdt_stream_end(ctx, &fresh);
        }
        if (rc != PDT_OK) {
            printf("Demodulation failed: %s\n", pdt_strerror(rc));
            return 1;
        }
        if (fresh > cap) {
            cap = fresh + 64;
            fr = (pdt_frame *)realloc(fr, cap * sizeof *fr);
            if (!fr) return 1;
        }
        if (fresh) put_frames(out, fr, pdt_stream_frames(ctx, fr, fresh));
        total += fresh;
        samples += got;
        if (!got) break;
        printf("\r%0.1fKsps :%0.3f Sec: %llu Frames", sampleRate, (double)samples / (sampleRate * 1000.0), (unsigned long long)total);
        fflush(stdout);
    }
    pdt_stats st;
    pdt_get_stats(ctx, &st);
    if (st.lock_sample >= 0) printf("\n : PLL locked at %0.2fHz", st.lock_freq_hz);
    printf("\nNormalization Factor: %f\n%llu samples, %llu " UNIT "\n", st.norm_factor, (unsigned long long)samples,
           (unsigned long long)total);
    fclose(out);
    if (total == 0) remove(outFileName);
    free(block);
    free(fr);
    pdt_close(ctx);
    return 0;
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

int main(int argc, char **argv)
{
    const double t_main = now_ms();                      /* -T: where the run's time goes, one JSON line on stderr at the end */
    int timing = 0;
    unsigned long chunkSize = DEFAULT_CHUNKSIZE;
    double normFactor = 0, sampleRate = 0;
    int outputRawFiles = 0, device = 0, quality = 0, sampler = 0, live = 0, chunkGiven = 0, noProgress = 0, c;
    const char *outOverride = NULL;
    char outFileName[1100];

    printf(BANNER);
    while ((c = getopt(argc, argv, OPTS)) != -1) {
        switch (c) {
        case 's':
            sampleRate = atof(optarg);
            printf("Sample Rate Set To %f Khz\n", sampleRate);
            break;
        case 'r':
            outputRawFiles = 1;
            printf("Outputting Debugging Raw Files\n");
            break;
        case 'n':
            normFactor = atof(optarg);
            printf("Static Gain Override %f\n", normFactor);
            break;
        case 'c':
            chunkSize = (unsigned long)atoi(optarg);
            chunkGiven = 1;
            if (chunkSize != DEFAULT_CHUNKSIZE) printf("Override: Using %ld chunkSize\n", chunkSize);
            break;
        case 'o':
            outOverride = optarg;
            break;
        case 'd':
            device = atoi(optarg);
            break;
        case 'q':
            quality = 1;
            break;
        case 'T':
            timing = 1;
            break;
        case 'D': {                                     /* developer switch, e.g. -D PDT_HBM_LIMIT_MB=2048 (before the context opens) */
            char name[64];
            const char *eq = strchr(optarg, '=');
            const size_t len = eq ? (size_t)(eq - optarg) : strlen(optarg);
            if (len == 0 || len >= sizeof name) return 1;
            memcpy(name, optarg, len);
            name[len] = 0;
            pdt_dev_set(name, eq ? eq + 1 : "1");
            break;
        }
        case 'P':                                       /* no progress lines (and no averagePhase pass on the GPU) */
            noProgress = 1;
            break;
        case 'l':                                       /* the sound-card twin's chain */
            live = 1;
            printf("Using the live (sound card) chain\n");
            break;
        case 'm':                                       /* MMClockRecovery instead of Gardner (ARGOSdemod/main.c:277) */
            sampler = PDT_SAMPLER_MM;
            printf("Using M&M clock recovery\n");
            break;
        case '?':
            if (optopt == 's' || optopt == 'c' || optopt == 'n')
                fprintf(stderr, "Option -%c requires an argument.\n", optopt);
            else if (isprint(optopt))
                fprintf(stderr, "Unknown option `-%c'.\n", optopt);
            else
                fprintf(stderr, "Unknown option character `\\x%x'.\n", optopt);
            return 1;
        default:
            abort();
        }
    }
    if (live && !chunkGiven) chunkSize = 2400;          /* POESTIPdemodPortAudio/main.c:34 */
    if (chunkSize == (live ? 2400 : DEFAULT_CHUNKSIZE)) printf("Using default %ld chunkSize\n", chunkSize);
    if (optind >= argc) {
        printf("No wave file specified\n");
        return 1;
    }
    const char *inFileName = argv[optind];
    printf("%s\n", inFileName);
    printf("Opening IO files..\n");
    const int from_stdin = live && strcmp(inFileName, "-") == 0;
    FILE *in = from_stdin ? stdin : fopen(inFileName, "rb");

    time_t t = time(NULL);
    struct tm tm = *localtime(&t);
    if (outOverride)
        snprintf(outFileName, sizeof outFileName, "%s", outOverride);
    else
        snprintf(outFileName, sizeof outFileName, PREFIX "_%4d%02d%02d_%02d%02d%02d.txt", tm.tm_year + 1900, tm.tm_mon + 1,
                 tm.tm_mday, tm.tm_hour, tm.tm_min, tm.tm_sec);
    FILE *out = fopen(outFileName, "w+");
    if (!in || !out) {
        printf("Error opening output files\n");
        exit(1);
    }
    if (outputRawFiles) {
        FILE *raw = fopen("output.raw", "wb");
        if (!raw) {
            printf("Error opening output file\n");
            exit(1);
        }
        fclose(raw);
    }

    if (from_stdin) return live_loop(in, out, outFileName, sampleRate, chunkSize, normFactor, device, sampler);
    int is_raw = 0;
    if (strcasecmp(get_filename_ext(inFileName), "wav") != 0) {
#ifdef PDT_ARGOS
        printf("RAW files not yet supported :(\n");
        exit(1);
#else
        if (strcasecmp(get_filename_ext(inFileName), "raw") == 0) {
            if (sampleRate < 1) {                                     /* main.c:317-321 */
                printf("Sample Rate (in Khz) must be specified when using RAW files\n");
                exit(1);
            }
            printf("Assuming 32-bit IEEE Floating Point RAW input\n");
            is_raw = 1;
        } else {
            printf("Unrecognized file format %s\n", get_filename_ext(inFileName));
            exit(1);
        }
#endif
    }

    uint32_t rate = 0, channels = 2, bits = 32, format = 1, data_bytes = 0;
    long data_offset = 0;
    long num_samples = 44515000;                                     /* RAW: main.c:337 (progress bar only) */
    if (!is_raw) {
        uint8_t hdr[44];
        if (fread(hdr, 1, 44, in) != 44) {
            printf("Error reading WAV header\n");
            exit(1);
        }
        pdt_wav_parse_header(hdr, &rate, &channels, &bits, &format, &data_bytes);
        data_offset = 44;
        if (channels != 2) {
            printf("Complex read requires 2 channels (I and Q)\n");
            exit(1);
        }
        if (format != 1) {
            printf("Only PCM is currently supported :(\n");
            exit(1);
        }
        if (bits != 16) {
            printf("Only 16-bit PCM is supported by the MI355X build (the reference truncates other widths, Q5)\n");
            exit(1);
        }
#ifndef PDT_ARGOS
        if (sampleRate > 1) rate = (uint32_t)sampleRate;             /* main.c:343-344 (Q6) */
#endif
#ifdef PDT_ARGOS
        num_samples = (long)((uint32_t)(8u * data_bytes) / (channels * bits));   /* ARGOSdemod/main.c:244, unsigned int arithmetic */
#else
        num_samples = (long)(unsigned long)((8.0 * data_bytes) / (channels * bits));   /* main.c:349 */
#endif
        printf("Sample Rate %.2fKHz and %d bits per sample. Total samples %ld\n", (float)rate / 1000.0, bits, num_samples);
    } else {
        rate = (uint32_t)(sampleRate * 1000.0);                      /* main.c:329: entered in kHz */
    }

    /* the reference reads until EOF, not header.data_size; the library reads the file itself (threaded reads into pinned
     * memory overlapped with the copy to the GPU: pdt_demod_fd) */
    fseek(in, 0, SEEK_END);
    long fsz = ftell(in);
    const size_t frame_bytes = is_raw ? 8 : 4;
    uint64_t nframes = fsz > data_offset ? (uint64_t)(fsz - data_offset) / frame_bytes : 0;

    pdt_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.mode = MODE;
    cfg.sample_rate = rate;
    cfg.chunk = chunkSize;
    cfg.norm_override = normFactor;
    cfg.device = device;
    cfg.sampler = sampler;
    cfg.chain = live ? PDT_CHAIN_LIVE : PDT_CHAIN_FILE;
    pdt_ctx *ctx = NULL;
    const double t_open0 = now_ms();
    int rc = pdt_open(&cfg, &ctx);
    const double t_open1 = now_ms();
    if (rc != PDT_OK) {
        printf("GPU demodulator unavailable: %s\n", pdt_strerror(rc));
        fclose(out);
        remove(outFileName);
        exit(1);
    }
#ifdef PDT_ARGOS
    if (outputRawFiles) pdt_keep_presquelch(ctx, 1);                 /* -r: the AGC output before Squelch, ARGOSdemod/main.c:273-274 */
#endif
    pdt_keep_pll(ctx, 0);                                            /* nothing here reads the PLL output stream */
    progress_state prog;
    memset(&prog, 0, sizeof prog);
    prog.num_samples = num_samples;
    prog.chunkSize = (unsigned long)chunkSize;
    prog.total_chunks = (nframes + (uint64_t)chunkSize - 1) / (uint64_t)chunkSize;
    prog.extra = nframes % (uint64_t)chunkSize == 0;
    prog.norm_wanted = normFactor == 0;
    if (!noProgress && num_samples > 0) {                            /* the chunk loop's progress / quality line */
        pdt_keep_quality(ctx, 1);
        pdt_set_progress(ctx, on_progress, &prog);
    }
    /* POES: the one-call form -- a large file is demodulated in segments while it is still being read; a finished segment's
     * text goes to the file (as the reference's fprintf calls do, ByteSync.c:62-101) and its progress lines to the console
     * while the next one runs */
    int text_written = 0;
    fflush(out);
#ifndef PDT_ARGOS
    rc = pdt_demod_file(ctx, fileno(in), (uint64_t)data_offset, nframes, is_raw ? PDT_FMT_F32 : PDT_FMT_PCM16, fileno(out), NULL);
    text_written = 1;
#else
    rc = pdt_demod_fd(ctx, fileno(in), (uint64_t)data_offset, nframes, is_raw ? PDT_FMT_F32 : PDT_FMT_PCM16);
#endif
    const double t_demod1 = now_ms();
    fclose(in);
    if (rc != PDT_OK) {
        printf("Demodulation failed: %s\n", pdt_strerror(rc));
        fclose(out);
        remove(outFileName);
        exit(1);
    }
#ifdef PDT_ARGOS
    if (rc == PDT_OK && outputRawFiles) {
        /* ARGOSdemod -r: every chunk's post-AGC, pre-Squelch doubles, appended to output.raw (main.c:171-180,273-274) */
        FILE *raw = fopen("output.raw", "wb");
        /* (the sound-card twin, -l, is the float build: its DECIMAL_TYPE -- and this context's stage elements -- are floats) */
        const size_t es = live ? sizeof(float) : sizeof(double);
        const uint64_t total = pdt_stage_len(ctx, PDT_ST_AGC_RAW), piece = 1u << 20;
        void *tmp = malloc((size_t)piece * sizeof(double));
        if (!raw || !tmp) {
            printf("Error opening output file\n");
            exit(1);
        }
        for (uint64_t at = 0; at < total; at += piece) {
            const int64_t got = pdt_read_stage(ctx, PDT_ST_AGC_RAW, at, piece, tmp);
            if (got <= 0) break;
            fwrite(tmp, es, (size_t)got, raw);
        }
        fclose(raw);
        free(tmp);
    }
#endif
    pdt_stats st;
    pdt_get_stats(ctx, &st);
    print_norm_and_lock(&prog, &st);                                 /* (what the progress function has not printed) */

    char *text = NULL;
    fflush(out);
    const double t_text0 = now_ms();
    if (!text_written && pdt_write_frames(ctx, fileno(out), NULL) != PDT_OK) {        /* the minor frames / packets, all at once */
        printf("Error writing output file\n");
        exit(1);
    }
    const double t_text1 = now_ms();
#ifdef PDT_ARGOS
    {
        uint64_t need = pdt_format_frames(ctx, NULL, 0);
        text = (char *)malloc(need + 1);
        pdt_format_frames(ctx, text, need);
        fwrite(text, 1, need, stdout);                               /* ARGOSdemod/ByteSync.c mirrors to stdout */
    }
#endif
    printf("100.0%% %0.3f Ks : %llu Sym : %llu Bits : %llu " UNIT "   (GPU %.3f ms)\n", st.samples / 1000.0,
           (unsigned long long)st.symbols, (unsigned long long)st.bits, (unsigned long long)st.frames, st.gpu_ms);

#ifndef PDT_ARGOS
    if (quality) {
        pdt_tip_summary q;
        if (pdt_tip_check(ctx, &q) == PDT_OK) {
            const char *name = q.spacecraft == 8 ? "NOAA-15" : q.spacecraft == 13 ? "NOAA-18" : q.spacecraft == 15 ? "NOAA-19" : "A UFO!";
            printf("\n%llu out of %llu Error Free Frames\n\n", (unsigned long long)q.good_frames, (unsigned long long)q.frames_checked);
            printf("%llu Good Chunks and %llu Bad Chunks\n\n", (unsigned long long)q.good_chunks, (unsigned long long)q.bad_chunks);
            if (q.t0_ms >= 0) {
                const double h = (double)q.t0_ms / 3600000.0;
                const int hh = (int)h, mm = (int)((h - hh) * 60.0);
                printf("T0 Best Guess: %lld which is %d:%d:%g\n", (long long)q.t0_ms, hh, mm, ((h - hh) * 60.0 - mm) * 60.0);
            }
            printf("Spacecraft: %d=>%s\n", q.spacecraft, name);
            if (q.day >= 0) printf("Julean Day: %d \n", q.day);
        }
    }
#else
    (void)quality;
#endif
    time_t t2 = time(NULL);
    struct tm tm2 = *localtime(&t2);
    printf("\nThat took %d seconds!\n", (tm2.tm_min * 60 + tm2.tm_sec) - (tm.tm_min * 60 + tm.tm_sec));
    if (fclose(out)) {
        printf("error closing file.");
        exit(-1);
    }
    if (st.frames == 0) {
        printf("\n\nNone bits found :(\nRemoving output file and exiting.\nMAY YOU HAVE MORE BETTER BITS ANOTHER DAY\n");
        remove(outFileName);
    } else {
        printf("\nAll done! Closing files and exiting.\nENJOY YOUR BITS AND HAVE A NICE DAY\n");
    }
    free(text);
    const double t_close0 = now_ms();
    pdt_get_stats(ctx, &st);
    /* (leaving the ~25 GB of buffers to the driver -- no pdt_close, _exit -- was measured: this process ends 30 ms sooner and the
     * NEXT one waits 0.8 s in its runtime start-up while the driver reclaims them) */
    pdt_close(ctx);
    if (timing)          /* (process start -> main and the dynamic loader's share are the caller's wall time minus `total`) */
        fprintf(stderr, "{\"timing_ms\": {\"main_to_open\": %.2f, \"open_hip_ready\": %.2f, \"demod_call\": %.2f, \"of_it_alloc\": %.2f, "
                        "\"of_it_ingest\": %.2f, \"of_it_gpu_last_run\": %.2f, \"progress_and_stats\": %.2f, \"text_write\": %.2f, \"report_and_close_file\": %.2f, "
                        "\"context_close\": %.2f, \"total\": %.2f}}\n",
                t_open0 - t_main, t_open1 - t_open0, t_demod1 - t_open1, st.alloc_ms, st.ingest_ms, st.gpu_ms, t_text0 - t_demod1, t_text1 - t_text0,
                t_close0 - t_text1, now_ms() - t_close0, now_ms() - t_main);
    return 0;
}
