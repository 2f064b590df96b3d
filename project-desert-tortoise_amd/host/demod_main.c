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
 *   -r opens/creates an empty output.raw (all writes are commented out)     main.c:299-307
 *   44-byte canonical header, no chunk walk            common/wave.c:303-378
 *   every byte after the header is sample data (while(!feof))               main.c:373
 *   output name minorFrames_YYYYMMDD_HHMMSS.txt / packets_YYYYMMDD_HHMMSS.txt   main.c:289 / ARGOS main.c:213
 *   "Normalization Factor: %f", " : PLL locked at %0.2fHz"                  main.c:388, CarrierTrackingPLL.c:269
 *   output removed when no frame was found             main.c:508-512
 * Additions: -o <file> chooses the output name (tests), -d <n> picks the GPU, -q (POES) prints the frame
 * validation the reference keeps in MATLAB (checkParity.m:91-92, daytimeDecode.m:36,82-95) after decoding,
 * -m selects MMClockRecovery (the sampler the reference keeps commented out at ARGOSdemod/main.c:277).
 * Not reproduced: the per-chunk "\r" progress line (there are no chunks on the GPU; one
 * summary line is printed instead).  RAW float32 input (".raw", -s mandatory) is supported for POES
 * exactly as in POESTIPdemod/main.c:313-339.
 */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>
#include <unistd.h>

#include "pdt.h"

#ifdef PDT_ARGOS
#define MODE PDT_MODE_ARGOS
#define DEFAULT_CHUNKSIZE 2400
#define OPTS "rn:c:o:d:m"
#define BANNER "Project Desert Tortoise: Wave file ARGOS Demodulator (MI355X build)\n"
#define PREFIX "packets"
#define UNIT "Packets"
#else
#define MODE PDT_MODE_POES
#define DEFAULT_CHUNKSIZE 10000
#define OPTS "s:rn:c:o:d:qm"
#define BANNER "Project Desert Tortoise: Wave file NOAA TIP Demodulator (MI355X build)\n"
#define PREFIX "minorFrames"
#define UNIT "Frames"
#endif

static const char *get_filename_ext(const char *filename)
{
    const char *dot = strrchr(filename, '.');
    if (!dot || dot == filename) return "";
    return dot + 1;
}

int main(int argc, char **argv)
{
    unsigned long chunkSize = DEFAULT_CHUNKSIZE;
    double normFactor = 0, sampleRate = 0;
    int outputRawFiles = 0, device = 0, quality = 0, sampler = 0, c;
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
    if (chunkSize == DEFAULT_CHUNKSIZE) printf("Using default %ld chunkSize\n", chunkSize);
    if (optind >= argc) {
        printf("No wave file specified\n");
        return 1;
    }
    const char *inFileName = argv[optind];
    printf("%s\n", inFileName);
    printf("Opening IO files..\n");
    FILE *in = fopen(inFileName, "rb");

    time_t t = time(NULL);
    struct tm tm = *localtime(&t);
    if (outOverride)
        snprintf(outFileName, sizeof outFileName, "%s", outOverride);
    else
        snprintf(outFileName, sizeof outFileName, PREFIX "_%4d%02d%02d_%02d%02d%02d.txt", tm.tm_year + 1900, tm.tm_mon + 1,
                 tm.tm_mday, tm.tm_hour, tm.tm_min, tm.tm_sec);
    FILE *out = fopen(outFileName, "w");
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
        long num_samples = (long)((8.0 * data_bytes) / (channels * bits));
        printf("Sample Rate %.2fKHz and %d bits per sample. Total samples %ld\n", (float)rate / 1000.0, bits, num_samples);
    } else {
        rate = (uint32_t)(sampleRate * 1000.0);                      /* main.c:329: entered in kHz */
    }

    /* the reference reads until EOF, not header.data_size */
    fseek(in, 0, SEEK_END);
    long fsz = ftell(in);
    fseek(in, data_offset, SEEK_SET);
    const size_t frame_bytes = is_raw ? 8 : 4;
    uint64_t nframes = fsz > data_offset ? (uint64_t)(fsz - data_offset) / frame_bytes : 0;
    void *samples = malloc(nframes * frame_bytes + 16);
    if (!samples) {
        printf("Error in malloc\n");
        exit(1);
    }
    if (fread(samples, frame_bytes, nframes, in) != nframes) {
        printf("Error reading samples\n");
        exit(1);
    }
    fclose(in);

    pdt_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.mode = MODE;
    cfg.sample_rate = rate;
    cfg.chunk = chunkSize;
    cfg.norm_override = normFactor;
    cfg.device = device;
    cfg.sampler = sampler;
    pdt_ctx *ctx = NULL;
    int rc = pdt_open(&cfg, &ctx);
    if (rc != PDT_OK) {
        printf("GPU demodulator unavailable: %s\n", pdt_strerror(rc));
        fclose(out);
        remove(outFileName);
        exit(1);
    }
    rc = is_raw ? pdt_demod_f32(ctx, (const float *)samples, nframes) : pdt_demod_pcm16(ctx, (const int16_t *)samples, nframes);
    if (rc != PDT_OK) {
        printf("Demodulation failed: %s\n", pdt_strerror(rc));
        fclose(out);
        remove(outFileName);
        exit(1);
    }
    pdt_stats st;
    pdt_get_stats(ctx, &st);
    if (normFactor == 0) printf("Normalization Factor: %f\n", st.norm_factor);
    if (st.lock_sample >= 0) printf(" : PLL locked at %0.2fHz\n", st.lock_freq_hz);

    uint64_t need = pdt_format_frames(ctx, NULL, 0);
    char *text = (char *)malloc(need + 1);
    pdt_format_frames(ctx, text, need);
    fwrite(text, 1, need, out);
#ifdef PDT_ARGOS
    fwrite(text, 1, need, stdout);                                   /* ARGOSdemod/ByteSync.c mirrors to stdout */
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
    free(samples);
    pdt_close(ctx);
    return 0;
}
