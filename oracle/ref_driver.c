/*
 * oracle/ref_driver.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * A small chunk-loop harness around the *unmodified* reference DSP objects
 * (oracle/_ref/libref_{poes,argos}.so, compiled straight from
 * /root/reference/common/*.c and the per-program ByteSync.c by oracle/Makefile).
 *
 * The reference mains (POESTIPdemod/main.c, ARGOSdemod/main.c) include the
 * Windows-only <conio.h> and therefore cannot be built in this image without a
 * stand-in header, so the ~40 lines of chunk loop they contain are restated
 * here.  Everything numerically relevant -- every DSP stage, the WAV reader,
 * the byte synchroniser and its fprintf formatting -- is the reference's own
 * compiled code.  The restated parts, with the lines they follow:
 *   - buffer allocation order / sizes     POESTIPdemod/main.c:241-250,355-357
 *                                          ARGOSdemod/main.c:169-176
 *   - interp / tap-count selection         POESTIPdemod/main.c:346-348
 *   - call-site constant expressions       POESTIPdemod/main.c:413,419,429,438,445,454
 *                                          ARGOSdemod/main.c:248,265-284
 *   - while(!feof) loop and first-chunk StaticGain   POESTIPdemod/main.c:373-389
 * The survey's golden (real main.c + empty conio.h) for 5sec_clip.wav is
 * md5 d3c496d003a29eeee061c01b00ce025c; tests/test_oracle_ref.py checks this
 * harness reproduces it, which pins the restated loop to the real program.
 *
 * (-DARGOS -DARGOS_FLOAT, linked against libref_argosf.so: the ARGOS sound-card twin's float build of the same loop,
 *  ARGOSdemodPortAudio/main.c:266-329 -- its buffer order :58-65, its time stamps :285-286, its ByteSync.c)
 * usage: ref_demod{POES,ARGOS} [-c chunk] [-n gain] [-s kHz] [-d dumpprefix] [-M -R r -K k] in.wav out.txt
 *        ref_demod{POES,ARGOS} -B [-c piece] bits.txt out.txt
 *        ref_demodPOES -L -s kHz [-c block] in.raw out.txt        (sound-card twin's composition)
 *   -d prefix : additionally dump every stage's per-chunk output, concatenated
 *               over chunks, to <prefix>.{pll,fir,agc,sym,symt,bits,bitt,lock}
 *               and per-chunk counts to <prefix>.counts (text).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <complex.h>
#include <math.h>
#include <unistd.h>
#include <strings.h>
#include <time.h>

#if defined(ARGOS) && !defined(ARGOS_FLOAT)
#define DECIMAL_TYPE double
#else
#define DECIMAL_TYPE float     /* POES; ARGOS_FLOAT: the ARGOS sound-card twin's build (ARGOSdemodPortAudio/config.h: USE_FLOATS 1) */
#endif
#define DT DECIMAL_TYPE

#include "wave.h"
#include "AGC.h"
#include "CarrierTrackPLL.h"
#include "LowPassFilter.h"
#include "GardenerClockRecovery.h"
#include "MMClockRecovery.h"
#include "ManchesterDecode.h"
#ifdef ARGOS
int FindSyncWords(unsigned char *, DT *, unsigned long, char *, unsigned int, FILE *);
#else
int ByteSyncOnSyncword(unsigned char *, DT *, unsigned long, char *, unsigned int, FILE *);
#endif

/* bench.py's cpu_baseline leg: seconds spent in the DSP stages (everything after the chunk has been read and converted),
 * printed beside the totals so that a DSP-only rate can be quoted next to the end-to-end one */
static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static FILE *dopen(const char *prefix, const char *ext)
{
    char name[1200];
    if (!prefix) return NULL;
    snprintf(name, sizeof name, "%s.%s", prefix, ext);
    FILE *f = fopen(name, "wb");
    if (!f) { perror(name); exit(2); }
    return f;
}
static void dput(FILE *f, const void *p, size_t sz, size_t n) { if (f && n) fwrite(p, sz, n, f); }

int main(int argc, char **argv)
{
    unsigned long chunk =
#ifdef ARGOS
        2400;   /* ARGOSdemod/main.c:27 */
#else
        10000;  /* POESTIPdemod/main.c:30 */
#endif
    DT normFactor = 0, sampleRate = 0;
    const char *dump = NULL;
    int use_mm = 0;              /* -M: MMClockRecovery at the sampler's call site (ARGOSdemod/main.c:277, commented out there) */
    DT mmRange = 3, mmKp = 0.15;
    double loopK[11] = {0};
    int haveK = 0;
    int c;
    int live = 0;                /* -L: the sound-card twin's composition (POESTIPdemodPortAudio/main.c:324-393) over a
                                    float32 file: the twin's constants, lock stream kept, Squelch between PLL and FIR */
    int bits_only = 0;           /* -B: in.wav is a text file of '0'/'1'; run only the reference's byte synchroniser on it */
    while ((c = getopt(argc, argv, "c:n:s:d:MR:K:BLk:")) != -1) {
        if (c == 'L') { live = 1; continue; }
        if (c == 'M') { use_mm = 1; continue; }
        if (c == 'B') { bits_only = 1; continue; }
        if (c == 'R') { mmRange = atof(optarg); continue; }
        if (c == 'K') { mmKp = atof(optarg); continue; }
        if (c == 'k') {        /* loop constants other than the mains' (POES, file chain): eleven numbers, hex floats welcome --
                                  freqRange, lock threshold, lockSigAlpha, loopbw_acq, loopbw_track, AGC attack, decay, baud, stepRange, kp, resync threshold */
            char *q = optarg;
            for (int j = 0; j < 11; j++) { loopK[j] = strtod(q, &q); if (*q == ',') q++; }
            haveK = 1;
            continue;
        }
        if (c == 'c') chunk = atoi(optarg);
        else if (c == 'n') normFactor = atof(optarg);
        else if (c == 's') sampleRate = atof(optarg);
        else if (c == 'd') dump = optarg;
        else return 2;
    }
    if (argc - optind < 2) { fprintf(stderr, "usage: %s [opts] in.wav out.txt\n", argv[0]); return 2; }

    if (bits_only) {
        /* the reference's own (commented-out) harness, POESTIPdemod/ByteSync.c:6-14: a literal bit string handed to the
         * synchroniser; here in pieces of `chunk` bits as the demodulator's loop hands them over, bit k stamped with time k */
        FILE *bf = fopen(argv[optind], "rb"), *out = fopen(argv[optind + 1], "w");
        if (!bf || !out) { fprintf(stderr, "cannot open files\n"); return 1; }
        unsigned char *bits = malloc(chunk);
        DT *t = malloc(sizeof(DT) * chunk);
        unsigned long k = 0, frames = 0, n;
        int ch = 0;
        while (ch != EOF) {
            n = 0;
            while (n < chunk && (ch = fgetc(bf)) != EOF)
                if (ch == '0' || ch == '1') { bits[n] = (unsigned char)ch; t[n] = (DT)(k++); n++; }
#ifdef ARGOS
            frames += FindSyncWords(bits, t, n, "0001011110000", 13, out);
#else
            frames += ByteSyncOnSyncword(bits, t, n, "1110110111100010000", 19, out);
#endif
        }
        fclose(out);
        fprintf(stderr, "bits %lu frames %lu\n", k, frames);
        return 0;
    }

    /* allocation order as in the reference main (heap layout matters for the
     * reads one-past-the-chunk) */
#ifdef ARGOS
    DT *filterCoeffs = malloc(sizeof(DT) * 50);
#endif
    char *inFileName = malloc(1024);
    DT complex *waveData = malloc(sizeof(DT complex) * chunk);
#ifdef ARGOS_FLOAT
    float *waveFrame = malloc(sizeof(float) * chunk * 2);      /* ARGOSdemodPortAudio/main.c:60: between waveData and waveDataTime */
    (void)waveFrame;
#endif
    DT *waveDataTime = malloc(sizeof(DT) * chunk);
    DT *dataStreamReal = malloc(sizeof(DT) * chunk);
#ifdef ARGOS
    DT *lockSignalStream = malloc(sizeof(DT) * chunk);
#else
    DT *lockSignalStream = live ? malloc(sizeof(DT) * chunk) : NULL;   /* the file program has none; the twin allocates it after dataStreamReal */
#endif
    DT *dataStreamSymbols = malloc(sizeof(DT) * chunk);
    unsigned char *dataStreamBits = malloc(chunk);
    strcpy(inFileName, argv[optind]);

    FILE *in = fopen(inFileName, "rb");
    FILE *out = fopen(argv[optind + 1], "w");
    if (!in || !out) { fprintf(stderr, "cannot open files\n"); return 1; }

    HEADER header;
    memset(&header, 0, sizeof header);
    int is_raw = 0;
#ifndef ARGOS
    {
        const char *dot = strrchr(inFileName, '.');
        is_raw = dot && strcasecmp(dot + 1, "raw") == 0;
    }
    if (is_raw) {                                               /* POESTIPdemod/main.c:313-339 */
        if (sampleRate < 1) { fprintf(stderr, "Sample Rate (in Khz) must be specified when using RAW files\n"); return 1; }
        header.type = 1;
        header.sample_rate = sampleRate * 1000.0;
        header.channels = 2;
        header.bits_per_sample = 32;
    } else
#endif
    {
        header = ReadWavHeader(in);
#ifndef ARGOS
        if (sampleRate > 1) header.sample_rate = sampleRate;   /* main.c:343-344 (Q6) */
#endif
    }
    DT Fs = (DT)header.sample_rate;
#ifdef ARGOS
    long num_samples = (8 * header.data_size) / (header.channels * header.bits_per_sample);        /* ARGOSdemod/main.c:244 */
#else
    unsigned long num_samples = 44515000;                                                         /* main.c:337 */
    if (!is_raw) num_samples = (8.0 * header.data_size) / (header.channels * header.bits_per_sample);   /* main.c:349 */
#endif

#ifdef ARGOS
    const int interp = 1, N = 50;
    MakeLPFIR(filterCoeffs, 50, 700, Fs, 1);                    /* ARGOSdemod/main.c:248 */
#else
    int interp = rint(150000.0 / Fs);                           /* main.c:347 */
    int N = 26 * interp;                                        /* main.c:348 */
    DT *filterCoeffs = malloc(sizeof(DT) * N);                  /* main.c:355-357 */
    DT *dataStreamLPF = malloc(sizeof(DT) * chunk * N);
    DT *dataStreamLPFTime = malloc(sizeof(DT) * chunk * N);
    MakeLPFIR(filterCoeffs, N, 11000.0, Fs * interp, interp);   /* main.c:369 */
#endif

    FILE *dpll = dopen(dump, "pll"), *dfir = dopen(dump, "fir"), *dagc = dopen(dump, "agc"),
         *dsym = dopen(dump, "sym"), *dsymt = dopen(dump, "symt"), *dbits = dopen(dump, "bits"),
         *dbitt = dopen(dump, "bitt"), *dcnt = dopen(dump, "counts"), *dlock = dopen(dump, "lock"),
         *dtaps = dopen(dump, "taps"), *diq = dopen(dump, "iq"), *dtime = dopen(dump, "time"), *dagcraw = dopen(dump, "agcraw");
    dput(dtaps, filterCoeffs, sizeof(DT), N);

    unsigned long i = 0, nSamples, nSymbols, nBits, totalFrames = 0;
    double dsp_s = 0, t_dsp;
    /* the progress line of the chunk loop (POESTIPdemod/main.c:457-481, ARGOSdemod/main.c:286-296), written to <dump>.progress
     * instead of the console; <dump>.avg = CarrierTrackPLL's return value of every pass of the loop */
    FILE *dprog = dopen(dump, "progress"), *davg = dopen(dump, "avg");
    unsigned long totalSymbols = 0, totalBits = 0, totalSamples = 0;
    int nFrames = 0, totalFramesI = 0;
    DT averagePhase = 0, percentComplete = 0;
    char qualityString[20];
    (void)qualityString;
    while (!feof(in)) {
        nSamples = is_raw ? GetComplexRawChunk(in, header, waveData, waveDataTime, chunk)
                          : GetComplexWaveChunk(in, header, waveData, waveDataTime, chunk);
        if (i == 0 && normFactor == 0) {
            normFactor = StaticGain(waveData, nSamples, 1.0);
            printf("Normalization Factor: %f\n", normFactor);
        }
        i += nSamples;
#ifdef ARGOS_FLOAT
        {   /* the twin stamps its samples itself (ARGOSdemodPortAudio/main.c:285-286): Time += (1/Fs), all float */
            static DT Time = 0;
            for (unsigned long q = 0; q < nSamples; q++) {
                Time += (1 / Fs);
                waveDataTime[q] = Time;
            }
        }
#endif
        dput(diq, waveData, sizeof(DT complex), nSamples);
        dput(dtime, waveDataTime, sizeof(DT), nSamples);
        t_dsp = now_s();
#ifdef ARGOS
        /* ARGOSdemod/main.c:265-284 */
        averagePhase = CarrierTrackPLL(waveData, dataStreamReal, lockSignalStream, nSamples, Fs, (550.0), (0.1),
                        (3.1831) * (2.0 * M_PI / Fs), (16) * (2.0 * M_PI / Fs), (16) * (2.0 * M_PI / Fs));
        dput(dpll, dataStreamReal, sizeof(DT), nSamples);
        dput(dlock, lockSignalStream, sizeof(DT), nSamples);
        LowPassFilter(dataStreamReal, nSamples, filterCoeffs, 50);
        dput(dfir, dataStreamReal, sizeof(DT), nSamples);
        NormalizingAGC(dataStreamReal, nSamples, normFactor, (79.5775) * (2.0 * M_PI / Fs), (159.1549) * (2.0 * M_PI / Fs));
        dput(dagcraw, dataStreamReal, sizeof(DT), nSamples);    /* what -r writes to output.raw (ARGOSdemod/main.c:273-274) */
        Squelch(dataStreamReal, lockSignalStream, nSamples, (0.15));
        dput(dagc, dataStreamReal, sizeof(DT), nSamples);
        if (use_mm)
            nSymbols = MMClockRecovery(dataStreamReal, waveDataTime, nSamples, dataStreamSymbols, Fs, (400 * 2.0), mmRange, mmKp);
        else
            nSymbols = GardenerClockRecovery(dataStreamReal, waveDataTime, nSamples, dataStreamSymbols, Fs, (400 * 2.0), (0.1), (3.0));
        dput(dsym, dataStreamSymbols, sizeof(DT), nSymbols);
        dput(dsymt, waveDataTime, sizeof(DT), nSymbols);
        nBits = ManchesterDecode(dataStreamSymbols, waveDataTime, nSymbols, dataStreamBits, (0.5));
        dput(dbits, dataStreamBits, 1, nBits);
        dput(dbitt, waveDataTime, sizeof(DT), nBits);
        nFrames = FindSyncWords(dataStreamBits, waveDataTime, nBits, "0001011110000", 13, out);
        totalFrames += nFrames;
#else
        /* POESTIPdemod/main.c:413-454 */
        if (live) {                                             /* POESTIPdemodPortAudio/main.c:41-57,367,370 */
            averagePhase = CarrierTrackPLL(waveData, dataStreamReal, lockSignalStream, nSamples, Fs, (4500.0), (0.10),
                            0.3979 * (2.0 * M_PI / Fs), 198.9437 * (2.0 * M_PI / Fs), 10.3451 * (2.0 * M_PI / Fs));
            dput(dlock, lockSignalStream, sizeof(DT), nSamples);
            Squelch(dataStreamReal, lockSignalStream, nSamples, (0.05));
        } else if (haveK)
        averagePhase = CarrierTrackPLL(waveData, dataStreamReal, NULL, nSamples, Fs, loopK[0], loopK[1], loopK[2], loopK[3], loopK[4]);
        else
        averagePhase = CarrierTrackPLL(waveData, dataStreamReal, NULL, nSamples, Fs, (4500.0), (0.08),
                        0.3979 * (2.0 * M_PI / Fs), 127.3240 * (2.0 * M_PI / Fs), 10.3451 * (2.0 * M_PI / Fs));
        dput(dpll, dataStreamReal, sizeof(DT), nSamples);
        LowPassFilterInterp(waveDataTime, dataStreamReal, dataStreamLPF, dataStreamLPFTime, nSamples, filterCoeffs, N, interp);
        dput(dfir, dataStreamLPF, sizeof(DT), nSamples * interp);
        if (haveK) NormalizingAGC(dataStreamLPF, nSamples * interp, normFactor, loopK[5], loopK[6]);
        else
        NormalizingAGC(dataStreamLPF, nSamples * interp, normFactor, (79.5775) * (2.0 * M_PI / (Fs * interp)),
                       (159.1549) * (2.0 * M_PI / (Fs * interp)));
        dput(dagc, dataStreamLPF, sizeof(DT), nSamples * interp);
        if (use_mm)
            nSymbols = MMClockRecovery(dataStreamLPF, dataStreamLPFTime, nSamples * interp, dataStreamSymbols,
                                       Fs * interp, (8320 * 2 + 0.3), mmRange, mmKp);
        else if (haveK)
            nSymbols = GardenerClockRecovery(dataStreamLPF, dataStreamLPFTime, nSamples * interp, dataStreamSymbols,
                                             Fs * interp, loopK[7], loopK[8], loopK[9]);
        else
            nSymbols = GardenerClockRecovery(dataStreamLPF, dataStreamLPFTime, nSamples * interp, dataStreamSymbols,
                                             Fs * interp, (8320 * 2 + 0.3), (0.1), (3.0));
        dput(dsym, dataStreamSymbols, sizeof(DT), nSymbols);
        dput(dsymt, dataStreamLPFTime, sizeof(DT), nSymbols);
        nBits = ManchesterDecode(dataStreamSymbols, dataStreamLPFTime, nSymbols, dataStreamBits, haveK ? loopK[10] : live ? (0.75) : 1.0);   /* twin: main.c:65,393 */
        dput(dbits, dataStreamBits, 1, nBits);
        dput(dbitt, dataStreamLPFTime, sizeof(DT), nBits);
        nFrames = ByteSyncOnSyncword(dataStreamBits, dataStreamLPFTime, nBits, "1110110111100010000", 19, out);
        totalFrames += nFrames;
#endif
        dsp_s += now_s() - t_dsp;
        if (dcnt) fprintf(dcnt, "%lu %lu %lu\n", nSamples, nSymbols, nBits);
        dput(davg, &averagePhase, sizeof(DT), 1);
        totalBits += nBits;
        totalFramesI += nFrames;
        totalSymbols += nSymbols;
        totalSamples += nSamples;
        if (dprog && ((((DT)(i) / num_samples) * 100.0 - percentComplete > 0.15) || feof(in))) {
            percentComplete = ((DT)(i) / num_samples) * 100.0;
            fprintf(dprog, "\r");
#ifdef ARGOS
            fprintf(dprog, "%0.1f%% %0.3f Ks : %0.1f Sec: %ld Sym : %ld Bits : %d Packets", ((DT)(i) / num_samples) * 100.0,
                    (totalSamples) / 1000.0, waveDataTime[0], totalSymbols, totalBits, totalFramesI);
#else
            fprintf(dprog, "%f\t", fabs((M_PI / 2.0) - averagePhase));
            averagePhase = 10.0 * log10f(powf(fabs((M_PI / 2.0) - averagePhase), 2));   /* <tgmath.h> there: log10 of a float */
            if (averagePhase > -4.3) snprintf(qualityString, 20, "%s%02.1fQ%s", "\x1b[32m", averagePhase, "\x1b[0m");
            else if (averagePhase > -5) snprintf(qualityString, 20, "%s%02.1fQ%s", "\x1b[33m", averagePhase, "\x1b[0m");
            else if (averagePhase > -6) snprintf(qualityString, 20, "%s%02.1fQ%s", "\x1b[33m", averagePhase, "\x1b[0m");
            else snprintf(qualityString, 20, "%s%02.1fQ%s", "\x1b[31m", averagePhase, "\x1b[0m");
            fprintf(dprog, "%0.1f%% %0.3f Ks : %0.1f Sec: %ld Sym : %ld Bits : %d Frames : %s   ", ((DT)(i) / num_samples) * 100.0,
                    (totalSamples) / 1000.0, waveDataTime[0], totalSymbols, totalBits, totalFramesI, qualityString);
#endif
        }
    }
    fclose(in);
    fclose(out);
    if (totalFrames == 0) remove(argv[optind + 1]);             /* main.c:508-512 */
    fprintf(stderr, "samples %lu frames %lu norm %.9g dsp_seconds %.6f\n", i, totalFrames, (double)normFactor, dsp_s);
    return 0;
}
