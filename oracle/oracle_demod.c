/*
 * oracle/oracle_demod.c -- TEST INFRASTRUCTURE ONLY.
 * Command-line front end of the CPU restatement: WAV in, minor-frame / packet
 * text out, optional per-stage dumps (same file set as ref_driver.c -d).
 *   oracle_demod [-a] [-c chunk] [-n gain] [-s rate] [-d prefix] [-t] in.wav out.txt
 *     -a  ARGOS chain (double) instead of POES (float)
 *     -t  print wall time of the demodulation (CPU baseline measurements)
 * Header parsing follows common/wave.c:303-378 (fixed 44-byte header, no chunk walk);
 * every byte after the header is sample data (POESTIPdemod/main.c:373, Q7).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "oracle.h"

static unsigned rd32(const unsigned char *b) { return b[0] | (b[1] << 8) | (b[2] << 16) | ((unsigned)b[3] << 24); }
static unsigned rd16(const unsigned char *b) { return b[0] | (b[1] << 8); }

int main(int argc, char **argv)
{
    int mode = ORC_POES, c, timing = 0;
    unsigned long chunk = 0;
    double norm = 0, srate = 0;
    const char *dump = NULL;
    while ((c = getopt(argc, argv, "ac:n:s:d:t")) != -1) {
        if (c == 'a') mode = ORC_ARGOS;
        else if (c == 'c') chunk = strtoul(optarg, NULL, 10);
        else if (c == 'n') norm = atof(optarg);
        else if (c == 's') srate = atof(optarg);
        else if (c == 'd') dump = optarg;
        else if (c == 't') timing = 1;
        else return 2;
    }
    if (argc - optind < 2) { fprintf(stderr, "usage: %s [-a] [-c chunk] [-n gain] [-s rate] [-d prefix] in.wav out.txt\n", argv[0]); return 2; }
    FILE *f = fopen(argv[optind], "rb");
    if (!f) { perror(argv[optind]); return 1; }
    unsigned char hdr[44];
    if (fread(hdr, 1, 44, f) != 44) { fprintf(stderr, "short header\n"); return 1; }
    unsigned channels = rd16(hdr + 22), rate = rd32(hdr + 24), bits = rd16(hdr + 34), fmt = rd16(hdr + 20);
    if (channels != 2 || fmt != 1 || bits != 16) { fprintf(stderr, "need 16-bit PCM, 2 channels\n"); return 1; }
    if (mode == ORC_POES && srate > 1) rate = (unsigned)srate;     /* POESTIPdemod/main.c:343-344 */
    fseek(f, 0, SEEK_END);
    long sz = ftell(f) - 44;
    fseek(f, 44, SEEK_SET);
    size_t nframes = (size_t)sz / 4;
    int16_t *pcm = (int16_t *)malloc(nframes * 4 + 4);
    if (fread(pcm, 4, nframes, f) != nframes) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);

    orc_pipe *p = orc_open(mode, rate, chunk, norm, dump != NULL);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    orc_run_pcm16(p, pcm, nframes);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    size_t len;
    const char *txt = orc_text(p, &len);
    if (orc_num_frames(p) > 0) {
        FILE *o = fopen(argv[optind + 1], "w");
        fwrite(txt, 1, len, o);
        fclose(o);
    } else {
        remove(argv[optind + 1]);
    }
    if (dump) {
        static const char *ext[ORC_ST_COUNT] = { "iq", "time", "pll", "lock", "fir", "agc", "sym", "symt", "bits", "bitt", "counts64", "taps", "symidx" };
        for (int s = 0; s < ORC_ST_COUNT; s++) {
            size_t n = orc_stage(p, s, NULL, 0);
            char name[1200];
            snprintf(name, sizeof name, "%s.%s", dump, ext[s]);
            void *buf = malloc(n + 1);
            orc_stage(p, s, buf, n);
            FILE *o = fopen(name, "wb");
            fwrite(buf, 1, n, o);
            fclose(o);
            free(buf);
        }
    }
    double dt = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    fprintf(stderr, "samples %zu frames %zu norm %.9g lock@%ld %.2fHz", nframes, orc_num_frames(p), orc_norm_factor(p),
            orc_lock_sample(p), orc_lock_freq_hz(p));
    fprintf(stderr, " dsp_seconds %.6f", dt);
    if (timing) fprintf(stderr, " time %.3fs %.3f Msamples/s", dt, nframes / dt / 1e6);
    fprintf(stderr, "\n");
    orc_close(p);
    free(pcm);
    return 0;
}
