/*
 * demod_multi.c -- demodMulti: several independent captures, one per GPU, frames gathered on one GPU with RCCL.
 *
 * The reference demodulates one capture per process (POESTIPdemod/main.c:143-531).  This host program is the multi-GPU
 * front of the port: capture i goes to GPU i (mod the number of GPUs), every capture is demodulated by its own context
 * (include/pdt.h, no collective on the data path) in its own thread; the decoded frame records of each wave of captures are
 * then gathered on the first GPU of the wave by libpdtgather (include/pdt_gather.h: all-gather of the counts + padded
 * all-gather of the records over xGMI), and this process writes one output file per capture -- the text
 * POESTIPdemod/ByteSync.c:62-69,96-101 / ARGOSdemod/ByteSync.c:62-70,99-103 print -- next to the input: <capture>.frames.txt.
 *
 * usage: demodMulti [-a] [-c chunk] [-g ngpus] capture1.wav capture2.wav ...      (-a: ARGOS chain; default POES)
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <fcntl.h>

#include "pdt.h"
#include "pdt_gather.h"

typedef struct job {
    const char *path;
    int device, mode, rc, started;
    unsigned long chunk;
    pdt_ctx *ctx;
    uint64_t nframes;
    double seconds;
} job;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *run_job(void *arg)
{
    job *j = (job *)arg;
    const double t0 = now_s();
    j->rc = PDT_ERR_FORMAT;
    const int fd = open(j->path, O_RDONLY);
    if (fd < 0) return NULL;
    uint8_t hdr[44];
    struct stat sb;
    if (pread(fd, hdr, 44, 0) != 44 || fstat(fd, &sb) != 0) { close(fd); return NULL; }
    uint32_t rate = 0, channels = 0, bits = 0, format = 0, data_bytes = 0;
    pdt_wav_parse_header(hdr, &rate, &channels, &bits, &format, &data_bytes);
    if (channels != 2 || format != 1 || bits != 16) { close(fd); return NULL; }      /* ReadWavHeader's canonical PCM (wave.c:303-378) */
    pdt_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.mode = j->mode;
    cfg.sample_rate = rate;
    cfg.chunk = j->chunk;
    cfg.device = j->device;
    j->rc = pdt_open(&cfg, &j->ctx);
    if (j->rc == PDT_OK) pdt_keep_pll(j->ctx, 0);                   /* nothing here reads the PLL output stream */
    if (j->rc == PDT_OK) {
        j->nframes = (uint64_t)(sb.st_size - 44) / 4;                                  /* to the end of the file (main.c:373) */
        j->rc = pdt_demod_fd(j->ctx, fd, 44, j->nframes, PDT_FMT_PCM16);
    }
    close(fd);
    j->seconds = now_s() - t0;
    return NULL;
}

int main(int argc, char **argv)
{
    int mode = PDT_MODE_POES, c, ngpu = pdt_device_count();
    unsigned long chunk = 0;
    while ((c = getopt(argc, argv, "ac:g:")) != -1) {
        if (c == 'a') mode = PDT_MODE_ARGOS;
        else if (c == 'c') chunk = strtoul(optarg, NULL, 10);
        else if (c == 'g') ngpu = atoi(optarg);
        else return 2;
    }
    const int n = argc - optind;
    if (n <= 0) { fprintf(stderr, "usage: %s [-a] [-c chunk] [-g ngpus] capture.wav ...\n", argv[0]); return 2; }
    if (ngpu <= 0) { printf("GPU demodulator unavailable: %s\n", pdt_strerror(PDT_ERR_NOGPU)); return 1; }
    if (ngpu > pdt_device_count()) ngpu = pdt_device_count();
    printf("Project Desert Tortoise: %d capture(s) on %d MI355X GPU(s), one capture per GPU at a time\n", n, ngpu);
    job *jobs = (job *)calloc((size_t)n, sizeof(job));
    pthread_t *th = (pthread_t *)calloc((size_t)n, sizeof(pthread_t));
    int failed = 0;
    const double t_all = now_s();
    uint64_t samples_all = 0;
    for (int w0 = 0; w0 < n; w0 += ngpu) {                       /* waves of one capture per GPU */
        const int wn = (n - w0 < ngpu) ? n - w0 : ngpu;
        for (int k = 0; k < wn; k++) {
            job *j = &jobs[w0 + k];
            j->path = argv[optind + w0 + k];
            j->device = k;
            j->mode = mode;
            j->chunk = chunk;
            j->started = pthread_create(&th[w0 + k], NULL, run_job, j) == 0;
            if (!j->started) j->rc = PDT_ERR_NOMEM;                /* (no thread: this capture fails, the others go on) */
        }
        for (int k = 0; k < wn; k++)
            if (jobs[w0 + k].started) pthread_join(th[w0 + k], NULL);
        /* the captures that were demodulated: only those take part in the gather; a failed capture costs its own output */
        pdt_ctx **ctxs = (pdt_ctx **)calloc((size_t)wn, sizeof(pdt_ctx *));
        int *who = (int *)calloc((size_t)wn, sizeof(int));
        int good = 0;
        for (int k = 0; k < wn; k++) {
            if (jobs[w0 + k].rc != PDT_OK) {
                printf("%s: %s\n", jobs[w0 + k].path, pdt_strerror(jobs[w0 + k].rc));
                failed++;
            } else {
                who[good] = w0 + k;
                ctxs[good++] = jobs[w0 + k].ctx;
            }
        }
        if (good) {
            pdt_frame *all = NULL;
            uint64_t *counts = (uint64_t *)calloc((size_t)good, sizeof(uint64_t));
            int rc = counts ? pdt_gather_frames(ctxs, good, 0, &all, counts) : PDT_ERR_NOMEM;   /* RCCL: counts, then padded records */
            if (rc != PDT_OK) printf("gather failed (%s): every capture's frames are taken from its own context\n", pdt_strerror(rc));
            uint64_t at = 0;
            for (int k = 0; k < good; k++) {
                const job *j = &jobs[who[k]];
                char name[1200];
                snprintf(name, sizeof name, "%s.frames.txt", j->path);
                const uint64_t nfr = rc == PDT_OK ? counts[k] : pdt_num_frames(j->ctx);
                const uint64_t need = rc == PDT_OK ? pdt_format_records(all + at, nfr, NULL, 0) : pdt_format_frames(j->ctx, NULL, 0);
                char *text = (char *)malloc(need + 1);
                int wrote = text != NULL;
                if (text) {
                    if (rc == PDT_OK) pdt_format_records(all + at, nfr, text, need);
                    else pdt_format_frames(j->ctx, text, need);
                    if (nfr) {
                        FILE *f = fopen(name, "w");
                        wrote = f && fwrite(text, 1, need, f) == need;
                        if (f && fclose(f) != 0) wrote = 0;
                    } else {
                        remove(name);                                              /* no frame, no file (main.c:508-512) */
                    }
                    free(text);
                }
                if (!wrote) { printf("%s: could not be written\n", name); failed++; }
                pdt_stats st;
                pdt_get_stats(j->ctx, &st);
                printf("GPU %d: %s: %0.3f Ks : %llu Sym : %llu Bits : %llu %s  (%.1f ms on the GPU, %.3f s with file read)\n", j->device,
                       j->path, st.samples / 1000.0, (unsigned long long)st.symbols, (unsigned long long)st.bits,
                       (unsigned long long)nfr, mode == PDT_MODE_ARGOS ? "Packets" : "Frames", st.gpu_ms, j->seconds);
                if (wrote) samples_all += st.samples;
                if (rc == PDT_OK) at += counts[k];
            }
            free(all);
            free(counts);
        }
        free(who);
        for (int k = 0; k < wn; k++)
            if (jobs[w0 + k].ctx) pdt_close(jobs[w0 + k].ctx);
        free(ctxs);
    }
    const double dt = now_s() - t_all;
    printf("%d capture(s), %.3f Msamples in %.3f s: %.1f Msamples/s\n", n - failed, samples_all / 1e6, dt, samples_all / dt / 1e6);
    free(jobs);
    free(th);
    return failed ? 1 : 0;
}
