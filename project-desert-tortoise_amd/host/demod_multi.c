/*
 * demod_multi.c -- demodMulti: several independent captures over the GPUs of one node, frames gathered on one GPU with RCCL.
 *
 * The reference demodulates one capture per process (POESTIPdemod/main.c:143-531).  This host program is the multi-GPU
 * front of the port: every GPU has TWO worker threads ("lanes", -l) with a context each (include/pdt.h; no collective on the
 * data path); a lane takes the NEXT capture from a shared queue whenever it has finished one -- a GPU is never idle while
 * captures are waiting.  Why two: a capture is PCIe time first (3.6 GB: ~67 ms) and GPU time second (~18 ms, most of it
 * behind the ingest since round 5, ~8 ms exposed); the library lets one ingest per GPU use the link at a time, so while one
 * lane's capture is in its chain the other lane's is already arriving: a GPU's queue moves at one capture per ingest time
 * (round 4, one context per GPU: ingest + chain, 67 + 18).  When the queue is empty the decoded frame records of every GPU's
 * captures are gathered on the first GPU by libpdtgather (include/pdt_gather.h: ONE gatherer = one set of RCCL communicators
 * for the whole run; all-gather of the counts + padded all-gather of the records over xGMI), and this process writes one
 * output file per capture -- the text POESTIPdemod/ByteSync.c:62-69,96-101 / ARGOSdemod/ByteSync.c:62-70,99-103 print -- next
 * to the input: <capture>.frames.txt.
 *
 * usage: demodMulti [-a] [-c chunk] [-g ngpus] [-l lanes] [-R passes] [-J] capture1.wav capture2.wav ...   (-a: ARGOS chain; default
 *        POES; -l: contexts per GPU, 1 or 2, default 2; -R: the whole queue this many times -- measurements: contexts, buffers and
 *        communicators are warm from the second pass on --; -J: one machine-readable line with every pass's clock (first open ->
 *        last output file closed, the gather inside) and every capture's ingest / GPU time: what bench.py --gpus N reports as the
 *        end-to-end figure of BASELINE's metric)
 *
 * Host budget at N GPUs: 2 N worker threads + per ingesting context up to min(8, cores / 2 / contexts) reader
 * threads (csrc/pdt_api.hip: ingest_capture) and 2 x 8 MiB of pinned staging per reader; frame records 136 B each (4.9 MB per
 * capture-hour) on the host until they are written.
 */
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <fcntl.h>

#include "pdt.h"
#include "pdt_gather.h"

typedef struct capture {
    const char *path;
    int rc, device, order;          /* order: the how-manieth capture of its GPU it was */
    uint64_t nsamples, nfr;
    pdt_frame *frames;              /* host copy of the records (the context goes on to the next capture) */
    pdt_stats st;
    double seconds;
} capture;

typedef struct worker {
    int device, mode, started;
    unsigned long chunk;
    pthread_t th;
    pdt_ctx *ctx;
    uint32_t ctx_rate;
} worker;

typedef struct gpu_tally {          /* what a GPU's lanes have done between them */
    atomic_int done;                /* captures this GPU demodulated */
    atomic_ullong nfr;              /* frames of all of them */
} gpu_tally;

static capture *g_cap;
static int g_ncap;
static atomic_int g_next;
static gpu_tally *g_gpu;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void run_capture(worker *w, capture *c)
{
    const double t0 = now_s();
    c->rc = PDT_ERR_FORMAT;
    c->device = w->device;
    c->order = -1;
    const int fd = open(c->path, O_RDONLY);
    if (fd < 0) return;
    uint8_t hdr[44];
    struct stat sb;
    if (pread(fd, hdr, 44, 0) != 44 || fstat(fd, &sb) != 0) { close(fd); return; }
    uint32_t rate = 0, channels = 0, bits = 0, format = 0, data_bytes = 0;
    pdt_wav_parse_header(hdr, &rate, &channels, &bits, &format, &data_bytes);
    if (channels != 2 || format != 1 || bits != 16) { close(fd); return; }           /* ReadWavHeader's canonical PCM (wave.c:303-378) */
    if (w->ctx && w->ctx_rate != rate) {                             /* a context is bound to a sample rate (taps, loop gains) */
        pdt_close(w->ctx);
        w->ctx = NULL;
    }
    c->rc = PDT_OK;
    if (!w->ctx) {
        pdt_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.mode = w->mode;
        cfg.sample_rate = rate;
        cfg.chunk = w->chunk;
        cfg.device = w->device;
        c->rc = pdt_open(&cfg, &w->ctx);
        w->ctx_rate = rate;
        if (c->rc == PDT_OK) pdt_keep_pll(w->ctx, 0);                /* nothing here reads the PLL output stream */
        else w->ctx = NULL;
    }
    if (c->rc == PDT_OK) {
        c->nsamples = (uint64_t)(sb.st_size - 44) / 4;               /* to the end of the file (main.c:373) */
        c->rc = pdt_demod_fd(w->ctx, fd, 44, c->nsamples, PDT_FMT_PCM16);
    }
    close(fd);
    if (c->rc == PDT_OK) {
        c->nfr = pdt_num_frames(w->ctx);
        c->frames = (pdt_frame *)malloc((size_t)(c->nfr ? c->nfr : 1) * sizeof(pdt_frame));
        if (!c->frames) c->rc = PDT_ERR_NOMEM;
        else if (c->nfr) pdt_frames(w->ctx, c->frames, c->nfr);
        pdt_get_stats(w->ctx, &c->st);
    }
    if (c->rc == PDT_OK) {
        c->order = atomic_fetch_add(&g_gpu[w->device].done, 1);      /* its place among its GPU's captures (both lanes count) */
        atomic_fetch_add(&g_gpu[w->device].nfr, (unsigned long long)c->nfr);
    }
    c->seconds = now_s() - t0;
}

static void *run_worker(void *arg)
{
    worker *w = (worker *)arg;
    for (;;) {
        const int k = atomic_fetch_add(&g_next, 1);                  /* the next capture nobody has taken yet */
        if (k >= g_ncap) break;
        run_capture(w, &g_cap[k]);
    }
    return NULL;
}

/* one capture's output file: <capture>.frames.txt from its records (no frame, no file: main.c:508-512) */
typedef struct write_job { capture *cp; const pdt_frame *src; int wrote; } write_job;
typedef struct write_queue { write_job *jobs; int n; atomic_int next; } write_queue;
static void *run_writer(void *arg)
{
    write_queue *q = (write_queue *)arg;
    for (;;) {
        const int j = atomic_fetch_add(&q->next, 1);
        if (j >= q->n) break;
        write_job *w = &q->jobs[j];
        char name[1200];
        snprintf(name, sizeof name, "%s.frames.txt", w->cp->path);
        w->wrote = 1;
        if (w->cp->nfr) {
            const int fd = open(name, O_RDWR | O_CREAT | O_TRUNC, 0644);
            w->wrote = fd >= 0 && pdt_write_records(w->src, w->cp->nfr, fd, NULL) == PDT_OK;
            if (fd >= 0 && close(fd) != 0) w->wrote = 0;
        } else {
            remove(name);
        }
    }
    return NULL;
}

/* what one pass over the queue took (bench.py reads the -J line) */
typedef struct rep_times {
    double wall_s, demod_s, gather_ms, write_ms;
    int failed;
} rep_times;

static worker *g_wk;
static int g_nw, g_ngpu, g_mode;
static pdt_gatherer *g_gatherer;             /* ONE set of RCCL communicators for the whole run, whatever the number of passes */
static int g_gatherer_ranks;

/* One pass: every capture through the queue, the records gathered on the first GPU, one output file per capture.  The clock runs
 * from the moment the first worker may open its first capture until the last output file has been closed.                     */
static void run_queue(int n, rep_times *T, int quiet)
{
    const int ngpu = g_ngpu, nw = g_nw, mode = g_mode;
    worker *wk = g_wk;
    for (int k = 0; k < n; k++) {
        free(g_cap[k].frames);
        g_cap[k].frames = NULL;
        g_cap[k].nfr = 0;
        g_cap[k].rc = PDT_ERR_STATE;                                 /* (never taken: no worker could be started) */
        g_cap[k].device = -1;
    }
    for (int d = 0; d < ngpu; d++) { atomic_store(&g_gpu[d].done, 0); atomic_store(&g_gpu[d].nfr, 0); }
    atomic_store(&g_next, 0);
    const double t_all = now_s();
    for (int i = 0; i < nw; i++)                                     /* lane 0 of every GPU first: the first captures go one per GPU */
        wk[i].started = pthread_create(&wk[i].th, NULL, run_worker, &wk[i]) == 0;
    for (int i = 0; i < nw; i++)
        if (wk[i].started) pthread_join(wk[i].th, NULL);
    const double t_demod = now_s() - t_all;

    /* ---- gather: rank i = GPU i's records, its captures in the order it took them; one gatherer for the run */
    int failed = 0, ranks = 0, failed_alloc = 0;
    int *devs = (int *)calloc((size_t)ngpu, sizeof(int));
    pdt_frame **rec = (pdt_frame **)calloc((size_t)ngpu, sizeof(pdt_frame *));
    uint64_t *cnt = (uint64_t *)calloc((size_t)ngpu, sizeof(uint64_t)), *got = (uint64_t *)calloc((size_t)ngpu, sizeof(uint64_t));
    int *rank_of = (int *)calloc((size_t)ngpu, sizeof(int));
    for (int d = 0; d < ngpu; d++) {
        rank_of[d] = -1;
        const int d_done = atomic_load(&g_gpu[d].done);
        const uint64_t d_nfr = (uint64_t)atomic_load(&g_gpu[d].nfr);
        if (!d_done) continue;                                       /* a GPU without a decoded capture takes no part */
        rank_of[d] = ranks;
        devs[ranks] = d;
        cnt[ranks] = d_nfr;
        rec[ranks] = (pdt_frame *)malloc((size_t)(d_nfr ? d_nfr : 1) * sizeof(pdt_frame));
        uint64_t at = 0;
        for (int o = 0; o < d_done && rec[ranks]; o++)               /* in the order the GPU finished them */
            for (int k = 0; k < n; k++)
                if (g_cap[k].rc == PDT_OK && g_cap[k].device == d && g_cap[k].order == o) {
                    memcpy(rec[ranks] + at, g_cap[k].frames, (size_t)g_cap[k].nfr * sizeof(pdt_frame));
                    at += g_cap[k].nfr;
                }
        if (!rec[ranks]) failed_alloc = 1;
        ranks++;
    }
    pdt_frame *all = NULL;
    const double t_g0 = now_s();
    int grc = failed_alloc ? PDT_ERR_NOMEM : PDT_OK;               /* no staging copy: no gather (every capture keeps its own frames) */
    if (grc != PDT_OK) printf("gather skipped (%s): every capture's frames are taken from its own GPU's copy\n", pdt_strerror(grc));
    if (ranks && grc == PDT_OK) {
        if (g_gatherer && g_gatherer_ranks != ranks) {               /* (another set of GPUs did work this time) */
            pdt_gatherer_close(g_gatherer);
            g_gatherer = NULL;
        }
        if (!g_gatherer) {
            grc = pdt_gatherer_open(devs, ranks, &g_gatherer);
            g_gatherer_ranks = ranks;
        }
        if (grc == PDT_OK) grc = pdt_gatherer_gather(g_gatherer, (const pdt_frame *const *)rec, cnt, 0, &all, got);   /* RCCL: counts, then padded records */
        if (grc == PDT_OK)
            for (int r = 0; r < ranks; r++)
                if (got[r] != cnt[r]) grc = PDT_ERR_STATE;
        if (grc != PDT_OK) {
            printf("gather failed (%s): every capture's frames are taken from its own GPU's copy\n", pdt_strerror(grc));
            pdt_gatherer_close(g_gatherer);
            g_gatherer = NULL;
        }
    }
    const double t_g1 = now_s();
    /* ---- one output file per capture, from the gathered array: the files are written side by side (eight hour-long captures:
     * 8 x 5 ms one after the other would be a third of the pass behind the last GPU) */
    uint64_t *rank_at = (uint64_t *)calloc((size_t)(ranks + 1), sizeof(uint64_t));
    for (int r = 0; r < ranks; r++) rank_at[r + 1] = rank_at[r] + cnt[r];
    uint64_t samples_all = 0;
    double ingest_all = 0;
    write_job *jobs = (write_job *)calloc((size_t)(n ? n : 1), sizeof(write_job));
    int njobs = 0;
    for (int d = 0; d < ngpu && jobs; d++) {
        uint64_t at = rank_of[d] >= 0 ? rank_at[rank_of[d]] : 0;
        for (int o = 0; o < atomic_load(&g_gpu[d].done); o++)
            for (int k = 0; k < n; k++) {
                capture *cp = &g_cap[k];
                if (cp->rc != PDT_OK || cp->device != d || cp->order != o) continue;
                jobs[njobs].cp = cp;
                jobs[njobs].src = (grc == PDT_OK && all) ? all + at : cp->frames;
                jobs[njobs].wrote = 0;
                njobs++;
                at += cp->nfr;
            }
    }
    {
        write_queue q;
        q.jobs = jobs;
        q.n = njobs;
        atomic_store(&q.next, 0);
        pthread_t th[16];
        int nth = njobs < 16 ? njobs : 16, started = 0;
        for (int i = 1; i < nth; i++)                                   /* (this thread is the first writer) */
            if (pthread_create(&th[started], NULL, run_writer, &q) == 0) started++;
        run_writer(&q);
        for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
    }
    for (int j = 0; j < njobs; j++) {
        capture *cp = jobs[j].cp;
        if (!jobs[j].wrote) { printf("%s.frames.txt: could not be written\n", cp->path); failed++; }
        if (!quiet)
            printf("GPU %d: %s: %0.3f Ks : %llu Sym : %llu Bits : %llu %s  (%.1f ms on the GPU, ingest %.1f ms, %.3f s in all)\n", cp->device,
                   cp->path, cp->st.samples / 1000.0, (unsigned long long)cp->st.symbols, (unsigned long long)cp->st.bits,
                   (unsigned long long)cp->nfr, mode == PDT_MODE_ARGOS ? "Packets" : "Frames", cp->st.gpu_ms, cp->st.ingest_ms, cp->seconds);
        if (jobs[j].wrote) samples_all += cp->st.samples;
        ingest_all += cp->st.ingest_ms;
    }
    free(jobs);
    const double t_w1 = now_s();
    for (int k = 0; k < n; k++)
        if (g_cap[k].rc != PDT_OK) {
            printf("%s: %s\n", g_cap[k].path, pdt_strerror(g_cap[k].rc));
            failed++;
        }
    const double dt = now_s() - t_all;
    printf("%d capture(s), %.3f Msamples in %.3f s (%.3f s until the last GPU was done): %.1f Msamples/s\n", n - failed, samples_all / 1e6,
           dt, t_demod, samples_all / dt / 1e6);
    /* (a GPU's link carries one ingest at a time: with two contexts per GPU the queue should move at about that pace) */
    printf("queue: sum of the captures' ingest times %.1f ms on %d GPU(s), %.1f ms until the last GPU was done\n", ingest_all, ngpu, t_demod * 1e3);
    T->wall_s = dt;
    T->demod_s = t_demod;
    T->gather_ms = (t_g1 - t_g0) * 1e3;
    T->write_ms = (t_w1 - t_g1) * 1e3;
    T->failed = failed;
    for (int r = 0; r < ranks; r++) free(rec[r]);
    free(all); free(rank_at); free(rank_of); free(got); free(cnt); free(rec); free(devs);
}

int main(int argc, char **argv)
{
    int mode = PDT_MODE_POES, c, ngpu = pdt_device_count(), lanes = 2, reps = 1, json = 0;
    unsigned long chunk = 0;
    const char *usage = "usage: %s [-a] [-c chunk] [-g ngpus] [-l 1|2] [-R passes] [-J] capture.wav ...\n";
    while ((c = getopt(argc, argv, "ac:g:l:R:J")) != -1) {
        if (c == 'a') mode = PDT_MODE_ARGOS;
        else if (c == 'c') chunk = strtoul(optarg, NULL, 10);
        else if (c == 'g') ngpu = atoi(optarg);
        else if (c == 'l') {
            lanes = atoi(optarg);
            if (lanes != 1 && lanes != 2) { fprintf(stderr, "-l: contexts per GPU, 1 or 2\n"); fprintf(stderr, usage, argv[0]); return 2; }
        }
        else if (c == 'R') reps = atoi(optarg) > 0 ? atoi(optarg) : 1;   /* the whole queue this many times (measurements: contexts, buffers and communicators are warm from the second pass on) */
        else if (c == 'J') json = 1;                                  /* one machine-readable line at the end (bench.py) */
        else return 2;
    }
    const int n = argc - optind;
    if (n <= 0) { fprintf(stderr, usage, argv[0]); return 2; }
    if (ngpu <= 0) { printf("GPU demodulator unavailable: %s\n", pdt_strerror(PDT_ERR_NOGPU)); return 1; }
    if (ngpu > pdt_device_count()) ngpu = pdt_device_count();
    if (ngpu > n) ngpu = n;
    if (lanes * ngpu > n && lanes > 1) {                             /* fewer captures than lanes: one context per GPU will do */
        printf("%d capture(s) for %d GPU(s): one context per GPU\n", n, ngpu);
        lanes = 1;
    }
    printf("Project Desert Tortoise: %d capture(s) on %d MI355X GPU(s), %d context(s) per GPU, every context takes the next capture when it is free\n",
           n, ngpu, lanes);
    g_cap = (capture *)calloc((size_t)n, sizeof(capture));
    g_ncap = n;
    const int nw = ngpu * lanes;
    worker *wk = (worker *)calloc((size_t)nw, sizeof(worker));
    g_gpu = (gpu_tally *)calloc((size_t)ngpu, sizeof(gpu_tally));
    rep_times *RT = (rep_times *)calloc((size_t)reps, sizeof(rep_times));
    if (!g_cap || !wk || !g_gpu || !RT) return 1;
    for (int k = 0; k < n; k++) g_cap[k].path = argv[optind + k];
    for (int i = 0; i < nw; i++) {
        wk[i].device = i % ngpu;
        wk[i].mode = mode;
        wk[i].chunk = chunk;
    }
    g_wk = wk; g_nw = nw; g_ngpu = ngpu; g_mode = mode;
    int failed = 0;
    for (int r = 0; r < reps; r++) {
        if (reps > 1) printf("pass %d of %d\n", r + 1, reps);
        run_queue(n, &RT[r], 0);
        failed = RT[r].failed;
    }
    if (json) {
        /* one line: every pass's clock, and of the LAST pass every capture's account (bench.py: e2e at N GPUs) */
        printf("{\"demodMulti\": {\"gpus\": %d, \"lanes\": %d, \"captures\": %d, \"passes\": [", ngpu, lanes, n);
        for (int r = 0; r < reps; r++)
            printf("%s{\"wall_s\": %.6f, \"until_last_gpu_s\": %.6f, \"gather_ms\": %.3f, \"write_ms\": %.3f, \"failed\": %d}", r ? ", " : "",
                   RT[r].wall_s, RT[r].demod_s, RT[r].gather_ms, RT[r].write_ms, RT[r].failed);
        printf("], \"last_pass\": [");
        int first = 1;
        for (int k = 0; k < n; k++) {
            const capture *cp = &g_cap[k];
            if (cp->rc != PDT_OK) continue;
            printf("%s{\"capture\": %d, \"gpu\": %d, \"samples\": %llu, \"bytes\": %llu, \"frames\": %llu, \"ingest_ms\": %.3f, \"gpu_ms\": %.3f, "
                   "\"seconds\": %.6f, \"segments\": %u, \"windowed\": %u, \"direct\": %u, \"numa_node\": %d}", first ? "" : ", ", k, cp->device,
                   (unsigned long long)cp->st.samples, (unsigned long long)cp->st.samples * 4ull, (unsigned long long)cp->nfr, cp->st.ingest_ms,
                   cp->st.gpu_ms, cp->seconds, cp->st.segments, cp->st.windowed, cp->st.ingest_direct, cp->st.ingest_numa_node);
            first = 0;
        }
        printf("]}}\n");
    }
    if (g_gatherer) pdt_gatherer_close(g_gatherer);
    for (int i = 0; i < nw; i++)
        if (wk[i].ctx) pdt_close(wk[i].ctx);
    for (int k = 0; k < n; k++) free(g_cap[k].frames);
    free(RT); free(wk); free(g_gpu); free(g_cap);
    return failed ? 1 : 0;
}
