/* tests/fake_pdt.c -- TEST INFRASTRUCTURE: stand-ins for the libpdt / libpdtgather entry points bin/demodMulti calls, so that its
 * scheduling (one worker per GPU, a shared queue of captures, one gather at the end) runs on a machine without a GPU
 * (tests/test_multi_queue.py links host/demod_multi.c against this file instead of the libraries).
 * A "capture" is a 44-byte PCM16 stereo WAV header followed by: u32 milliseconds its chain takes, u32 frames it yields, u32
 * milliseconds its ingest takes (one ingest per GPU at a time, as in the library), then padding.  Frame k of a capture carries bytes derived from the file's size and k.  FAKE_DEVICES = number of GPUs. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <pthread.h>
#include "pdt.h"
#include "pdt_gather.h"

struct pdt_ctx { pdt_config cfg; pdt_frame *fr; uint64_t nfr; pdt_stats st; };
static int g_opens, g_closes, g_gather_opens;
static pthread_mutex_t g_link[64] = { [0 ... 63] = PTHREAD_MUTEX_INITIALIZER };     /* one ingest per GPU at a time */

const char *pdt_strerror(int c) { return c == PDT_OK ? "ok" : c == PDT_ERR_FORMAT ? "unsupported WAV format" : "error"; }
int pdt_device_count(void) { const char *e = getenv("FAKE_DEVICES"); return e ? atoi(e) : 0; }
int pdt_get_device(const pdt_ctx *c) { return c->cfg.device; }
int pdt_keep_pll(pdt_ctx *c, int e) { (void)c; (void)e; return PDT_OK; }
int pdt_open(const pdt_config *cfg, pdt_ctx **out)
{
    pdt_ctx *c = calloc(1, sizeof *c);
    c->cfg = *cfg;
    __atomic_add_fetch(&g_opens, 1, __ATOMIC_SEQ_CST);
    *out = c;
    return PDT_OK;
}
void pdt_close(pdt_ctx *c)
{
    if (!c) return;
    __atomic_add_fetch(&g_closes, 1, __ATOMIC_SEQ_CST);
    free(c->fr);
    free(c);
}
int pdt_wav_parse_header(const uint8_t h[44], uint32_t *rate, uint32_t *ch, uint32_t *bits, uint32_t *fmt, uint32_t *bytes)
{
    *fmt = h[20] | h[21] << 8; *ch = h[22] | h[23] << 8; memcpy(rate, h + 24, 4); *bits = h[34] | h[35] << 8; memcpy(bytes, h + 40, 4);
    return PDT_OK;
}
int pdt_demod_fd(pdt_ctx *c, int fd, uint64_t off, uint64_t nframes, int fmt)
{
    uint32_t w[3] = {0, 0, 0};
    (void)fmt;
    if (pread(fd, w, 12, (off_t)off) != 12) return PDT_ERR_FORMAT;
    pthread_mutex_lock(&g_link[c->cfg.device & 63]);
    usleep(w[2] * 1000u);                                    /* the ingest: the link is this capture's */
    pthread_mutex_unlock(&g_link[c->cfg.device & 63]);
    usleep(w[0] * 1000u);                                    /* the chain */
    free(c->fr);
    c->nfr = w[1];
    c->fr = calloc(c->nfr ? c->nfr : 1, sizeof(pdt_frame));
    for (uint64_t k = 0; k < c->nfr; k++) {
        c->fr[k].time = (double)k * 0.1;
        c->fr[k].bit_index = (int64_t)(k * 832);
        c->fr[k].nbytes = 104;
        c->fr[k].complete = 1;
        for (int b = 0; b < 104; b++) c->fr[k].bytes[b] = (uint8_t)(nframes * 7 + k * 13 + (uint64_t)b);
    }
    memset(&c->st, 0, sizeof c->st);
    c->st.samples = nframes; c->st.frames = c->nfr; c->st.gpu_ms = w[0]; c->st.ingest_ms = w[2]; c->st.segments = 1;
    return PDT_OK;
}
uint64_t pdt_num_frames(const pdt_ctx *c) { return c->nfr; }
uint64_t pdt_frames(const pdt_ctx *c, pdt_frame *out, uint64_t m)
{
    const uint64_t n = m < c->nfr ? m : c->nfr;
    memcpy(out, c->fr, n * sizeof(pdt_frame));
    return n;
}
int pdt_get_stats(const pdt_ctx *c, pdt_stats *o) { *o = c->st; return PDT_OK; }
int pdt_write_records(const pdt_frame *f, uint64_t n, int fd, uint64_t *bytes)
{
    uint64_t tot = 0;
    for (uint64_t k = 0; k < n; k++) {
        char line[64];
        const int w = snprintf(line, sizeof line, "%.5f %02X %02X %02X\n", f[k].time, f[k].bytes[0], f[k].bytes[1], f[k].bytes[103]);
        if (write(fd, line, (size_t)w) != w) return PDT_ERR_IO;
        tot += (uint64_t)w;
    }
    if (bytes) *bytes = tot;
    return PDT_OK;
}
/* the gather: the same exchange format as the library's (plan / unpad over a padded buffer), without the collectives */
struct pdt_gatherer { int n; };
int pdt_gatherer_open(const int *dev, int n, pdt_gatherer **out)
{
    (void)dev;
    g_gather_opens++;
    *out = calloc(1, sizeof **out);
    (*out)->n = n;
    return PDT_OK;
}
int pdt_gatherer_gather(pdt_gatherer *g, const pdt_frame *const *rec, const uint64_t *cin, int root, pdt_frame **out, uint64_t *counts)
{
    uint64_t nmax = 1, tot = 0;
    (void)root;
    for (int i = 0; i < g->n; i++) { counts[i] = cin[i]; if (cin[i] > nmax) nmax = cin[i]; tot += cin[i]; }
    pdt_frame *pad = calloc((size_t)g->n * nmax, sizeof(pdt_frame));
    for (int i = 0; i < g->n; i++) memcpy(pad + (size_t)i * nmax, rec[i], (size_t)cin[i] * sizeof(pdt_frame));
    pdt_frame *res = malloc((tot ? tot : 1) * sizeof(pdt_frame));
    uint64_t at = 0;
    for (int i = 0; i < g->n; i++) { memcpy(res + at, pad + (size_t)i * nmax, (size_t)cin[i] * sizeof(pdt_frame)); at += cin[i]; }
    free(pad);
    *out = res;
    return PDT_OK;
}
void pdt_gatherer_close(pdt_gatherer *g) { free(g); }
static void report(void) { fprintf(stderr, "fake: %d context(s) opened, %d closed, %d gatherer(s)\n", g_opens, g_closes, g_gather_opens); }
__attribute__((constructor)) static void init(void) { atexit(report); }
