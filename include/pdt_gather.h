/*
 * include/pdt_gather.h -- C ABI of libpdtgather.so: decoded frame records of several captures, demodulated on one GPU each,
 * brought together on one GPU / one host buffer with RCCL over xGMI.
 *
 * The reference has nothing of the kind (one process, one capture: POESTIPdemod/main.c:143-531); this is the "independent
 * captures shard one per GPU, trivial gather of decoded frames" part of the port's brief.  The data path itself has no
 * collective: every capture is demodulated by its own context (include/pdt.h) on its own GPU.  What is gathered is small
 * (136 bytes per minor frame, ~10 frames per second of signal) and ragged, so: one all-gather of the per-rank counts,
 * then one all-gather of the records padded to the largest count; the root's copy goes to the host.
 */
#ifndef PDT_GATHER_H
#define PDT_GATHER_H
#include "pdt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Gather the frames of the last demodulation of each of the n contexts (they must live on n DIFFERENT GPUs of this process)
 * on context `root`'s GPU and from there into one host array: *out (malloc'ed here, the caller frees it) holds the records of
 * context 0, then context 1, ...; counts[i] = frames of context i.  Returns PDT_OK or a PDT_ERR_* code.                      */
int pdt_gather_frames(pdt_ctx *const *ctxs, int n, int root, pdt_frame **out, uint64_t *counts);
/* (pdt_gather_frames keeps one gatherer per set of GPUs for the rest of the process; pdt_gather_shutdown frees them.) */
void pdt_gather_shutdown(void);

/* The same exchange with its state in the caller's hands (round 4): the communicators (one ncclCommInitAll), a stream and the
 * device / pinned buffers per GPU are set up ONCE by pdt_gatherer_open and serve any number of gathers -- demodMulti opens
 * one for its GPUs and gathers the records of all its captures at the end.  records[i] / counts_in[i]: rank i's records in host
 * memory (frame records carry their time stamps, which the host evaluates -- csrc/pdt_timeaxis.h -- so the host is where a
 * finished record first exists; they are staged through pinned memory of the gatherer, 136 bytes a frame).  Results as
 * pdt_gather_frames.  A gatherer serialises concurrent gathers on itself.                                                 */
typedef struct pdt_gatherer pdt_gatherer;
int  pdt_gatherer_open(const int *devices, int n, pdt_gatherer **out);
int  pdt_gatherer_gather(pdt_gatherer *g, const pdt_frame *const *records, const uint64_t *counts_in, int root, pdt_frame **out,
                         uint64_t *counts);
void pdt_gatherer_close(pdt_gatherer *g);

/* The exchange format, host only (no GPU, no RCCL needed to call them): every rank contributes max(1, largest count) records,
 * its own first and padding behind them.  pdt_gather_plan: nmax and (optional, n + 1 entries) the offset of every rank's
 * records in the unpadded result; pdt_gather_unpad: padded (n x nmax records) -> out (sum of counts records).  bench.py's
 * gather across processes (torch.distributed) uses the same two functions.                                                */
int pdt_gather_plan(const uint64_t *counts, int n, uint64_t *nmax_out, uint64_t *offsets);
int pdt_gather_unpad(const void *padded, const uint64_t *counts, int n, uint64_t nmax, uint64_t record_bytes, void *out);

#ifdef __cplusplus
}
#endif
#endif
