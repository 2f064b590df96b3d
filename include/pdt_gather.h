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

#ifdef __cplusplus
}
#endif
#endif
