/* pdt_dev.h -- TEST-ONLY entry of libpdt.so: developer switches.
 *
 * The product never reads the environment.  The A/B switches the tests and the tuning sweeps use (older kernel variants kept as
 * fallbacks for geometries the default kernels do not cover, block geometry overrides, thresholds brought down to test sizes)
 * live in a process-wide registry that only this entry fills; a context copies them once, in pdt_open.  The Python binding the
 * tests and bench.py go through mirrors the process's PDT_* environment variables into it before every pdt_open, so a test
 * still says os.environ["PDT_GSPAN"] = "4"; bin/demodPOES, bin/demodARGOS, bin/demodMulti and any other client of
 * include/pdt.h never call it.  The names are listed in tools/README.md. */
#ifndef PDT_DEV_H
#define PDT_DEV_H
#ifdef __cplusplus
extern "C" {
#endif
/* value != NULL: set switch `name`; value == NULL: remove it; name == NULL: clear all.  Contexts opened afterwards see it. */
int pdt_dev_set(const char *name, const char *value);
/* (measurement aid, round 6) The boundary-state tables of the last demodulation when their rows span several chunks: the number
 * of rows, and for the first `max` of them the number of DISTINCT exits of the row's first chunk (0xffffffff: a row that was
 * left untabulated).  0 when the last run had no such rows.  What tools/span_hist.py turns into profiles/r6/gardner_row_exits_*.  */
struct pdt_ctx;
unsigned long long pdt_dev_span_rows(const struct pdt_ctx *ctx, unsigned *n_exits_out, unsigned long long max);
#ifdef __cplusplus
}
#endif
#endif
