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
#ifdef __cplusplus
}
#endif
#endif
