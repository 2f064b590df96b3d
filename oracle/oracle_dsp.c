/*
 * oracle/oracle_dsp.c -- TEST INFRASTRUCTURE ONLY.
 * Instantiates the stage restatements for float (POES) and double (ARGOS) and
 * implements the chunk loop + byte synchronisers on top of them.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* ---- float instantiation */
#define DT float
#define SFX _f32
#define SINCOS(p, s, c) orc_sincosf((p), (s), (c))
#define ARCTAN2(y, x) orc_arctan2_f32((y), (x))
#define FABS(x) fabsf(x)
#define FABSNARROW(x) fabsf((float)(x))
#define SIN(x) sinf(x)
#define HYPOT(x, y) orc_hypotf((x), (y))
#define RINT(x) rintf(x)
#define RINT_MM(x) rintf(x)                 /* MMClockRecovery.c:25,28: rint() through tgmath.h on a float */
#include "oracle_dsp_tmpl.inc"
#undef RINT_MM
#undef DT
#undef SFX
#undef SINCOS
#undef ARCTAN2
#undef FABS
#undef FABSNARROW
#undef SIN
#undef HYPOT
#undef RINT

/* ---- double instantiation */
#define DT double
#define SFX _f64
#define SINCOS(p, s, c) orc_sincos((p), (s), (c))
#define ARCTAN2(y, x) orc_arctan2_f64((y), (x))
#define FABS(x) fabs(x)
#define FABSNARROW(x) fabs(x)
#define SIN(x) sin(x)
#define HYPOT(x, y) orc_hypot((x), (y))
#define RINT(x) rint(x)
#define RINT_MM(x) ((double)rintf((float)(x)))  /* MMClockRecovery.c:55,58: the double build calls rintf (sic) */
#include "oracle_dsp_tmpl.inc"
#undef RINT_MM
#undef DT
#undef SFX
#undef SINCOS
#undef ARCTAN2
#undef FABS
#undef FABSNARROW
#undef SIN
#undef HYPOT
#undef RINT

#include "oracle_pipeline.inc"
