/*
 * oracle_tip.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's downstream frame validation, which it keeps in MATLAB:
 *   standalone_matlab/Functionized/checkParity.m:1-92   five even-parity checks per TIP minor frame
 *   standalone_matlab/Functionized/daytimeDecode.m:1-40 minor-frame counter, spacecraft id, day / ms of day, T0
 * MATLAB is 1-indexed: minorFrames(frame, w) is byte w-1 of the 104-byte frame the C demodulator prints.
 *
 * Pinning: there is no MATLAB/Octave in this image, so the restatement cannot be run against the .m files.
 * It is pinned on data instead (tests/test_oracle_tip.py): the 47 complete frames the reference decodes from
 * its own bundled 5sec_clip.wav (a real NOAA-15 pass) satisfy all 5 x 47 parity equations under exactly this
 * word/bit mapping, carry consecutive minor-frame counters and the spacecraft id 8 = NOAA-15; any other
 * mapping fails about half of them.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* checkParity.m:20-86, written the MATLAB way: count the ones of 17 words bit by bit, compare the count
 * modulo 2 with one bit of word 104 (byte 103). */
static int group_fails(const uint8_t *b, int first_word, int last_word, int shift)
{
    int ones = 0;
    for (int word = first_word; word <= last_word; word++) {        /* 1-based, inclusive */
        const int byte = b[word - 1];
        for (int s = 0; s <= 7; s++) ones += (byte >> s) & 1;
    }
    return ((ones % 2) == ((b[104 - 1] >> shift) & 1)) ? 0 : 1;
}

void orc_tip_check(const orc_frame *frames, size_t n, orc_tip_frame *out, orc_tip_summary *sum)
{
    static const int first[5] = {3, 20, 37, 54, 71}, last[5] = {19, 36, 53, 70, 87}, shift[5] = {5, 4, 3, 2, 1};
    size_t hist_sc[256], hist_day[512];
    memset(hist_sc, 0, sizeof hist_sc);
    memset(hist_day, 0, sizeof hist_day);
    memset(sum, 0, sizeof *sum);
    sum->spacecraft = -1;
    sum->day = -1;
    sum->t0_ms = -1;
    double *t0 = (double *)malloc((n + 1) * sizeof(double));
    size_t n_t0 = 0;
    for (size_t f = 0; f < n; f++) {
        orc_tip_frame *o = &out[f];
        memset(o, 0, sizeof *o);
        const uint8_t *b = frames[f].bytes;
        if (!frames[f].complete || frames[f].nbytes != 104) continue;      /* only whole minor frames are rows of the matrix */
        o->checked = 1;
        for (int g = 0; g < 5; g++) {
            const int bad = group_fails(b, first[g], last[g], shift[g]);
            if (bad) { o->parity |= (uint8_t)(1u << g); sum->bad_chunks++; } else sum->good_chunks++;
        }
        sum->frames_checked++;
        if (o->parity == 0) sum->good_frames++;
        /* daytimeDecode.m:4,16 */
        o->minor_id = (uint16_t)(((b[5 - 1] & 1) << 8) | b[6 - 1]);
        o->spacecraft = b[3 - 1];
        hist_sc[o->spacecraft]++;
        if (o->minor_id == 0) {                                             /* :18-31 */
            o->has_time = 1;
            o->day = (uint16_t)((b[8] << 1) + ((b[9] | 128) >> 7));         /* the reference ORs where an AND was meant */
            hist_day[o->day]++;
            const long ms = ((long)(b[9] & 7) << 24) + ((long)b[10] << 16) + ((long)b[11] << 8) + (long)b[12];
            o->day_ms = (ms < 86400000L) ? (int32_t)ms : -1;
            sum->time_frames++;
            if (o->day_ms >= 0) {
                /* frameTime comes from the text file: "%.5f" */
                const double t = round(frames[f].time * 1e5) / 1e5;
                const double v = (double)o->day_ms - t * 1000.0;
                if (v > 0) t0[n_t0++] = round(v);
            }
        }
    }
    /* MATLAB mode(): the most frequent value, the smallest one on ties */
    size_t best = 0;
    for (int v = 0; v < 256; v++) if (hist_sc[v] > best) { best = hist_sc[v]; sum->spacecraft = v; }
    best = 0;
    for (int v = 0; v < 512; v++) if (hist_day[v] > best) { best = hist_day[v]; sum->day = v; }
    best = 0;
    for (size_t i = 0; i < n_t0; i++) {
        size_t c = 0;
        for (size_t j = 0; j < n_t0; j++) c += (t0[j] == t0[i]) ? 1 : 0;
        if (c > best || (c == best && (int64_t)t0[i] < sum->t0_ms)) { best = c; sum->t0_ms = (int64_t)t0[i]; }
    }
    free(t0);
}
