/* tools/lab/acq_merge.c -- LAB: do two runs of the acquisition-mode loop filter (CarrierTrackingPLL.c:165-252 in float, the sweep on or off) that
 * start from different states ever become bitwise equal on a stream that is noise throughout?  (They do where there is a signal: that
 * is what the block-parallel tracking kernels rest on.)  Walkers started every 50 000 samples from the first-call state against a
 * reference run from sample 0; Fs 250 k, sweep on: 48 of 99 merge within 3 M samples, the median does not; sweep off: median 0.94 M.
 * So the noise in front of a pass cannot be walked in blocks with warm-ups: the pre-lock stretch is serial.
 *   gcc -O2 -ffp-contract=off -o acq_merge acq_merge.c -lm && ./acq_merge [Fs] [samples] [carrier amplitude / noise sigma] [sweep 0|1] */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
typedef struct { float ph, fr, sw; } st_t;
static float alpha, beta, maxf, minf;
static inline void step(st_t *s, float th, int open)
{
    const float hi = 6.2831854820251465f, d = 1.7484555314695172e-07f;
    float e = th - s->ph;
    if (fabsf(e) >= 3.14159274101257324f) e = (e - copysignf(hi, e)) + copysignf(d, e);
    float f1 = s->fr + beta * e;
    float p = s->ph + f1;
    p = p + alpha * e;
    float k = truncf(p * 0.15915494309189535f);
    s->ph = fmaf(k, d, fmaf(k, -hi, p));
    float f = f1 > maxf ? maxf : (f1 < minf ? minf : f1);
    if (open) {
        float f2 = f + s->sw;
        float mag = fabsf(s->sw);
        float by = (f2 >= 0) ? mag : -mag;
        s->sw = (f2 >= maxf || f2 <= minf) ? -s->sw : by;
        f = f2;
    }
    s->fr = f;
}
static uint64_t rng = 88172645463325252ull;
static inline uint64_t xs(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
int main(int argc, char **argv)
{
    const double Fs = argc > 1 ? atof(argv[1]) : 250000.0;
    const long N = argc > 2 ? atol(argv[2]) : 6000000;
    const double snr_amp = argc > 3 ? atof(argv[3]) : 0.0;      /* carrier amplitude relative to noise sigma (0 = pure noise) */
    const int open = argc > 4 ? atoi(argv[4]) : 1;
    const double w = 2.0 * M_PI / Fs;
    const float bw = (float)(127.3240 * w), damp = 0.999f;
    alpha = (4 * damp * bw) / (1 + 2 * damp * bw + bw * bw);
    beta = (4 * bw * bw) / (1 + 2 * damp * bw + bw * bw);
    maxf = (float)(2.0 * M_PI * 4500.0 / Fs); minf = -maxf;
    const float sweep0 = (float)(0.2 * w);
    float *th = malloc(sizeof(float) * N);
    double cph = 0.3;
    for (long i = 0; i < N; i++) {
        /* gaussian-ish noise (sum of 4 uniforms) + optional carrier at +1 kHz */
        double a = 0, b = 0;
        for (int k = 0; k < 4; k++) { a += (double)(xs() >> 11) / 9007199254740992.0 - 0.5; b += (double)(xs() >> 11) / 9007199254740992.0 - 0.5; }
        a *= 1.732; b *= 1.732;                     /* sigma 1 */
        cph += 2.0 * M_PI * 1000.0 / Fs;
        a += snr_amp * cos(cph); b += snr_amp * sin(cph);
        th[i] = atan2f((float)b, (float)a);
    }
    st_t *ref = malloc(sizeof(st_t) * N);
    st_t s = { 0.1f, 0.0f, sweep0 };
    for (long i = 0; i < N; i++) { step(&s, th[i], open); ref[i] = s; }
    printf("alpha %.5g beta %.5g maxf %.5g sweep %.3g; ref fr at end %.4f\n", alpha, beta, maxf, sweep0, s.fr);
    /* walkers starting every 'stride' samples from the canonical state */
    const long stride = 50000, limit = 3000000;
    long cnt = 0, merged = 0; long times[512];
    for (long s0 = stride; s0 + limit < N && cnt < 100; s0 += stride) {
        st_t t = { 0.1f, 0.0f, sweep0 };
        long m = -1;
        for (long i = s0; i < s0 + limit; i++) {
            step(&t, th[i], open);
            if (memcmp(&t, &ref[i], sizeof t) == 0) { m = i - s0; break; }
        }
        times[cnt++] = m;
        if (m >= 0) merged++;
    }
    /* sort */
    for (long i = 0; i < cnt; i++) for (long j = i + 1; j < cnt; j++) if ((times[j] >= 0 && times[j] < times[i]) || times[i] < 0) { long x = times[i]; times[i] = times[j]; times[j] = x; }
    printf("%ld walkers, %ld merged within %ld; merge time min %ld median %ld 90%% %ld max %ld\n", cnt, merged, limit, times[0], times[cnt / 2], times[cnt * 9 / 10], times[cnt - 1]);
    return 0;
}
