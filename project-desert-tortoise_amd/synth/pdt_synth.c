/*
 * pdt_synth.c -- host side of the synthetic capture generator (see pdt_synth.h).
 * Built into libpdtsynth.so (ctypes from Python) and the synth_wav tool.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "pdt_synth.h"

static int16_t g_tab[PDT_SYNTH_TABLE_SIZE];
static int g_tab_ready;

const int16_t *pdt_synth_sine_table(void)
{
    if (!g_tab_ready) {
        for (uint32_t i = 0; i < PDT_SYNTH_TABLE_SIZE; i++)
            g_tab[i] = (int16_t)lrint(32767.0 * sin(2.0 * M_PI * (double)i / (double)PDT_SYNTH_TABLE_SIZE));
        g_tab_ready = 1;
    }
    return g_tab;
}

static uint32_t turns(double rad) { return (uint32_t)(int64_t)llrint(rad / (2.0 * M_PI) * 4294967296.0); }

void pdt_synth_default_params(pdt_synth_params *p, int kind, uint32_t sample_rate, double f0_hz, uint64_t seed)
{
    memset(p, 0, sizeof *p);
    p->kind = (uint32_t)kind;
    p->sample_rate = sample_rate;
    p->carrier_step = (uint32_t)(int64_t)llrint(f0_hz / (double)sample_rate * 4294967296.0);
    p->phase0 = turns(0.3);
    p->amplitude = 9830;                                   /* 0.3 full scale */
    if (kind == 0) {
        p->mod_index = turns(1.06);
        p->noise_gain = 1204;                              /* sigma = A/10/sqrt2 per component: 20 dB */
    } else {
        p->mod_index = turns(1.1);
        p->noise_gain = 677;                               /* 25 dB */
    }
    p->seed = seed;
}

/* A pass: noise only outside [signal_start, signal_end), the carrier offset ramps linearly from f_start_hz to f_end_hz between the
 * two (Doppler), the amplitude follows floor + (1 - floor) 4x(1 - x).  Everything is turned into integers here, once. */
void pdt_synth_set_pass(pdt_synth_params *p, uint64_t signal_start, uint64_t signal_end, double f_start_hz, double f_end_hz,
                        double env_floor)
{
    const double fs = (double)p->sample_rate;
    p->signal_start = signal_start;
    p->signal_end = signal_end;
    p->carrier_step = (uint32_t)(int64_t)llrint(f_start_hz / fs * 4294967296.0);
    p->doppler_q32 = 0;
    if (signal_end > signal_start + 1 && f_end_hz != f_start_hz)
        p->doppler_q32 = (int64_t)llrint((f_end_hz - f_start_hz) / fs * 4294967296.0 * 4294967296.0 / (double)(signal_end - signal_start));
    p->env_floor_q15 = env_floor > 0 ? (uint32_t)llrint(env_floor * 32768.0) : 0u;
    if (p->env_floor_q15 > 32768u) p->env_floor_q15 = 32768u;
}

void pdt_synth_fill(const pdt_synth_params *p, uint64_t start, uint64_t count, int16_t *out)
{
    const int16_t *tab = pdt_synth_sine_table();
    for (uint64_t i = 0; i < count; i++)
        pdt_synth_sample(p, tab, start + i, &out[2 * i], &out[2 * i + 1]);
}

static void put32(uint8_t *b, uint32_t v) { b[0] = v & 0xFF; b[1] = (v >> 8) & 0xFF; b[2] = (v >> 16) & 0xFF; b[3] = (v >> 24) & 0xFF; }
static void put16(uint8_t *b, uint32_t v) { b[0] = v & 0xFF; b[1] = (v >> 8) & 0xFF; }

void pdt_synth_wav_header(uint8_t h[44], uint32_t sample_rate, uint64_t nframes)
{
    uint64_t bytes = nframes * 4u;
    memcpy(h, "RIFF", 4);
    put32(h + 4, (uint32_t)(bytes + 36));
    memcpy(h + 8, "WAVE", 4);
    memcpy(h + 12, "fmt ", 4);
    put32(h + 16, 16);
    put16(h + 20, 1);
    put16(h + 22, 2);
    put32(h + 24, sample_rate);
    put32(h + 28, sample_rate * 4u);
    put16(h + 32, 4);
    put16(h + 34, 16);
    memcpy(h + 36, "data", 4);
    put32(h + 40, (uint32_t)bytes);
}

void pdt_synth_poes_frame(const pdt_synth_params *p, uint64_t fr, uint8_t out[104])
{
    for (uint32_t b = 0; b < 104; b++)
        out[b] = (uint8_t)pdt_synth_poes_frame_byte(p->seed, fr, b);
}

void pdt_synth_argos_payload(const pdt_synth_params *p, uint64_t burst, uint8_t out[7])
{
    for (uint32_t b = 0; b < 7; b++)
        out[b] = (uint8_t)pdt_synth_argos_payload_byte(p->seed, burst, b);
}

#ifdef PDT_SYNTH_MAIN
/* synth_wav poes|argos <sample_rate> <seconds> <f0_hz> <seed> out.wav [lead_s tail_s f_end_hz env_floor [noise_x]]
 * with the optional arguments: a pass -- lead_s / tail_s seconds of noise only at the ends, the carrier ramping from f0_hz to
 * f_end_hz in between, the amplitude envelope's floor (0 = flat), the noise amplitude multiplied by noise_x */
#include <stdio.h>
int main(int argc, char **argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: %s poes|argos sample_rate seconds f0_hz seed out.wav [lead_s tail_s f_end_hz env_floor [noise_x]]\n", argv[0]);
        return 2;
    }
    int kind = strcmp(argv[1], "argos") == 0;
    uint32_t fs = (uint32_t)strtoul(argv[2], NULL, 10);
    double secs = atof(argv[3]);
    pdt_synth_params p;
    pdt_synth_default_params(&p, kind, fs, atof(argv[4]), strtoull(argv[5], NULL, 10));
    uint64_t n = (uint64_t)llrint(secs * fs);
    if (argc >= 11) {
        const uint64_t lead = (uint64_t)llrint(atof(argv[7]) * fs), tail = (uint64_t)llrint(atof(argv[8]) * fs);
        pdt_synth_set_pass(&p, lead, n > tail ? n - tail : 0, atof(argv[4]), atof(argv[9]), atof(argv[10]));
        if (argc >= 12) p.noise_gain = (int32_t)llrint(p.noise_gain * atof(argv[11]));
    }
    FILE *f = fopen(argv[6], "wb");
    if (!f) { perror(argv[6]); return 1; }
    uint8_t hdr[44];
    pdt_synth_wav_header(hdr, fs, n);
    fwrite(hdr, 1, 44, f);
    enum { BLK = 1 << 16 };
    int16_t *buf = (int16_t *)malloc(BLK * 4);
    for (uint64_t s = 0; s < n; s += BLK) {
        uint64_t c = n - s < BLK ? n - s : BLK;
        pdt_synth_fill(&p, s, c, buf);
        fwrite(buf, 4, c, f);
    }
    fclose(f);
    free(buf);
    return 0;
}
#endif
