/*
 * pdt_synth.h -- integer-only, counter-based synthetic IQ capture models.
 *
 * Every output sample is a pure function of (parameters, sample index), so a
 * capture can be generated in any order, in parallel, on the host or on the
 * GPU, and is bit-identical everywhere (no floating point after the one-time
 * 65 536-entry sine table, whose CRC is pinned in tests/test_synth.py).
 *
 * Signal models (validated there against the reference decoder):
 *  POES/TIP : s[n] = A exp(j(2 pi f0 n/Fs + phi0 + m d[n])) + w[n], m = 1.06 rad,
 *             d = Manchester symbol (+1/-1) at 16 640 sym/s ('1' -> +1,-1,
 *             '0' -> -1,+1: the inverse of common/ManchesterDecode.c:60-83),
 *             8 320 bit/s, contiguous 104-byte minor frames "ED E2 0b000xxxxx .."
 *             (the 19-bit sync word of POESTIPdemod/main.c:454 is ED E2 + 000),
 *             complex noise 20 dB below the carrier.
 *  ARGOS    : noise floor plus one burst every 1.5 s: 160 ms of unmodulated
 *             carrier, then 400 bit/s Manchester PM (m = 1.1 rad) of
 *             15 x '1', 0001 0111, 1, 0000, 56 payload bits, 1010
 *             (sync "0001011110000": ARGOSdemod/main.c:284), noise 25 dB down.
 *
 * A pass as a receiver sees it (round 6; every field 0 = off, the captures above unchanged): noise only before
 * `signal_start` (the receiver is on before the satellite rises) and from `signal_end` on (it stays on after it has set);
 * between the two a linear Doppler ramp -- the carrier's phase step grows by `doppler_q32` / 2^32 per sample, an integer
 * phase accumulator in closed form, so that any sample is still a pure function of its index -- and an amplitude envelope
 * `env_floor_q15` + (1 - floor) 4x(1 - x), x = the position inside the pass: strongest at culmination, `floor` of that at
 * the horizon; and a fade -- `fade_len` samples from `fade_start` at `fade_q15` of that amplitude (an antenna null, a building).
 *
 * All arithmetic is uint32/uint64/int32; phases are 32-bit turns (2^32 = 2 pi).
 */
#ifndef PDT_SYNTH_H
#define PDT_SYNTH_H
#include <stdint.h>

#ifdef __HIPCC__
#define PDT_SYNTH_FN __host__ __device__ static inline
#else
#define PDT_SYNTH_FN static inline
#endif

#define PDT_SYNTH_TABLE_BITS 16
#define PDT_SYNTH_TABLE_SIZE (1u << PDT_SYNTH_TABLE_BITS)

typedef struct pdt_synth_params {
    uint32_t kind;          /* 0 = POES, 1 = ARGOS                                        */
    uint32_t sample_rate;   /* Hz                                                         */
    uint32_t carrier_step;  /* round(f0 / Fs * 2^32)                                      */
    uint32_t phase0;        /* phi0 in turns * 2^32                                       */
    uint32_t mod_index;     /* m in turns * 2^32                                          */
    int32_t  amplitude;     /* carrier amplitude, int16 units (9830 = 0.3 FS)             */
    int32_t  noise_gain;    /* Q16 multiplier applied to the 4-uniform Irwin-Hall sum     */
    uint64_t seed;          /* payload + noise seed                                       */
    uint64_t signal_start;  /* samples before this index are noise only (carrier off): a  */
                            /* receiver switched on before the satellite rises; 0 = none  */
    uint64_t signal_end;    /* samples from this index on are noise only; 0 = never       */
    int64_t  doppler_q32;   /* carrier_step changes by doppler_q32 / 2^32 per sample      */
                            /* behind signal_start (a linear Doppler ramp); 0 = none      */
    uint32_t env_floor_q15; /* amplitude envelope over [signal_start, signal_end): this   */
                            /* fraction (Q15) at both ends, 1 in the middle; 0 = flat     */
    uint32_t fade_q15;      /* a fade inside the pass: the amplitude times this fraction (Q15)  */
    uint64_t fade_start;    /* ... over the samples [fade_start, fade_start + fade_len); len 0 = */
    uint64_t fade_len;      /* none                                                               */
} pdt_synth_params;

PDT_SYNTH_FN uint64_t pdt_synth_mix(uint64_t z)
{   /* splitmix64 finaliser */
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* byte `b` (0..103) of POES minor frame `fr` */
PDT_SYNTH_FN uint32_t pdt_synth_poes_frame_byte(uint64_t seed, uint64_t fr, uint32_t b)
{
    if (b == 0) return 0xEDu;
    if (b == 1) return 0xE2u;
    uint32_t v = (uint32_t)(pdt_synth_mix(seed ^ (fr * 104u + b) * 0xD6E8FEB86659FD93ull) >> 24) & 0xFFu;
    return b == 2 ? (v & 0x1Fu) : v;
}

/* payload byte `b` (0..6) of ARGOS burst `burst` */
PDT_SYNTH_FN uint32_t pdt_synth_argos_payload_byte(uint64_t seed, uint64_t burst, uint32_t b)
{
    return (uint32_t)(pdt_synth_mix(seed ^ (burst * 8u + b + 1u) * 0xA24BAED4963EE407ull) >> 24) & 0xFFu;
}

/* bit `j` (0..87) of an ARGOS burst message */
PDT_SYNTH_FN uint32_t pdt_synth_argos_bit(uint64_t seed, uint64_t burst, uint32_t j)
{
    if (j < 15) return 1;
    if (j < 23) return (0x17u >> (22 - j)) & 1u;         /* 0001 0111 */
    if (j == 23) return 1;
    if (j < 28) return 0;
    if (j < 84) {
        uint32_t k = j - 28;
        return (pdt_synth_argos_payload_byte(seed, burst, k >> 3) >> (7 - (k & 7))) & 1u;
    }
    return (j & 1u) ? 0u : 1u;                            /* 84..87 = 1010 */
}

PDT_SYNTH_FN int32_t pdt_synth_noise(uint64_t h, int32_t gain_q16)
{   /* Irwin-Hall(4) of 16-bit uniforms, zero mean, scaled by gain/65536 */
    int32_t s = (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) + (int32_t)((h >> 32) & 0xFFFF) +
                (int32_t)((h >> 48) & 0xFFFF) - 2 * 65535;
    int64_t v = (int64_t)s * gain_q16;
    return (int32_t)((v + (v >= 0 ? 32768 : -32768)) / 65536);
}

PDT_SYNTH_FN int16_t pdt_synth_clip16(int32_t v)
{
    return (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
}

/* One IQ sample.  sine_tab[i] = lrint(32767 sin(2 pi i / 65536)). */
PDT_SYNTH_FN void pdt_synth_sample(const pdt_synth_params *p, const int16_t *sine_tab, uint64_t n, int16_t *i_out,
                                   int16_t *q_out)
{
    uint32_t theta = (uint32_t)(n * (uint64_t)p->carrier_step) + p->phase0;
    int32_t amp = n < p->signal_start ? 0 : p->amplitude;
    if (p->signal_end && n >= p->signal_end) amp = 0;
    if (amp && p->doppler_q32) {
        /* sum over the m samples since signal_start of (k * doppler_q32) / 2^32 turns, k = 0 .. m - 1, floor taken of the sum:
           bits 32..63 of the two's complement product doppler_q32 * m (m - 1) / 2 -- its low 64 bits are exact under wrap-around */
        const uint64_t m = n - p->signal_start;
        const uint64_t tri = (m & 1u) ? m * ((m - 1u) >> 1) : (m >> 1) * (m - 1u);
        theta += (uint32_t)(((uint64_t)p->doppler_q32 * tri) >> 32);
    }
    if (amp && p->env_floor_q15 && p->signal_end > p->signal_start) {
        const uint64_t m = n - p->signal_start, len = p->signal_end - p->signal_start;      /* (len < 2^40) */
        const uint64_t x = (m << 16) / len;                                                 /* Q16 position, < 65536 */
        const uint64_t par = 4u * x * (65536u - x);                                         /* Q32: 4x(1 - x) <= 2^32 */
        const uint64_t e = p->env_floor_q15 + (((32768u - (uint64_t)p->env_floor_q15) * par) >> 32);   /* Q15 */
        amp = (int32_t)(((uint64_t)amp * e + 16384u) >> 15);
    }
    if (amp && p->fade_len && n >= p->fade_start && n - p->fade_start < p->fade_len)
        amp = (int32_t)(((uint64_t)amp * p->fade_q15 + 16384u) >> 15);
    if (p->kind == 0) {
        uint64_t k = (n * 16640ull) / p->sample_rate;     /* Manchester symbol index */
        uint64_t bit = k >> 1;
        uint64_t fr = bit / 832u;
        uint32_t j = (uint32_t)(bit % 832u);
        uint32_t v = (pdt_synth_poes_frame_byte(p->seed, fr, j >> 3) >> (7 - (j & 7))) & 1u;
        int up = (v != 0) ^ (int)(k & 1);                 /* '1' -> +,-   '0' -> -,+ */
        theta += up ? p->mod_index : (uint32_t)(0u - p->mod_index);
    } else {
        uint64_t period = (uint64_t)p->sample_rate * 3u / 2u;          /* 1.5 s */
        uint64_t burst = n / period;
        uint64_t off = n % period;
        uint64_t lead = (uint64_t)p->sample_rate * 4u / 25u;           /* 160 ms */
        if (off < lead) {
            /* unmodulated carrier */
        } else {
            uint64_t k = ((off - lead) * 800ull) / p->sample_rate;     /* symbol index */
            if (k < 176) {
                uint32_t v = pdt_synth_argos_bit(p->seed, burst, (uint32_t)(k >> 1));
                int up = (v != 0) ^ (int)(k & 1);
                theta += up ? p->mod_index : (uint32_t)(0u - p->mod_index);
            } else {
                amp = 0;                                               /* carrier off */
            }
        }
    }
    const uint32_t idx = theta >> (32 - PDT_SYNTH_TABLE_BITS);
    const int32_t s = sine_tab[idx];
    const int32_t c = sine_tab[(idx + PDT_SYNTH_TABLE_SIZE / 4) & (PDT_SYNTH_TABLE_SIZE - 1)];
    int32_t vi = (amp * c + (1 << 14)) >> 15;
    int32_t vq = (amp * s + (1 << 14)) >> 15;
    vi += pdt_synth_noise(pdt_synth_mix(p->seed ^ (2 * n + 0x51ull) * 0xC2B2AE3D27D4EB4Full), p->noise_gain);
    vq += pdt_synth_noise(pdt_synth_mix(p->seed ^ (2 * n + 0x52ull) * 0xC2B2AE3D27D4EB4Full), p->noise_gain);
    *i_out = pdt_synth_clip16(vi);
    *q_out = pdt_synth_clip16(vq);
}

#ifdef __cplusplus
extern "C" {
#endif
/* host-side helpers (pdt_synth.c) */
const int16_t *pdt_synth_sine_table(void);
void pdt_synth_default_params(pdt_synth_params *p, int kind, uint32_t sample_rate, double f0_hz, uint64_t seed);
/* a pass (round 6): noise only outside [signal_start, signal_end), linear Doppler ramp f_start_hz -> f_end_hz, amplitude envelope */
void pdt_synth_set_pass(pdt_synth_params *p, uint64_t signal_start, uint64_t signal_end, double f_start_hz, double f_end_hz,
                        double env_floor);
/* fill out[2*count] with samples [start, start+count) */
void pdt_synth_fill(const pdt_synth_params *p, uint64_t start, uint64_t count, int16_t *out);
/* canonical 44-byte header (fmt=1, 2 ch, 16 bit) for `nframes` IQ samples */
void pdt_synth_wav_header(uint8_t hdr[44], uint32_t sample_rate, uint64_t nframes);
/* frames / payloads the generator transmitted (for round-trip checks) */
void pdt_synth_poes_frame(const pdt_synth_params *p, uint64_t fr, uint8_t out[104]);
void pdt_synth_argos_payload(const pdt_synth_params *p, uint64_t burst, uint8_t out[7]);
#ifdef __cplusplus
}
#endif
#endif
