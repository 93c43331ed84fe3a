/*
 * sots_oracle.c -- CPU restatement of the reference's per-generation hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see sots_oracle.h, which also says what pins it: since
 * round 4 the reference's own device kernels, compiled as they stand and run on
 * the GPU - tests/test_ocl_reference.py; its CPU host path cannot be built
 * here).  Compile with -O2 -ffp-contract=off (the Makefile does): every fp32
 * expression below is meant to round exactly as written.
 *
 * Deliberate deviations from the reference's (defective) CPU code, SURVEY 8c:
 *  (1) oscillator phases restart at 0 for every individual (the CPU keeps them
 *      in Objective members, Evolutionary_Strategy.hpp:178-180; every GPU
 *      kernel starts at 0, ocl_program.cl:305-306);
 *  (2) every stage loops over P = populationLength;
 *  (3) 3-op synth second offset is params[5] (CPU, Evolutionary_Strategy.hpp:427),
 *      not params_scaled[4] (ocl_program.cl:368);
 *  (4) fitness over k = 0..N/2-1 (Evolutionary_Strategy_CPU.hpp:235), not the
 *      OpenCL kernel's N/2+3 bins (ocl_program.cl:606);
 *  (5) window from the double table (Evolutionary_Strategy.hpp:308-317);
 *  (6) counter-based Philox4x32-10 instead of wall-clock-seeded MWC64X;
 *  (7) scaleParams = min + v*(max-min) (ocl_program.cl:297); equals the CPU's
 *      v*max for the shipped mins = 0;
 *  (8) wavetable index clamped to [0, W-1] (NaN -> 0): the reference reads out of bounds
 *      when a wrapped phase rounds to exactly W or a phase step exceeds W.
 */
#include "sots_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------- */
/* Philox4x32-10                                                             */
/* ------------------------------------------------------------------------- */
void sots_or_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static uint32_t draw_word(uint64_t seed, uint32_t gid, uint32_t epoch, uint32_t index, uint32_t tag)
{
    uint32_t ctr[4] = { gid, epoch, index >> 2, tag };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t out[4];
    sots_or_philox4x32_10(ctr, key, out);
    return out[index & 3u];
}

/* (float)((int)MWC64X) / 2147483647.0f, ocl_program.cl:27,61 */
static float draw_unit(uint32_t w)
{
    return (float)((int32_t)w) / 2147483647.0f;
}

/* ------------------------------------------------------------------------- */
/* Objective                                                                 */
/* ------------------------------------------------------------------------- */
void sots_or_wavetable(float *table)
{
    /* Evolutionary_Strategy.hpp:328-331 */
    const float inv = 1.0f / ((float)SOTS_OR_WAVETABLE_SIZE - 1.0f);
    for (uint32_t i = 0; i < SOTS_OR_WAVETABLE_SIZE; ++i)
        table[i] = sinf((float)i * inv * 2 * (float)M_PI);
}

float sots_or_window(double *window, uint32_t n)
{
    /* Evolutionary_Strategy.hpp:296,308-317: fftOneOverSize is a float, the
     * factor accumulates in a float. */
    const float one_over = 1.0f / (float)n;
    const double two_pi = 2.0 * M_PI;
    float factor = 0.0f;
    for (uint32_t i = 0; i < n; ++i) {
        window[i] = (1.0 - cos((double)i * (one_over - 1) * two_pi));
        factor += window[i];
    }
    factor *= one_over;
    return factor;
}

uint32_t sots_or_synth_dims(uint32_t kind)
{
    switch (kind) {
    case SOTS_OR_SYNTH_2OP: return 4;
    case SOTS_OR_SYNTH_3OP_SERIES: return 6;
    case SOTS_OR_SYNTH_TRIPLE_PAR: return 12;
    case SOTS_OR_SYNTH_4OP_SERIES: return 8;
    default: return 0;
    }
}

static inline float tab_at(const float *table, float pos)
{
    /* (unsigned int)pos for every in-range phase.  The clamp is decided on the float so that it is
     * defined for every value (a phase beyond +-2^31 would make the C conversion undefined): at or
     * beyond W -> W-1, negative or NaN -> 0; the device's saturating v_cvt_i32_f32 + clamp agrees. */
    if (pos >= (float)SOTS_OR_WAVETABLE_SIZE) return table[SOTS_OR_WAVETABLE_SIZE - 1];
    if (!(pos > 0.0f)) return table[0];
    return table[(int32_t)pos];
}

#define WRAP_HI(p) do { if ((p) >= wsize) (p) -= wsize; } while (0)
#define WRAP_LO(p) do { if ((p) < 0.0f) (p) += wsize; } while (0)

void sots_or_synth(uint32_t kind, const float *values, const float *pmin, const float *pmax,
                   const float *table, uint32_t n, float *audio)
{
    const float wsize = (float)SOTS_OR_WAVETABLE_SIZE;
    /* w2srRatio, Evolutionary_Strategy.hpp:203 */
    const float c = SOTS_OR_WAVETABLE_SIZE / (float)SOTS_OR_SAMPLE_RATE;
    float p[SOTS_OR_MAX_DIMS];
    const uint32_t d = sots_or_synth_dims(kind);
    for (uint32_t i = 0; i < d; ++i) {
        /* the triple-parallel voice scales all three voices by entries 0..3,
         * Evolutionary_Strategy.hpp:453-455 */
        const uint32_t s = (kind == SOTS_OR_SYNTH_TRIPLE_PAR) ? (i & 3u) : i;
        p[i] = pmin[s] + values[i] * (pmax[s] - pmin[s]);
    }

    if (kind == SOTS_OR_SYNTH_2OP) {
        /* Evolutionary_Strategy.hpp:372-401 */
        const float mod = p[0] * p[1];
        const float fc = p[2];
        const float amp = p[3];
        const float inc1 = c * p[0];
        float pos1 = 0.0f, pos2 = 0.0f;
        for (uint32_t i = 0; i < n; ++i) {
            const float cur = tab_at(table, pos1) * mod + fc;
            pos1 += inc1;
            WRAP_HI(pos1);
            audio[i] = tab_at(table, pos2) * amp;
            pos2 += c * cur;
            WRAP_HI(pos2);
            WRAP_LO(pos2);
        }
    } else if (kind == SOTS_OR_SYNTH_3OP_SERIES) {
        /* Evolutionary_Strategy.hpp:407-445 */
        const float m1 = p[0] * p[1], m2 = p[2] * p[3], m3 = p[4] * p[5];
        const float inc1 = c * p[1];
        float pos1 = 0.0f, pos2 = 0.0f, pos3 = 0.0f;
        for (uint32_t i = 0; i < n; ++i) {
            const float cur1 = tab_at(table, pos1) * m1 + p[3];
            pos1 += inc1;
            WRAP_HI(pos1);
            const float cur2 = tab_at(table, pos2) * m2 + p[5];
            pos2 += c * cur1;
            WRAP_HI(pos2);
            WRAP_LO(pos2);
            audio[i] = tab_at(table, pos3) * m3;
            pos3 += c * cur2;
            WRAP_HI(pos3);
            WRAP_LO(pos3);
        }
    } else if (kind == SOTS_OR_SYNTH_4OP_SERIES) {
        /* build-defined: the 3-op chain of Evolutionary_Strategy.hpp:403-449
         * extended by one more modulator stage. */
        const float m1 = p[0] * p[1], m2 = p[2] * p[3], m3 = p[4] * p[5], m4 = p[6] * p[7];
        const float inc1 = c * p[1];
        float pos1 = 0.0f, pos2 = 0.0f, pos3 = 0.0f, pos4 = 0.0f;
        for (uint32_t i = 0; i < n; ++i) {
            const float cur1 = tab_at(table, pos1) * m1 + p[3];
            pos1 += inc1;
            WRAP_HI(pos1);
            const float cur2 = tab_at(table, pos2) * m2 + p[5];
            pos2 += c * cur1;
            WRAP_HI(pos2);
            WRAP_LO(pos2);
            const float cur3 = tab_at(table, pos3) * m3 + p[7];
            pos3 += c * cur2;
            WRAP_HI(pos3);
            WRAP_LO(pos3);
            audio[i] = tab_at(table, pos4) * m4;
            pos4 += c * cur3;
            WRAP_HI(pos4);
            WRAP_LO(pos4);
        }
    } else {
        /* Evolutionary_Strategy.hpp:457-494 */
        float mod[3], fc[3], amp[3], inc[3], pa[3] = { 0, 0, 0 }, pb[3] = { 0, 0, 0 };
        for (int j = 0; j < 3; ++j) {
            mod[j] = p[4 * j + 0] * p[4 * j + 1];
            fc[j] = p[4 * j + 2];
            amp[j] = p[4 * j + 3];
            inc[j] = c * p[4 * j + 0];
        }
        for (uint32_t i = 0; i < n; ++i) {
            float tot[3];
            for (int j = 0; j < 3; ++j) {
                const float cur = tab_at(table, pa[j]) * mod[j] + fc[j];
                pa[j] += inc[j];
                WRAP_HI(pa[j]);
                tot[j] = tab_at(table, pb[j]) * amp[j];
                pb[j] += c * cur;
                WRAP_HI(pb[j]);
                WRAP_LO(pb[j]);
            }
            audio[i] = (tot[0] + tot[1] + tot[2]) / 3.0;
        }
    }
}

/* The three voices with the arithmetic of the reference's OpenCL kernels (ocl_program.cl:280-443) instead of its CPU
 * path's: there the sample-rate ratio is the DOUBLE expression (WAVETABLE_SIZE / 44100.0), so an increment is a double
 * product rounded to float once (:308, :356, :417) and a modulated phase advances by a double multiply-add rounded to
 * float once (:319, :364, :368, :425); the 3-op voice's second offset is params[4] (:363; the CPU path: params[5]);
 * the index is the bare (uint) conversion (here: saturating, as the device converts), so a phase that lands on W reads
 * entry W - `table` must hold W + 1 entries.  `contract`: bit 0 - the float multiply-adds are fused, bit 1 - the double
 * ones are (OpenCL C lets the compiler contract; tests/test_ocl_reference.py finds which the compiled kernels did).
 * Only there to explain, bit for bit, what separates sots_or_synth from the reference's device kernels. */
static inline float ocl_tab(const float *table, float pos)
{
    if (!(pos > 0.0f)) return table[0];
    if (pos >= 4294967296.0f) return table[SOTS_OR_WAVETABLE_SIZE];
    const uint32_t i = (uint32_t)pos;
    return table[i > SOTS_OR_WAVETABLE_SIZE ? SOTS_OR_WAVETABLE_SIZE : i];
}
static inline float ocl_mad(float a, float b, float c, uint32_t contract) { return (contract & 1u) ? fmaf(a, b, c) : a * b + c; }
static inline float ocl_advance(float pos, float cur, uint32_t contract)
{
    const double cd = SOTS_OR_WAVETABLE_SIZE / 44100.0;
    return (contract & 2u) ? (float)fma(cd, (double)cur, (double)pos) : (float)((double)pos + cd * (double)cur);
}
void sots_or_synth_ocl(uint32_t kind, const float *values, const float *pmin, const float *pmax,
                       const float *table, uint32_t n, float *audio, uint32_t contract)
{
    const float wsize = (float)SOTS_OR_WAVETABLE_SIZE;
    const double cd = SOTS_OR_WAVETABLE_SIZE / 44100.0;
    float p[SOTS_OR_MAX_DIMS];
    const uint32_t d = sots_or_synth_dims(kind);
    for (uint32_t i = 0; i < d; ++i) p[i] = ocl_mad(values[i], pmax[i] - pmin[i], pmin[i], contract); /* :296, :348, :405 */
    if (kind == SOTS_OR_SYNTH_2OP) {
        const float mod = p[0] * p[1], fc = p[2], amp = p[3];
        const float inc1 = (float)(cd * (double)p[0]);
        float pos1 = 0.0f, pos2 = 0.0f;
        for (uint32_t i = 0; i < n; ++i) {
            const float cur = ocl_mad(ocl_tab(table, pos1), mod, fc, contract);
            audio[i] = ocl_tab(table, pos2) * amp;
            pos1 += inc1;
            pos2 = ocl_advance(pos2, cur, contract);
            WRAP_HI(pos1);   /* (:322-323: no lower wrap for the first phase) */
            WRAP_HI(pos2);
            WRAP_LO(pos2);
        }
    } else if (kind == SOTS_OR_SYNTH_3OP_SERIES) {
        const float m1 = p[0] * p[1], m2 = p[2] * p[3], m3 = p[4] * p[5];
        const float inc1 = (float)(cd * (double)p[1]);
        float pos1 = 0.0f, pos2 = 0.0f, pos3 = 0.0f;
        for (uint32_t i = 0; i < n; ++i) {
            const float cur1 = ocl_mad(ocl_tab(table, pos1), m1, p[3], contract);
            pos1 += inc1;
            const float cur2 = ocl_mad(ocl_tab(table, pos2), m2, p[4], contract);
            pos2 = ocl_advance(pos2, cur1, contract);
            audio[i] = ocl_tab(table, pos3) * m3;
            pos3 = ocl_advance(pos3, cur2, contract);
            WRAP_HI(pos1);
            WRAP_LO(pos1);
            WRAP_HI(pos2);
            WRAP_LO(pos2);
            WRAP_HI(pos3);
            WRAP_LO(pos3);
        }
    } else if (kind == SOTS_OR_SYNTH_TRIPLE_PAR) {
        float mod[3], fc[3], amp[3], inc[3], pa[3] = { 0, 0, 0 }, pb[3] = { 0, 0, 0 };
        for (int j = 0; j < 3; ++j) {
            mod[j] = p[4 * j + 0] * p[4 * j + 1];
            fc[j] = p[4 * j + 2];
            amp[j] = p[4 * j + 3];
            inc[j] = (float)(cd * (double)p[4 * j + 0]);
        }
        for (uint32_t i = 0; i < n; ++i) {
            float tot[3];
            for (int j = 0; j < 3; ++j) {
                const float cur = ocl_mad(ocl_tab(table, pa[j]), mod[j], fc[j], contract);
                tot[j] = ocl_tab(table, pb[j]) * amp[j];
                pa[j] += inc[j];
                pb[j] = ocl_advance(pb[j], cur, contract);
                WRAP_HI(pa[j]);
                WRAP_HI(pb[j]);
                WRAP_LO(pb[j]);
            }
            audio[i] = (float)((double)(tot[0] + tot[1] + tot[2]) / 3.0);
        }
    } else {
        for (uint32_t i = 0; i < n; ++i) audio[i] = 0.0f; /* (no OpenCL kernel for the build-defined 4-op voice) */
    }
}

/* ------------------------------------------------------------------------- */
/* fp64 real FFT (stands in for FFTW's fftw_plan_dft_r2c_1d,                 */
/* Evolutionary_Strategy.hpp:286,511: a forward real DFT is mathematically   */
/* fixed, any correct fp64 FFT agrees with FFTW to ~1e-15 relative).         */
/* ------------------------------------------------------------------------- */
typedef struct {
    uint32_t n;        /* real length */
    double *tw_re, *tw_im; /* e^{-2 pi i k / n}, k < n/2 */
    uint32_t *rev;     /* bit reversal for n/2 */
    double *zr, *zi;   /* work, n/2 */
} rfft_plan;

#define MAX_PLANS 16
static rfft_plan g_plans[MAX_PLANS];
static int g_nplans = 0;

static rfft_plan *plan_for(uint32_t n)
{
    for (int i = 0; i < g_nplans; ++i)
        if (g_plans[i].n == n) return &g_plans[i];
    if (g_nplans == MAX_PLANS) abort();
    rfft_plan *pl = &g_plans[g_nplans++];
    const uint32_t m = n / 2;
    pl->n = n;
    pl->tw_re = (double *)malloc(sizeof(double) * m);
    pl->tw_im = (double *)malloc(sizeof(double) * m);
    pl->rev = (uint32_t *)malloc(sizeof(uint32_t) * m);
    pl->zr = (double *)malloc(sizeof(double) * m);
    pl->zi = (double *)malloc(sizeof(double) * m);
    for (uint32_t k = 0; k < m; ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)n;
        pl->tw_re[k] = cos(a);
        pl->tw_im[k] = sin(a);
    }
    uint32_t bits = 0;
    while ((1u << bits) < m) ++bits;
    for (uint32_t i = 0; i < m; ++i) {
        uint32_t r = 0;
        for (uint32_t b = 0; b < bits; ++b)
            if (i & (1u << b)) r |= 1u << (bits - 1 - b);
        pl->rev[i] = r;
    }
    return pl;
}

/* bins 0..n/2 of the forward DFT of x[0..n-1] */
static void rfft_exec_ws(const rfft_plan *pl, const double *x, double *re, double *im, double *zr, double *zi)
{
    const uint32_t n = pl->n, m = n / 2; /* zr, zi: caller's work space, n/2 doubles each */
    for (uint32_t i = 0; i < m; ++i) {
        const uint32_t r = pl->rev[i];
        zr[r] = x[2 * i];
        zi[r] = x[2 * i + 1];
    }
    /* radix-2 DIT on m complex points; twiddle e^{-2 pi i j/len} = tw[j * (n/len)] */
    for (uint32_t len = 2; len <= m; len <<= 1) {
        const uint32_t half = len / 2, stride = n / len;
        for (uint32_t base = 0; base < m; base += len) {
            for (uint32_t j = 0; j < half; ++j) {
                const double wr = pl->tw_re[j * stride], wi = pl->tw_im[j * stride];
                const uint32_t a = base + j, b = a + half;
                const double tr = zr[b] * wr - zi[b] * wi;
                const double ti = zr[b] * wi + zi[b] * wr;
                zr[b] = zr[a] - tr;
                zi[b] = zi[a] - ti;
                zr[a] += tr;
                zi[a] += ti;
            }
        }
    }
    /* split: X[k] = E + W^k O, E = (Z[k]+conj Z[m-k])/2, O = -i (Z[k]-conj Z[m-k])/2 */
    for (uint32_t k = 0; k <= m; ++k) {
        const uint32_t ka = (k == m) ? 0 : k, kb = (k == 0 || k == m) ? 0 : m - k;
        const double ar = zr[ka], ai = zi[ka], br = zr[kb], bi = -zi[kb];
        const double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);
        const double dr = 0.5 * (ar - br), di = 0.5 * (ai - bi);
        /* O = -i * d = (di, -dr) */
        const double or_ = di, oi = -dr;
        double wr, wi;
        if (k == m) { wr = -1.0; wi = 0.0; } else { wr = pl->tw_re[k]; wi = pl->tw_im[k]; }
        re[k] = er + (or_ * wr - oi * wi);
        im[k] = ei + (or_ * wi + oi * wr);
    }
}

static void rfft_exec(rfft_plan *pl, const double *x, double *re, double *im)
{
    rfft_exec_ws(pl, x, re, im, pl->zr, pl->zi); /* the plan's own work space: single-threaded callers */
}

void sots_or_rfft(const float *audio, uint32_t n, const double *window, double *re, double *im)
{
    rfft_plan *pl = plan_for(n);
    double *x = (double *)malloc(sizeof(double) * n);
    for (uint32_t i = 0; i < n; ++i) x[i] = audio[i] * window[i]; /* Evolutionary_Strategy.hpp:506-508 */
    rfft_exec(pl, x, re, im);
    free(x);
}

void sots_or_rfft_naive(const float *audio, uint32_t n, const double *window, double *re, double *im)
{
    for (uint32_t k = 0; k <= n / 2; ++k) {
        double sr = 0.0, si = 0.0;
        for (uint32_t i = 0; i < n; ++i) {
            const double x = audio[i] * window[i];
            const uint64_t ph = ((uint64_t)k * i) % n;
            const double a = -2.0 * M_PI * (double)ph / (double)n;
            sr += x * cos(a);
            si += x * sin(a);
        }
        re[k] = sr;
        im[k] = si;
    }
}

void sots_or_spectrum(const float *audio, uint32_t n, const double *window, float window_factor,
                      float *mag)
{
    /* Evolutionary_Strategy.hpp:503-523 / 524-542 */
    rfft_plan *pl = plan_for(n);
    const uint32_t half = n / 2;
    double *x = (double *)malloc(sizeof(double) * (size_t)(n + 2 * (half + 1) + 2 * half));
    double *re = x + n, *im = re + half + 1, *zr = im + half + 1, *zi = zr + half;
    for (uint32_t i = 0; i < n; ++i) x[i] = audio[i] * window[i];
    rfft_exec_ws(pl, x, re, im, zr, zi); /* private work space: callable from several threads once the plan exists */
    const float one_over_size = 1.0f / (float)n;               /* :296 */
    const float one_over_wf = 1.f / window_factor;              /* :317 */
    for (uint32_t k = 0; k < half; ++k) {
        const float raw = hypotf((float)re[k], (float)im[k]);   /* :517 */
        const float scaled = raw * one_over_size;               /* :518 */
        mag[k] = scaled * one_over_wf;                          /* :519 */
    }
    free(x);
}

float sots_or_fitness(const float *mag, const float *target, uint32_t half)
{
    /* Evolutionary_Strategy_CPU.hpp:230-265 */
    float error = 0.0f;
    for (uint32_t j = 0; j < half; ++j) {
        const float t = mag[j] - target[j];
        error += t * t;
    }
    return error;
}

/* ------------------------------------------------------------------------- */
/* Evolutionary operators                                                    */
/* ------------------------------------------------------------------------- */
void sots_or_init_population(float *values, float *steps, uint32_t p, uint32_t d,
                             uint64_t seed, uint32_t gid_base, uint32_t chunk)
{
    /* ocl_program.cl:56-65 */
    for (uint32_t i = 0; i < p; ++i)
        for (uint32_t j = 0; j < d; ++j) {
            const float r = draw_unit(draw_word(seed, gid_base + i, chunk, j, SOTS_OR_TAG_INIT));
            steps[(size_t)i * d + j] = 0.1f;
            values[(size_t)i * d + j] = (r < 0.0f) ? -r : r;
        }
}

void sots_or_recombine(const float *vin, const float *sin_, float *vout, float *sout,
                       uint32_t p, uint32_t d, uint32_t num_parents, uint32_t block)
{
    /* ocl_program.cl:99-148.  NUM_WGS_FOR_PARENTS = numParents / WRKGRPSIZE
     * (Evolutionary_Strategy_OpenCL.hpp:130) is 0 when numParents < block,
     * which the reference then uses as a modulus; taken as 1 here. */
    uint32_t npb = num_parents / block;
    if (npb == 0) npb = 1;
    const uint32_t nblocks = p / block;
    for (uint32_t b = 0; b < nblocks; ++b) {
        const uint32_t pb = b % npb;
        for (uint32_t l = 0; l < block; ++l)
            for (uint32_t g = 0; g < d; ++g) {
                /* new_idx = (l*D + g + D*(g*(b+1))) mod (block*D), :132-133 */
                const uint32_t dst_l = (uint32_t)(((uint64_t)l + (uint64_t)g * (b + 1)) % block);
                const size_t src = ((size_t)pb * block + l) * d + g;
                const size_t dst = ((size_t)b * block + dst_l) * d + g;
                vout[dst] = vin[src];
                sout[dst] = sin_[src];
            }
    }
}

float sots_or_draw_unit(uint32_t word) { return draw_unit(word); }

/* One gene's mutation from the 13 random words the rule consumes (ocl_program.cl:168-188): words[0] is the coin,
 * words[1..12] make the "gaussian" (gauss_rand, :21-31).  sots_or_mutate below draws the words from the counter-based
 * generator; tests hand in the reference's MWC64X words to compare the RULE with the reference's kernel. */
void sots_or_mutate_gene(float *value, float *step, uint32_t d, const uint32_t words[13])
{
    /* constants: Evolutionary_Strategy.hpp:611-627 */
    const float mpi = (float)3.14159265358979323846;
    const float alpha = 1.4f;
    const float one_over_alpha = 1.f / alpha;
    const float root_two_over_pi = sqrtf(2.f / (float)mpi);
    const float beta_scale = 1.f / (float)d;
    const float beta = sqrtf(beta_scale);
    const float ek = (words[0] % 2u == 0u) ? alpha : one_over_alpha;
    float s = *step;
    const float x = *value;
    float sum = 0.0f;
    for (uint32_t t = 0; t < 12; ++t) sum += draw_unit(words[1 + t]);
    sum /= 12.0f;
    float gauss = sum;
    float new_x = x + ek * s * gauss;
    if (new_x < 0.0f || new_x > 1.0f) {
        gauss = gauss * -0.5f;
        new_x = x + ek * s * gauss;
    }
    const float es = expf(fabsf(gauss) - root_two_over_pi);
    s *= powf(ek, beta) * powf(es, beta_scale);
    *step = s;
    *value = new_x;
}

void sots_or_mutate(float *values, float *steps, uint32_t p, uint32_t d,
                    uint64_t seed, uint32_t gid_base, uint32_t generation)
{
    for (uint32_t i = 0; i < p; ++i)
        for (uint32_t j = 0; j < d; ++j) {
            /* 13 draws per gene, at counter indices 16 j .. 16 j + 12 */
            uint32_t words[13];
            for (uint32_t t = 0; t < 13; ++t)
                words[t] = draw_word(seed, gid_base + i, generation, j * 16u + t, SOTS_OR_TAG_MUTATE);
            sots_or_mutate_gene(&values[(size_t)i * d + j], &steps[(size_t)i * d + j], d, words);
        }
}

static int fit_less(float a, float b)
{
    if (isnan(a)) return 0;
    if (isnan(b)) return 1;
    return a < b;
}

void sots_or_sort_perm(const float *fitness, uint32_t p, uint32_t *perm)
{
    /* stable ascending (what the reference's bubble sort produces,
     * Evolutionary_Strategy.hpp:108-124), NaN after every number. Bottom-up merge. */
    uint32_t *a = perm, *b = (uint32_t *)malloc(sizeof(uint32_t) * p);
    for (uint32_t i = 0; i < p; ++i) a[i] = i;
    for (uint32_t w = 1; w < p; w <<= 1) {
        for (uint32_t lo = 0; lo < p; lo += 2 * w) {
            uint32_t mid = lo + w < p ? lo + w : p, hi = lo + 2 * w < p ? lo + 2 * w : p;
            uint32_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                if (fit_less(fitness[a[j]], fitness[a[i]])) b[k++] = a[j++];
                else b[k++] = a[i++];
            }
            while (i < mid) b[k++] = a[i++];
            while (j < hi) b[k++] = a[j++];
        }
        uint32_t *t = a; a = b; b = t;
    }
    if (a != perm) { memcpy(perm, a, sizeof(uint32_t) * p); free(a); }
    else free(b);
}

/* ------------------------------------------------------------------------- */
/* Whole strategy                                                            */
/* ------------------------------------------------------------------------- */
struct sots_or_es {
    sots_or_config cfg;
    uint32_t p, d, n, half, generation;
    float *values, *steps, *fitness;     /* current population, [P][D], [P][D], [P] */
    float *values2, *steps2, *fitness2;  /* scratch */
    float *audio, *mag, *target, *table;
    double *window;
    float window_factor;
    uint32_t *perm;
};

sots_or_es *sots_or_es_create(const sots_or_config *cfg)
{
    sots_or_es *es = (sots_or_es *)calloc(1, sizeof(*es));
    es->cfg = *cfg;
    es->p = cfg->num_parents + cfg->num_offspring;
    es->d = cfg->num_dims;
    es->n = 1u << cfg->audio_log2;
    es->half = es->n / 2;
    const size_t pd = (size_t)es->p * es->d;
    es->values = (float *)calloc(pd, sizeof(float));
    es->steps = (float *)calloc(pd, sizeof(float));
    es->fitness = (float *)calloc(es->p, sizeof(float));
    es->values2 = (float *)calloc(pd, sizeof(float));
    es->steps2 = (float *)calloc(pd, sizeof(float));
    es->fitness2 = (float *)calloc(es->p, sizeof(float));
    es->audio = (float *)calloc((size_t)es->p * es->n, sizeof(float));
    es->mag = (float *)calloc((size_t)es->p * es->half, sizeof(float));
    es->target = (float *)calloc(es->half, sizeof(float));
    es->table = (float *)malloc(sizeof(float) * SOTS_OR_WAVETABLE_SIZE);
    es->window = (double *)malloc(sizeof(double) * es->n);
    es->perm = (uint32_t *)malloc(sizeof(uint32_t) * es->p);
    sots_or_wavetable(es->table);
    es->window_factor = sots_or_window(es->window, es->n);
    return es;
}

void sots_or_es_destroy(sots_or_es *es)
{
    if (!es) return;
    free(es->values); free(es->steps); free(es->fitness);
    free(es->values2); free(es->steps2); free(es->fitness2);
    free(es->audio); free(es->mag); free(es->target); free(es->table);
    free(es->window); free(es->perm); free(es);
}

void sots_or_es_set_target_audio(sots_or_es *es, const float *audio)
{
    /* Evolutionary_Strategy_CPU.hpp:484-489 */
    sots_or_spectrum(audio, es->n, es->window, es->window_factor, es->target);
}

void sots_or_es_set_target_spectrum(sots_or_es *es, const float *mag)
{
    memcpy(es->target, mag, sizeof(float) * es->half);
}

void sots_or_es_init_population(sots_or_es *es, uint32_t chunk)
{
    sots_or_init_population(es->values, es->steps, es->p, es->d, es->cfg.seed, es->cfg.gid_base, chunk);
    memset(es->fitness, 0, sizeof(float) * es->p);
    es->generation = 0;
}

void sots_or_es_write_population(sots_or_es *es, const float *values, const float *steps, const float *fitness)
{
    const size_t pd = (size_t)es->p * es->d;
    if (values) memcpy(es->values, values, pd * sizeof(float));
    if (steps) memcpy(es->steps, steps, pd * sizeof(float));
    if (fitness) memcpy(es->fitness, fitness, es->p * sizeof(float));
}

void sots_or_es_read_population(const sots_or_es *es, float *values, float *steps, float *fitness)
{
    const size_t pd = (size_t)es->p * es->d;
    if (values) memcpy(values, es->values, pd * sizeof(float));
    if (steps) memcpy(steps, es->steps, pd * sizeof(float));
    if (fitness) memcpy(fitness, es->fitness, es->p * sizeof(float));
}

void sots_or_es_set_generation(sots_or_es *es, uint32_t generation) { es->generation = generation; }

void sots_or_es_recombine(sots_or_es *es)
{
    sots_or_recombine(es->values, es->steps, es->values2, es->steps2, es->p, es->d,
                      es->cfg.num_parents, es->cfg.recomb_block);
    float *t;
    t = es->values; es->values = es->values2; es->values2 = t;
    t = es->steps; es->steps = es->steps2; es->steps2 = t;
}

void sots_or_es_mutate(sots_or_es *es)
{
    sots_or_mutate(es->values, es->steps, es->p, es->d, es->cfg.seed, es->cfg.gid_base, es->generation);
}

/* Threads for sots_or_es_evaluate (bench.py's all-cores CPU baseline).  The default, 1, is the
 * reference's behaviour: its CPU path has no threads anywhere.  Individuals are independent, so
 * the results do not depend on the thread count. */
static int g_eval_threads = 1;
void sots_or_set_threads(int n) { g_eval_threads = n < 1 ? 1 : n; }

void sots_or_es_evaluate(sots_or_es *es)
{
    (void)plan_for(es->n); /* create the shared plan before any thread needs it */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_eval_threads)
#endif
    for (uint32_t i = 0; i < es->p; ++i) {
        float *a = es->audio + (size_t)i * es->n;
        float *m = es->mag + (size_t)i * es->half;
        sots_or_synth(es->cfg.synth_kind, es->values + (size_t)i * es->d, es->cfg.param_min,
                      es->cfg.param_max, es->table, es->n, a);
        sots_or_spectrum(a, es->n, es->window, es->window_factor, m);
        es->fitness[i] = sots_or_fitness(m, es->target, es->half);
    }
}

void sots_or_es_sort(sots_or_es *es)
{
    sots_or_sort_perm(es->fitness, es->p, es->perm);
    const uint32_t d = es->d;
    for (uint32_t r = 0; r < es->p; ++r) {
        const uint32_t s = es->perm[r];
        memcpy(es->values2 + (size_t)r * d, es->values + (size_t)s * d, d * sizeof(float));
        memcpy(es->steps2 + (size_t)r * d, es->steps + (size_t)s * d, d * sizeof(float));
        es->fitness2[r] = es->fitness[s];
    }
    float *t;
    t = es->values; es->values = es->values2; es->values2 = t;
    t = es->steps; es->steps = es->steps2; es->steps2 = t;
    t = es->fitness; es->fitness = es->fitness2; es->fitness2 = t;
}

void sots_or_es_generation(sots_or_es *es)
{
    /* Evolutionary_Strategy_CPU.hpp:353-418 */
    sots_or_es_recombine(es);
    sots_or_es_mutate(es);
    sots_or_es_evaluate(es);
    sots_or_es_sort(es);
    es->generation++;
}

void sots_or_es_inject(sots_or_es *es, const float *rows, uint32_t n_rows)
{
    const uint32_t d = es->d, w = 2 * d + 1;
    /* tail of the rows recombination reads: whole blocks of parents (ocl_program.cl:99-112) */
    uint32_t npb = es->cfg.num_parents / es->cfg.recomb_block;
    if (npb == 0) npb = 1;
    const uint32_t first = npb * es->cfg.recomb_block - n_rows;
    for (uint32_t r = 0; r < n_rows; ++r) {
        const float *row = rows + (size_t)r * w;
        es->fitness[first + r] = row[0];
        memcpy(es->values + (size_t)(first + r) * d, row + 1, d * sizeof(float));
        memcpy(es->steps + (size_t)(first + r) * d, row + 1 + d, d * sizeof(float));
    }
}

void sots_or_es_pack_elites(const sots_or_es *es, float *rows, uint32_t n_rows)
{
    const uint32_t d = es->d, w = 2 * d + 1;
    for (uint32_t r = 0; r < n_rows; ++r) {
        float *row = rows + (size_t)r * w;
        row[0] = es->fitness[r];
        memcpy(row + 1, es->values + (size_t)r * d, d * sizeof(float));
        memcpy(row + 1 + d, es->steps + (size_t)r * d, d * sizeof(float));
    }
}

const float *sots_or_es_audio(const sots_or_es *es) { return es->audio; }
const float *sots_or_es_spectrum(const sots_or_es *es) { return es->mag; }
