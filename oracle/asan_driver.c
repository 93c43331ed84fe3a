/*
 * asan_driver.c -- drives every entry point of the CPU oracle (sots_oracle.c) at awkward sizes under
 * AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`, run by tests/test_sanitizers.py).
 *
 * TEST INFRASTRUCTURE ONLY, like the oracle itself.  The class of bug it is there to prove absent is the
 * reference's own: `new float(n)` where `new float[n]` was meant (Evolutionary_Strategy.hpp:236-244) and loops
 * over populationSize (a byte-ish count) where populationLength was meant - both write past
 * an allocation without any visible symptom.  Prints a checksum per case; the test compares it with the same
 * calls made through the ordinary (unsanitised) build.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sots_oracle.h"

static unsigned long long fnv(unsigned long long h, const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

static unsigned long long run_case(uint32_t kind, uint32_t parents, uint32_t offspring, uint32_t block, uint32_t log2n,
                                   uint32_t gens, uint32_t immigrants)
{
    sots_or_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.num_parents = parents, cfg.num_offspring = offspring, cfg.num_dims = sots_or_synth_dims(kind);
    cfg.audio_log2 = log2n, cfg.synth_kind = kind, cfg.recomb_block = block, cfg.gid_base = 77, cfg.seed = 0x5EED0001ull;
    const float pm[4] = {3520.0f, 8.0f, 3520.0f, 1.0f};
    for (uint32_t i = 0; i < cfg.num_dims; ++i) cfg.param_max[i] = kind == SOTS_OR_SYNTH_2OP || kind == SOTS_OR_SYNTH_TRIPLE_PAR ? pm[i & 3] : pm[i & 1];
    const uint32_t p = parents + offspring, d = cfg.num_dims, n = 1u << log2n, w = 2 * d + 1;
    sots_or_es *es = sots_or_es_create(&cfg);
    if (!es) return 0;
    float *table = malloc(sizeof(float) * SOTS_OR_WAVETABLE_SIZE), *audio = malloc(sizeof(float) * n);
    float *vals = malloc(sizeof(float) * d);
    sots_or_wavetable(table);
    for (uint32_t i = 0; i < d; ++i) vals[i] = 0.1f + 0.8f * (float)i / (float)d;
    sots_or_synth(kind, vals, cfg.param_min, cfg.param_max, table, n, audio);
    sots_or_es_set_target_audio(es, audio);
    sots_or_es_init_population(es, 1);
    float *rows = malloc(sizeof(float) * (size_t)(immigrants ? immigrants : 1) * w);
    for (uint32_t g = 0; g < gens; ++g) {
        sots_or_es_generation(es);
        if (immigrants) { /* an island exchanging with itself: its best rows land in its breeding tail */
            sots_or_es_pack_elites(es, rows, immigrants);
            sots_or_es_inject(es, rows, immigrants);
        }
    }
    /* the stages one by one, NaN and infinities in the fitness column, then a full read-back */
    sots_or_es_recombine(es);
    sots_or_es_mutate(es);
    sots_or_es_evaluate(es);
    float *v = malloc(sizeof(float) * (size_t)p * d), *s = malloc(sizeof(float) * (size_t)p * d), *f = malloc(sizeof(float) * p);
    sots_or_es_read_population(es, v, s, f);
    if (p > 3) f[1] = NAN, f[2] = INFINITY, f[3] = -0.0f;
    sots_or_es_write_population(es, v, s, f);
    sots_or_es_sort(es);
    sots_or_es_read_population(es, v, s, f);
    unsigned long long h = 1469598103934665603ull;
    h = fnv(h, v, sizeof(float) * (size_t)p * d);
    h = fnv(h, s, sizeof(float) * (size_t)p * d);
    h = fnv(h, f, sizeof(float) * p);
    h = fnv(h, sots_or_es_audio(es), sizeof(float) * (size_t)p * n);
    h = fnv(h, sots_or_es_spectrum(es), sizeof(float) * (size_t)p * (n / 2));
    free(v), free(s), free(f), free(rows), free(vals), free(audio), free(table);
    sots_or_es_destroy(es);
    return h;
}

int main(void)
{
    /* kind, parents, offspring, block, log2n, generations, immigrants: populations that are not powers of two, parents
     * that are not a multiple of the block, one-row blocks, the shortest and a long transform, every voice */
    const uint32_t cases[][7] = {
        {SOTS_OR_SYNTH_2OP, 32, 32, 32, 10, 3, 0},        {SOTS_OR_SYNTH_2OP, 80, 176, 32, 9, 3, 48},
        {SOTS_OR_SYNTH_2OP, 3, 5, 1, 9, 2, 2},            {SOTS_OR_SYNTH_3OP_SERIES, 16, 16, 32, 9, 2, 0},
        {SOTS_OR_SYNTH_TRIPLE_PAR, 96, 160, 32, 9, 2, 16}, {SOTS_OR_SYNTH_4OP_SERIES, 24, 40, 8, 12, 1, 8},
        {SOTS_OR_SYNTH_2OP, 1, 1, 2, 13, 1, 0},
    };
    for (size_t c = 0; c < sizeof cases / sizeof cases[0]; ++c) {
        const uint32_t *k = cases[c];
        printf("case %zu %016llx\n", c, run_case(k[0], k[1], k[2], k[3], k[4], k[5], k[6]));
    }
    /* the free functions: sort ranks of a column full of ties, the naive DFT against the fast one */
    {
        float fit[37];
        uint32_t perm[37];
        for (int i = 0; i < 37; ++i) fit[i] = (float)((i * 7) % 5);
        fit[5] = NAN, fit[36] = -INFINITY;
        sots_or_sort_perm(fit, 37, perm);
        unsigned long long h = fnv(1469598103934665603ull, perm, sizeof perm);
        const uint32_t n = 512;
        double *win = malloc(sizeof(double) * n), *re = malloc(sizeof(double) * (n / 2 + 1)), *im = malloc(sizeof(double) * (n / 2 + 1));
        double *re2 = malloc(sizeof(double) * (n / 2 + 1)), *im2 = malloc(sizeof(double) * (n / 2 + 1));
        float *a = malloc(sizeof(float) * n), *mag = malloc(sizeof(float) * (n / 2));
        const float wf = sots_or_window(win, n);
        for (uint32_t i = 0; i < n; ++i) a[i] = sinf(0.37f * (float)i) * 0.5f;
        sots_or_rfft(a, n, win, re, im);
        sots_or_rfft_naive(a, n, win, re2, im2);
        double worst = 0.0;
        for (uint32_t k = 0; k <= n / 2; ++k) worst = fmax(worst, fmax(fabs(re[k] - re2[k]), fabs(im[k] - im2[k])));
        sots_or_spectrum(a, n, win, wf, mag);
        h = fnv(h, mag, sizeof(float) * (n / 2));
        printf("free %016llx dft_agree %d fitness %.9g\n", h, worst < 1e-9, (double)sots_or_fitness(mag, mag, n / 2));
        free(win), free(re), free(im), free(re2), free(im2), free(a), free(mag);
    }
    return 0;
}
