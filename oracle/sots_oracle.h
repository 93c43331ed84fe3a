/*
 * sots_oracle.h -- CPU restatement of the reference's per-generation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the HIP library, its
 * C++ host wrapper, the Python binding) may include, link or call this file.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY: pinned, stage by stage, by the reference's OWN DEVICE KERNELS since round 4 - kernels/ocl_program.cl
 * compiled as it stands for gfx950 (oracle/build_ref_ocl.py -> code objects under oracle/_ref), run on an MI355X through the HIP
 * module API, inputs and outputs committed as tests/golden/ocl_ref_v1.npz (tests/test_ocl_reference.py): init,
 * mutation rule (given the kernel's MWC64X words), recombination (the race-free reading of an in-place kernel),
 * the three voices (sots_or_synth_ocl restates the kernels' arithmetic bit for bit; sots_or_synth follows the
 * reference's CPU path - fp32 sample-rate ratio - and stays within table steps of it), window (to the kernel's own fp32
 * cosine error), fitness, sort.  STILL UNPINNED by the reference: its CPU host path (Evolutionary_Strategy.hpp:11 needs
 * fftw_cpp.hh; fftw3, glm, sndfile are absent, nothing was stubbed), the FFT (clFFT / FFTW are third-party and absent:
 * a forward real DFT is fixed mathematically; fp64 FFT against a naive O(N^2) DFT and NumPy), the build-defined 4-op
 * voice (no reference row), and the random streams (the reference seeds from the wall clock).  Further pins: an
 * independent NumPy restatement (tests/golden/make_golden.py), the published Random123 Philox4x32-10 known answers,
 * the self-match property the reference's debug constants imply (ocl_program.cl:247-250).
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef SOTS_ORACLE_H
#define SOTS_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOTS_OR_MAX_DIMS 16
#define SOTS_OR_WAVETABLE_SIZE 32768u /* Evolutionary_Strategy.hpp:197 */
#define SOTS_OR_SAMPLE_RATE 44100u    /* Evolutionary_Strategy.hpp:196 */

/* synth kinds; numDimensions is implied (4, 6, 12, 8) */
enum {
    SOTS_OR_SYNTH_2OP = 0,          /* Evolutionary_Strategy.hpp:368-402 */
    SOTS_OR_SYNTH_3OP_SERIES = 1,   /* Evolutionary_Strategy.hpp:403-449 */
    SOTS_OR_SYNTH_TRIPLE_PAR = 2,   /* Evolutionary_Strategy.hpp:450-495 */
    SOTS_OR_SYNTH_4OP_SERIES = 3    /* build-defined (no reference row), SURVEY.md 8a footnote */
};

/* PRNG domains (4th counter word) */
enum { SOTS_OR_TAG_INIT = 0x494e4954u, SOTS_OR_TAG_MUTATE = 0x4d555441u };

/* ---- counter-based PRNG: Philox4x32-10 (Salmon et al., SC'11) ---- */
void sots_or_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* ---- Objective pieces ---- */
/* Evolutionary_Strategy.hpp:325-332 */
void sots_or_wavetable(float *table /* [32768] */);
/* Evolutionary_Strategy.hpp:308-317; returns fftWindowFactor (float) */
float sots_or_window(double *window, uint32_t n);
/* scaleParams, Evolutionary_Strategy.hpp:567-576 (GPU form min + v*(max-min), ocl_program.cl:297) */
uint32_t sots_or_synth_dims(uint32_t kind);
void sots_or_synth(uint32_t kind, const float *values, const float *pmin, const float *pmax,
                   const float *table, uint32_t n, float *audio /* [n] */);
/* window -> r2c (double) -> hypotf/N/wf ; Evolutionary_Strategy.hpp:503-542. mag has n/2 entries */
void sots_or_spectrum(const float *audio, uint32_t n, const double *window, float window_factor,
                      float *mag);
/* same but returns the complex bins 0..n/2 (double), used to check the HIP FFT stage */
void sots_or_rfft(const float *audio, uint32_t n, const double *window, double *re, double *im);
/* naive O(n^2) DFT of window*audio, bins 0..n/2: cross-check of sots_or_rfft */
void sots_or_rfft_naive(const float *audio, uint32_t n, const double *window, double *re, double *im);
/* Evolutionary_Strategy_CPU.hpp:228-268, k = 0..n/2-1 */
float sots_or_fitness(const float *mag, const float *target, uint32_t half);

/* ---- evolutionary operators (GPU definitions) ---- */
/* ocl_program.cl:46-66 with the counter-based PRNG; chunk = audio chunk index */
void sots_or_init_population(float *values, float *steps, uint32_t p, uint32_t d,
                             uint64_t seed, uint32_t gid_base, uint32_t chunk);
/* ocl_program.cl:73-149, race-free (all reads precede all writes): in -> out */
void sots_or_recombine(const float *vin, const float *sin_, float *vout, float *sout,
                       uint32_t p, uint32_t d, uint32_t num_parents, uint32_t block);
/* the voices with the arithmetic of the reference's OpenCL kernels (ocl_program.cl:280-443; double sample-rate ratio,
 * bare index conversion: `table` holds W + 1 entries); contract bit 0 / 1: fused float / double multiply-adds */
void sots_or_synth_ocl(uint32_t kind, const float *values, const float *pmin, const float *pmax,
                       const float *table, uint32_t n, float *audio, uint32_t contract);
/* the [-1, 1] float of one random word (ocl_program.cl:60, :27) and one gene's mutation from its 13 words */
float sots_or_draw_unit(uint32_t word);
void sots_or_mutate_gene(float *value, float *step, uint32_t d, const uint32_t words[13]);
/* ocl_program.cl:155-190, in place */
void sots_or_mutate(float *values, float *steps, uint32_t p, uint32_t d,
                    uint64_t seed, uint32_t gid_base, uint32_t generation);
/* ascending by fitness, stable (Evolutionary_Strategy.hpp:108-124), NaN last. perm[r] = source row */
void sots_or_sort_perm(const float *fitness, uint32_t p, uint32_t *perm);

/* ---- whole strategy (Evolutionary_Strategy_CPU.hpp:353-469 stage order) ---- */
typedef struct sots_or_config {
    uint32_t num_parents, num_offspring, num_dims, audio_log2;
    uint32_t synth_kind, recomb_block, gid_base, reserved;
    uint64_t seed;
    float param_min[SOTS_OR_MAX_DIMS];
    float param_max[SOTS_OR_MAX_DIMS];
} sots_or_config;

typedef struct sots_or_es sots_or_es;
sots_or_es *sots_or_es_create(const sots_or_config *cfg);
void sots_or_es_destroy(sots_or_es *es);
void sots_or_es_set_target_audio(sots_or_es *es, const float *audio);
void sots_or_es_set_target_spectrum(sots_or_es *es, const float *mag);
void sots_or_es_init_population(sots_or_es *es, uint32_t chunk);
void sots_or_es_write_population(sots_or_es *es, const float *values, const float *steps, const float *fitness);
void sots_or_es_read_population(const sots_or_es *es, float *values, float *steps, float *fitness);
void sots_or_es_set_generation(sots_or_es *es, uint32_t generation);
/* individual stages on the current state */
void sots_or_es_recombine(sots_or_es *es);
void sots_or_es_mutate(sots_or_es *es);
void sots_or_set_threads(int n); /* threads of sots_or_es_evaluate; default 1 */
void sots_or_es_evaluate(sots_or_es *es); /* synth + window + fft + fitness */
void sots_or_es_sort(sots_or_es *es);
void sots_or_es_generation(sots_or_es *es);
/* replace the last n_rows parent rows by immigrant rows [fitness, v0..vD-1, s0..sD-1] */
void sots_or_es_inject(sots_or_es *es, const float *rows, uint32_t n_rows);
void sots_or_es_pack_elites(const sots_or_es *es, float *rows, uint32_t n_rows);
const float *sots_or_es_audio(const sots_or_es *es);    /* [P][N] unwindowed */
const float *sots_or_es_spectrum(const sots_or_es *es); /* [P][N/2] magnitudes */

#ifdef __cplusplus
}
#endif
#endif
