/*
 * sots_hip.h -- C-ABI of libsots_hip.so, the MI355X (gfx950) backend for the
 * per-generation evolutionary FM sound-matching loop.
 *
 * This is the drop-in boundary: Evolutionary_Strategy_HIP (C++, host/) and the
 * ctypes binding (capi.py) sit on top of exactly these entry points.  Plain
 * pointers and sizes only; no C++ or torch types cross it.  One context is used
 * from one host thread at a time (the reference's objects are single-threaded
 * too).  Every call returns SOTS_OK (0) or a negative code;
 * sots_last_error() gives the text.  All file:line citations are relative to the
 * reference tree.
 *
 * Population state, as in the reference's device buffer set
 * (Evolutionary_Strategy_OpenCL.hpp:60-63,278-292):
 *   value, step : float[2][P][D]   two rotation halves, one row per individual
 *   fitness     : float[2][P]
 *   audio       : float[P][N]
 *   spectrum    : float[P][N+8]    interleaved complex, N/2+4 bins per row,
 *                                  bins 0..N/2 valid (clFFT layout, :164-168)
 *   target      : float[N/2]
 * P = numParents + numOffspring, D = numDimensions, N = 2^audioLengthLog2.
 */
#ifndef SOTS_HIP_H
#define SOTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOTS_MAX_DIMS 16
#define SOTS_WAVETABLE_SIZE 32768u /* Evolutionary_Strategy.hpp:197 */
#define SOTS_SAMPLE_RATE 44100u    /* Evolutionary_Strategy.hpp:196 */

enum sots_status {
    SOTS_OK = 0,
    SOTS_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
    SOTS_ERR_HIP = -2,       /* a HIP runtime call failed */
    SOTS_ERR_NO_DEVICE = -3, /* no usable gfx950 device */
    SOTS_ERR_SIZE = -4,      /* caller buffer too small / wrong byte count */
    SOTS_ERR_STATE = -5      /* call out of order (e.g. no target set) */
};

/* Which FM voice synthesisePopulation runs; numDimensions must match. */
enum sots_synth_kind {
    SOTS_SYNTH_2OP = 0,        /* D=4,  ocl_program.cl:280-330, Evolutionary_Strategy.hpp:368-402 */
    SOTS_SYNTH_3OP_SERIES = 1, /* D=6,  ocl_program.cl:332-386, Evolutionary_Strategy.hpp:403-449 */
    SOTS_SYNTH_TRIPLE_PAR = 2, /* D=12, ocl_program.cl:388-443, Evolutionary_Strategy.hpp:450-495 */
    SOTS_SYNTH_4OP_SERIES = 3  /* D=8,  build-defined 4-operator series chain (BASELINE config 4) */
};

/* Stage ids, in the order of the reference's kernelNames_
 * (Evolutionary_Strategy_OpenCL.hpp:54,117). */
enum sots_stage {
    SOTS_STAGE_INIT = 0,
    SOTS_STAGE_RECOMBINE = 1,
    SOTS_STAGE_MUTATE = 2,
    SOTS_STAGE_SYNTHESISE = 3,
    SOTS_STAGE_WINDOW = 4,
    SOTS_STAGE_FFT = 5,
    SOTS_STAGE_FITNESS = 6,
    SOTS_STAGE_SORT = 7,
    SOTS_STAGE_ROTATE = 8,
    /* kernels of the fused generation loop (sots_execute_generations) */
    SOTS_STAGE_FUSED_VARIATION = 9, /* recombine + mutate */
    SOTS_STAGE_FUSED_SYNTH = 10,    /* synthesise */
    SOTS_STAGE_FUSED_SPECTRAL = 11, /* window + FFT + fitness */
    SOTS_STAGE_SORT_TAIL = 12,      /* the rest of the order after a selection, produced when it is read */
    SOTS_STAGE_COUNT = 13
};

/* What sortPopulation delivers inside sots_execute_generations (and sots_stage_select).
 * The next generation's recombinePopulation reads whole parent blocks only (ocl_program.cl:99-112),
 * i.e. rows 0..S-1 of the sorted half, S = max(numParents, max(1, numParents/workgroupSize)*workgroupSize). */
enum sots_sort_mode {
    SOTS_SORT_LAZY_TAIL = 0, /* default: each generation places rows 0..S-1, in order and bit-identical to the full
                              * sort; rows S..P-1 are produced, from the still intact unsorted half, by the first
                              * call that looks at them (read/write population, any sots_stage_*, packing more
                              * than S elites).  A population outside 1024 < P <= 131072 or with S > P/2 is sorted in full.
                              * Calls that WRITE rows (stages, write_population, init) end the pending state. */
    SOTS_SORT_FULL = 1,      /* the reference's behaviour: every generation sorts all P rows
                              * (ocl_program.cl:664-711, Evolutionary_Strategy.hpp:108-124) */
    SOTS_SORT_TOP_ONLY = 2   /* like 0 but rows S..P-1 are never produced (their content is unspecified; packing more
                              * than S elites fails with SOTS_ERR_STATE) */
};

/* Which of the reference's two arithmetics synthesisePopulation uses.  Its CPU path (Evolutionary_Strategy.hpp:203,368-495)
 * keeps the sample-rate ratio in fp32; its device kernels (ocl_program.cl:280-443) write it as a double expression, fuse
 * their multiply-adds and, in the 3-op voice, add params[4] where the CPU path adds params[5]: the same parameters give
 * audio a few wavetable steps apart.  north_star names the CPU path as the parity target: the default,
 * and what every tuned kernel computes. */
enum sots_synth_arith {
    SOTS_ARITH_CPU_PATH = 0,
    SOTS_ARITH_DEVICE_KERNELS = 1 /* bit-identical to the reference's OpenCL kernels as compiled for this GPU
                                   * (tests/test_ocl_reference.py); one plain kernel, not tuned; not for the 4-op voice */
};

/* Replaces Evolutionary_Strategy_OpenCL_Arguments
 * (Evolutionary_Strategy_OpenCL.hpp:25-38) + Evolutionary_Strategy_Arguments
 * (Evolutionary_Strategy.hpp:579-589). */
typedef struct sots_config {
    uint32_t struct_size;       /* = sizeof(sots_config) */
    uint32_t num_parents;       /* es_args.pop.numParents */
    uint32_t num_offspring;     /* es_args.pop.numOffspring */
    uint32_t num_dimensions;    /* es_args.pop.numDimensions */
    uint32_t audio_length_log2; /* es_args.audioLengthLog2, 8..15 (N = 256 ... 32768; main.cpp:90 takes any) */
    uint32_t num_generations;   /* es_args.numGenerations */
    uint32_t synth_kind;        /* enum sots_synth_kind */
    uint32_t workgroup_size;    /* workgroupX: the recombination block (WRKGRPSIZE in ocl_program.cl:86-148) */
    int32_t device;             /* HIP device ordinal (replaces deviceType) */
    uint32_t gid_base;          /* global id of individual 0 (island offset for the PRNG) */
    uint64_t seed;              /* PRNG key (replaces the wall-clock seed, ...OpenCL.hpp:383) */
    float param_min[SOTS_MAX_DIMS]; /* es_args.paramMin */
    float param_max[SOTS_MAX_DIMS]; /* es_args.paramMax */
} sots_config;

typedef struct sots_ctx sots_ctx;

/* ---- lifetime (replaces Evolutionary_Strategy_OpenCL::init, ...OpenCL.hpp:138-150) ---- */
int sots_create(const sots_config *cfg, sots_ctx **out);
void sots_destroy(sots_ctx *ctx);
/* text of the last failure on ctx (or of the last failed sots_create when ctx == NULL) */
const char *sots_last_error(const sots_ctx *ctx);
/* run every later launch/copy on this hipStream_t (NULL = the context's own stream) */
int sots_set_stream(sots_ctx *ctx, void *hip_stream);
int sots_synchronize(sots_ctx *ctx);

/* ---- target (replaces setTargetAudio, ...OpenCL.hpp:563-570; Objective::calculateFFT,
 *      Evolutionary_Strategy.hpp:524-542) ---- */
int sots_set_target_audio(sots_ctx *ctx, const float *audio, uint32_t num_samples);
int sots_set_target_spectrum(sots_ctx *ctx, const float *magnitudes, uint32_t num_bins);

/* ---- population (initPopulationCL ...OpenCL.hpp:369-378; write/readPopulationData :403-430) ---- */
int sots_init_population(sots_ctx *ctx, uint32_t chunk_index);
/* any pointer may be NULL; byte counts must equal P*D*4 (values, steps) and P*4 (fitness).
 * Reads and writes address the CURRENT rotation half. */
int sots_write_population(sots_ctx *ctx, const float *values, size_t values_bytes,
                          const float *steps, size_t steps_bytes,
                          const float *fitness, size_t fitness_bytes);
int sots_read_population(sots_ctx *ctx, float *values, size_t values_bytes,
                         float *steps, size_t steps_bytes,
                         float *fitness, size_t fitness_bytes);
/* the other rotation half (the reference's "output" arrays, main.cpp:241) */
int sots_read_population_other(sots_ctx *ctx, float *values, size_t values_bytes,
                               float *steps, size_t steps_bytes,
                               float *fitness, size_t fitness_bytes);

/* ---- synthesiser buffers (write/readSynthesizerData, ...OpenCL.hpp:436-453) ----
 * audio: P*N*4 bytes; spectrum: P*(N+8)*4 bytes; target: (N/2)*4 bytes; NULL skips. */
int sots_write_synth(sots_ctx *ctx, const float *audio, size_t audio_bytes,
                     const float *spectrum, size_t spectrum_bytes);
int sots_read_synth(sots_ctx *ctx, float *audio, size_t audio_bytes,
                    float *spectrum, size_t spectrum_bytes,
                    float *target, size_t target_bytes);

/* ---- the nine stages, one launch sequence each (executeGeneration, ...OpenCL.hpp:471-541) ---- */
int sots_stage_recombine(sots_ctx *ctx);
int sots_stage_mutate(sots_ctx *ctx);
int sots_stage_synthesise(sots_ctx *ctx);
int sots_stage_window(sots_ctx *ctx);
int sots_stage_fft(sots_ctx *ctx);
int sots_stage_fitness(sots_ctx *ctx);
int sots_stage_sort(sots_ctx *ctx);
/* the fused loop's sortPopulation as a stage: rows 0..S-1 only (enum sots_sort_mode); must be followed by
 * sots_stage_rotate.  Falls back to sots_stage_sort where the selection does not apply. */
int sots_stage_select(sots_ctx *ctx);
int sots_stage_rotate(sots_ctx *ctx);

/* stage-separated generation: the eight stages above in reference order */
int sots_execute_generation(sots_ctx *ctx);
/* n generations of the fused loop (recombine+mutate | synthesise | window+FFT+fitness |
 * sort | rotate); bit-identical population results to n x sots_execute_generation
 * (sortPopulation places the rows the next generation reads and leaves the rest of the order to
 * the first reader, enum sots_sort_mode).
 * The window is applied as the FFT kernel loads a row, so afterwards the audio buffer holds
 * the UN-windowed synthesis and the spectrum buffer is untouched.
 * Only enqueues - with ONE exception: after sots_fuse_exchange_next_sort with a host_gate_event the call blocks the
 * calling thread on that event (hipEventSynchronize) before it enqueues the last generation's sort, so the host never
 * runs more than the kernels of one generation ahead of a gathered exchange.
 * (executeAllGenerations, ...OpenCL.hpp:542-547) */
int sots_execute_generations(sots_ctx *ctx, uint32_t n);

int sots_set_sort_mode(sots_ctx *ctx, uint32_t mode); /* enum sots_sort_mode */
/* enum sots_synth_arith; applies to sots_stage_synthesise and to both generation loops from the next call on
 * (replaces nothing: the reference picks its arithmetic by picking a backend, main.cpp:105-163) */
int sots_set_synth_arithmetic(sots_ctx *ctx, uint32_t arith);
int sots_get_generation(const sots_ctx *ctx, uint32_t *generation);
int sots_set_generation(sots_ctx *ctx, uint32_t generation);

/* ---- per-stage device timing (feeds Benchmarker::addTimer, Benchmarker.hpp:109-130) ---- */
int sots_timing_enable(sots_ctx *ctx, int enabled);
int sots_timing_reset(sots_ctx *ctx);
/* sum of hipEvent-measured durations and the number of launches of that stage
 * since the last reset; synchronises the stream */
int sots_stage_time_ms(sots_ctx *ctx, int stage, double *total_ms, uint64_t *count);
/* the individual launch durations behind that sum, oldest first (at most 65536 are kept per stage
 * between resets): one Benchmarker::addTimer(name, ms) per launch gives the CSV the reference's
 * per-launch Average/Max/Min columns (Benchmarker.hpp:33-72,109-130).  *written <= capacity. */
int sots_stage_launch_times_ms(sots_ctx *ctx, int stage, float *out_ms, uint64_t capacity, uint64_t *written);

/* ---- island model (new) ----
 * A row is [fitness, v0..v(D-1), s0..s(D-1)] = (2D+1) floats.  pack copies the best
 * n_rows rows of the current (sorted) half; inject overwrites the last n_rows of the
 * PARENT rows that recombination reads - whole blocks of workgroupSize rows:
 * B = max(1, numParents / workgroupSize) * workgroupSize, rows B-n_rows .. B-1
 * (= numParents-n_rows .. numParents-1 when numParents is a multiple of the block,
 * ocl_program.cl:99-112) - so that immigrants take part in the next recombination.  *_device take device pointers on this context's
 * device and run on its stream (no host sync); *_host are blocking. */
int sots_pack_elites_device(sots_ctx *ctx, void *device_rows, uint32_t n_rows);
int sots_inject_immigrants_device(sots_ctx *ctx, const void *device_rows, uint32_t n_rows);
/* gathered_rows = the all-gather result, world x elites rows in rank order: injects every
 * island's rows except this rank's own block (one launch, no intermediate copy) */
int sots_inject_gathered_device(sots_ctx *ctx, const void *gathered_rows, uint32_t world, uint32_t rank,
                                uint32_t elites);
/* The same exchange WITHOUT its two launches: the sortPopulation of the LAST generation of the next
 * sots_execute_generations call also (a) takes the immigrant rows straight from gathered_rows (as
 * sots_inject_gathered_device would after that sort) and (b) writes the best n_elite_rows rows of the result, immigrants
 * included where the ranges overlap, to elite_rows (as sots_pack_elites_device would after the inject).  Either
 * pointer may be NULL; both are device pointers that must be ready when that sort runs on the context's stream and
 * stay valid until it has.  Used once, then forgotten; sots_init_population forgets it too.  Where sortPopulation
 * places only the rows recombination reads (enum sots_sort_mode), n_elite_rows may not exceed them.
 * host_gate_event (optional, a hipEvent_t): for gathered_rows filled by a collective on ANOTHER stream.  The host waits
 * for the event (hipEventSynchronize) right before it enqueues that sort - the generation's variation, synthesis and
 * spectral kernels are on the stream by then, so the device stays busy - instead of the stream waiting for it: on this
 * runtime a cross-stream hipStreamWaitEvent costs the waiting stream ~18 us per generation even for an event that
 * completed long ago.  With NULL the caller orders the rows on the context's stream itself. */
int sots_fuse_exchange_next_sort(sots_ctx *ctx, void *elite_rows, uint32_t n_elite_rows, const void *gathered_rows,
                                 uint32_t world, uint32_t rank, uint32_t elites, void *host_gate_event);
int sots_pack_elites_host(sots_ctx *ctx, float *rows, uint32_t n_rows);
int sots_inject_immigrants_host(sots_ctx *ctx, const float *rows, uint32_t n_rows);

/* ---- island group: one process, one island per listed device (new) ----
 * The reference has one device per process (Evolutionary_Strategy_OpenCL.hpp:194-226).  A group owns one
 * sots_ctx per entry of `devices` (island i: device devices[i], PRNG ids gid_base + i * P, so an island's
 * random stream does not depend on the group size) and runs them from one host thread each.  Every
 * `migration_interval` generations each island's best `num_elites` rows are all-gathered - RCCL
 * (ncclCommInitAll + ncclAllGather on the islands' streams, librccl opened on first use) when the devices
 * are distinct, device-to-device copies ordered by HIP events when islands share a device - and the other
 * islands' rows overwrite the tail of the rows recombination reads (sots_inject_gathered_device).
 * With one device the group is that one island and exchanges nothing. */
#define SOTS_MAX_GROUP_DEVICES 16
enum sots_group_flags {
    SOTS_GROUP_OVERLAP = 1,    /* the all-gather started after generation g runs on a side stream underneath
                                * generation g+1 and is injected after g+1's sort (rows arrive one exchange later) */
    SOTS_GROUP_FORCE_RCCL = 2, /* use RCCL even for a single island (a one-rank communicator; exercises the
                                * collective path on a one-GPU machine) */
    SOTS_GROUP_EVENT_WAITS = 8,/* overlapped schedule: the island's STREAM waits for the side stream's events instead of its host
                                * thread (sots_fuse_exchange_next_sort, host_gate_event): same results, ~18 us per
                                * generation slower on this runtime; for tests and timing comparisons */
    SOTS_GROUP_UNFUSED = 4     /* pack and inject as launches of their own (sots_pack_elites_device,
                                * sots_inject_gathered_device) instead of inside the sort kernels
                                * (sots_fuse_exchange_next_sort): same results, for tests and timing comparisons */
};
typedef struct sots_group sots_group;
int sots_group_create(const sots_config *island_cfg, const int32_t *devices, uint32_t num_devices,
                      uint32_t num_elites, uint32_t migration_interval, uint32_t flags, sots_group **out);
void sots_group_destroy(sots_group *group);
/* text of the last failure on the group (or of the last failed sots_group_create when group == NULL) */
const char *sots_group_last_error(const sots_group *group);
uint32_t sots_group_size(const sots_group *group);
int sots_group_uses_rccl(const sots_group *group);
/* island i's context: read its population, timers, info; do not destroy it or change its stream */
sots_ctx *sots_group_island(sots_group *group, uint32_t i);
int sots_group_set_target_audio(sots_group *group, const float *audio, uint32_t num_samples);
int sots_group_set_target_spectrum(sots_group *group, const float *magnitudes, uint32_t num_bins);
int sots_group_init_population(sots_group *group, uint32_t chunk_index);
/* n generations on every island with the elite exchange; returns once everything is ENQUEUED.  In the overlapped
 * schedule (SOTS_GROUP_OVERLAP without SOTS_GROUP_EVENT_WAITS, "host-gated") every island's thread waits, before it
 * enqueues the sort of a generation in which an exchange falls due, for the PREVIOUS exchange's all-gather to have
 * completed on the device: the call can block for up to that long, and the host runs at most one exchange interval
 * ahead of the devices.  This schedule over RCCL has run on a one-rank communicator only (no machine with two GPUs has
 * been available to the tests); SOTS_GROUP_EVENT_WAITS is the stream-ordered fallback.
 * After an error the islands' populations are unspecified (a failed island stops running generations, the others go on
 * and may receive its last good - or +inf-fitness - rows). */
int sots_group_execute_generations(sots_group *group, uint32_t n);
int sots_group_synchronize(sots_group *group);
/* the island holding the lowest fitness and that fitness (blocking) */
int sots_group_best(sots_group *group, uint32_t *island, float *fitness);

/* ---- introspection ---- */
typedef struct sots_info {
    uint32_t population_length, num_dimensions, audio_length, spectrum_row_floats;
    uint32_t rotation_index, generation, compute_units, reserved;
    char device_name[128];
    char arch[32];
} sots_info;
int sots_get_info(const sots_ctx *ctx, sots_info *info);

#ifdef __cplusplus
}
#endif
#endif /* SOTS_HIP_H */
