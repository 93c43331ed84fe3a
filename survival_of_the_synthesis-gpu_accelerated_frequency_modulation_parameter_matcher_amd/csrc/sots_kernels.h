// sots_kernels.h -- host-callable launchers of the gfx950 kernels (sots_kernels.hip).
// Internal to libsots_hip.so; the public boundary is include/sots_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/sots_hip.h"

namespace sots {

constexpr uint32_t kWavetableSize = SOTS_WAVETABLE_SIZE;
constexpr uint32_t kTagInit = 0x494e4954u;   // PRNG domain words (4th Philox counter word)
constexpr uint32_t kTagMutate = 0x4d555441u;

// ES constants, Evolutionary_Strategy.hpp:611-627; the two pow(Ek, beta) values are
// evaluated once on the host (Ek only ever takes two values, ocl_program.cl:168,185).
struct MutateConsts {
    float alpha, one_over_alpha, root_two_over_pi, beta_scale;
    float pow_alpha_beta, pow_inv_alpha_beta;
};

struct SynthParams {
    float pmin[SOTS_MAX_DIMS];
    float pmax[SOTS_MAX_DIMS];
};

// Population geometry shared by most launches.
struct PopDims {
    uint32_t p;            // populationLength
    uint32_t d;            // numDimensions
    uint32_t num_parents;
    uint32_t block;        // recombination block (reference WRKGRPSIZE)
    uint32_t gid_base;
    uint32_t seed_lo, seed_hi;
    // recombine_source's divisions as shifts and masks where the sizes allow (they usually do: blocks of 32, 512 parent blocks)
    uint32_t block_shift;  // log2(block) when block is a power of two, else kNoPow2
    uint32_t npb;          // parent blocks: max(1, num_parents / block)
    uint32_t npb_mask;     // npb - 1 when npb is a power of two, else kNoPow2
};
constexpr uint32_t kNoPow2 = 0xFFFFFFFFu;
inline PopDims make_pop_dims(uint32_t p, uint32_t d, uint32_t num_parents, uint32_t block, uint32_t gid_base, uint32_t seed_lo,
                             uint32_t seed_hi)
{
    PopDims pd{p, d, num_parents, block, gid_base, seed_lo, seed_hi, kNoPow2, 1u, kNoPow2};
    if (block && (block & (block - 1u)) == 0u)
        for (uint32_t sh = 0; sh < 32; ++sh)
            if ((1u << sh) == block) pd.block_shift = sh;
    pd.npb = block && num_parents / block ? num_parents / block : 1u;
    if ((pd.npb & (pd.npb - 1u)) == 0u) pd.npb_mask = pd.npb - 1u;
    return pd;
}

// Variation folded into the synthesis kernel (fused generation loop): when vin != nullptr every
// lane first builds its individual - recombination source rows of the current half, mutation -
// writes it to the other half and synthesises from it; with vin == nullptr the kernel reads
// `values` as is.
struct Variation {
    const float *vin, *sin; // current half (sorted)
    float *vout, *sout;     // other half
    PopDims pd;
    MutateConsts mc;
    uint32_t generation;
};

// Resident-workgroup counts of the persistent spectral kernels (occupancy queries), cached per
// CONTEXT: a context belongs to one device and is used from one thread at a time, so the cache
// needs no synchronisation (a process-wide cache would be shared by every device and thread).
struct OccCache {
    int fft[16], fitness[16], fused_win[16], fused_raw[16];
    int x_fft[16], x_fitness[16], x_fused_win[16], x_fused_raw[16], x_small[16];
    int wide[4]; // k_fft with twelve wavefronts per workgroup: spectrum writer, fused with / without window
    // device memory of the context: the fused long-row kernel's tables as they lie in LDS (launch_x_tables; null = the workgroups make them)
    const float *x_image;
};

// ---- variation ----
hipError_t launch_init_population(hipStream_t st, float *values, float *steps, float *fitness,
                                  const PopDims &pd, uint32_t chunk);
hipError_t launch_recombine(hipStream_t st, const float *vin, const float *sin, float *vout, float *sout,
                            const PopDims &pd);
hipError_t launch_mutate(hipStream_t st, float *values, float *steps, const PopDims &pd,
                         const MutateConsts &mc, uint32_t generation);
// recombine (in -> out) and mutate fused
hipError_t launch_recombine_mutate(hipStream_t st, const float *vin, const float *sin, float *vout,
                                   float *sout, const PopDims &pd, const MutateConsts &mc,
                                   uint32_t generation);

// ---- evaluation ----
// Small populations of the series voices: the synthesis kernel with the time axis in the lanes applies (it makes its own
// individuals from one thread per gene when given a Variation, whatever the number of genes)
bool synth_time_parallel(uint32_t kind, uint32_t p, uint32_t num_cus);
// ... and where the series voices run with the operators in the lanes (k_synth_ol: the workgroup makes its individuals too, a thread per gene)
bool synth_operators_in_lanes(uint32_t kind, uint32_t p, uint32_t num_cus);
// Audio rows are `pitch` floats apart (pitch >= N, a multiple of 4): a power-of-two row stride
// would put every lane of a row-per-lane store on the same memory channel.
// the voices with the arithmetic of the reference's device kernels (ocl_program.cl:280-443); no 4-op voice
hipError_t launch_synth_device_arith(hipStream_t st, uint32_t kind, const float *values, const float *wavetable, float *audio,
                                     const SynthParams &sp, uint32_t p, uint32_t log2n, uint32_t pitch);
hipError_t launch_synth(hipStream_t st, uint32_t kind, const float *values, const float *wavetable,
                        float *audio, const SynthParams &sp, uint32_t p, uint32_t log2n, uint32_t pitch,
                        uint32_t num_cus, const Variation *var = nullptr, bool allow_cut = true);
hipError_t launch_window(hipStream_t st, float *audio, const float *window, uint32_t p, uint32_t log2n,
                         uint32_t pitch);
// audio[P][pitch] -> spectrum[P][N+8]
hipError_t launch_fft(hipStream_t st, const float *audio, float *spectrum, const float2 *twiddle,
                      uint32_t p, uint32_t log2n, uint32_t pitch, uint32_t num_cus, OccCache *occ);
// spectrum[P][N+8] x target[N/2] -> fitness[P]
hipError_t launch_fitness(hipStream_t st, const float *spectrum, const float *target, float *fitness,
                          uint32_t p, uint32_t log2n, float inv_n, float inv_wf, uint32_t num_cus, OccCache *occ);
// audio[P][pitch] (x window when window != nullptr) x target -> fitness[P]; no spectrum in memory
size_t x_table_bytes(uint32_t log2n);
hipError_t launch_x_tables(hipStream_t st, float *image, const float2 *twiddle, const float *window, const float *target, uint32_t log2n);
hipError_t launch_fft_fitness(hipStream_t st, const float *audio, const float *window, const float *target,
                              float *fitness, const float2 *twiddle, uint32_t p, uint32_t log2n, uint32_t pitch,
                              float inv_n, float inv_wf, uint32_t num_cus, OccCache *occ);

// Island exchange folded into sortPopulation (one generation of the fused loop, set through
// sots_fuse_exchange_next_sort): the kernel that moves the sorted rows also
//   * takes destination rows [imm_first, imm_first + imm_rows) from `imm` (rows [fitness, values.., steps..] of an
//     all-gathered buffer, source rows [skip_first, skip_first + skip_count) - the island's own block - passed
//     over) instead of from the local population: sots_inject_gathered_device without its launch;
//   * copies the best `sink_rows` rows - immigrants included where the ranges overlap - to `sink` in the same row
//     format: sots_pack_elites_device without its launch.
// Either pointer may be null.
struct SortExchange {
    float *sink;
    const float *imm;
    uint32_t sink_rows, imm_first, imm_rows, skip_first, skip_count;
};

// ---- selection ----
// keys: sort_keys_bytes(P) bytes; scratch: sort_scratch_bytes(P) bytes.  Rows that would land in front of
// `first_row` are not written (the selection below has placed them already).
size_t sort_scratch_bytes(uint32_t p);
size_t sort_keys_bytes(uint32_t p);
hipError_t launch_sort(hipStream_t st, const float *vin, const float *sin, const float *fin,
                       float *vout, float *sout, float *fout, uint64_t *keys, void *scratch, uint32_t p,
                       uint32_t d, uint32_t first_row = 0, const SortExchange *exchange = nullptr);
// the best `need` rows in order into rows 0..need-1 of the out arrays, other rows untouched; only
// where select_applies() (otherwise hipErrorInvalidValue)
bool select_applies(uint32_t p, uint32_t need);
size_t select_scratch_bytes(uint32_t p); // the sort scratch must hold at least this much
hipError_t launch_select(hipStream_t st, const float *vin, const float *sin, const float *fin, float *vout,
                         float *sout, float *fout, uint64_t *keys, void *scratch, uint32_t p, uint32_t d, uint32_t need,
                         uint32_t num_cus, const SortExchange *exchange = nullptr);

// ---- island exchange ----
hipError_t launch_pack_rows(hipStream_t st, const float *values, const float *steps, const float *fitness,
                            float *rows, uint32_t first_row, uint32_t n_rows, uint32_t d);
// source rows [skip_first, skip_first + skip_count) are passed over
hipError_t launch_unpack_rows(hipStream_t st, float *values, float *steps, float *fitness,
                              const float *rows, uint32_t first_row, uint32_t n_rows, uint32_t d,
                              uint32_t skip_first, uint32_t skip_count);

uint32_t next_pow2(uint32_t v);

} // namespace sots
