// sots_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the per-generation
// evolutionary FM sound-matching loop.  Compiled with -ffp-contract=off: every fp32
// expression rounds exactly as written, which is what makes synthesis, recombination and
// the value half of mutation bit-identical to the CPU restatement.
//
// Layout choices (DESIGN.md has the reasoning):
//   * variation kernels        : one lane per gene, coalesced [P][D] rows
//   * synthesisePopulation     : one lane per individual (the phase recurrence is serial in
//                                fp32), 128 KiB wavetable staged in LDS, 16-byte row stores
//   * FFT / fitness            : one individual per wavefront, Stockham radix-8/4 passes
//                                (three/two radix-2 layers kept in registers), padded LDS
//                                exchange, xor-shuffle reduction of the squared error
//   * sortPopulation           : bitonic network on (fitness, index) 64-bit keys, LDS tiles
//
// Reference citations are file:line in the reference tree.
#include "sots_kernels.h"
#include <type_traits>

#include <cstdlib>

namespace sots {

uint32_t next_pow2(uint32_t v)
{
    uint32_t r = 1;
    while (r < v) r <<= 1;
    return r;
}

namespace {

constexpr int kWave = 64;

// ------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011).  Replaces MWC64X (ocl_program.cl:5-16): the state
// is the counter (global individual id, epoch, block, domain) under the key (seed).
// ------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64-bit multiply per product (v_mad_u64_u32) instead of a high and a low one: integer multiplies
        // run at a quarter of the vector rate and are most of what a gene costs
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

__device__ __forceinline__ uint32_t u4_at(const U4 &v, uint32_t i)
{
    return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

// (float)((int)MWC64X) / 2147483647.0f, ocl_program.cl:27,61
__device__ __forceinline__ float draw_unit(uint32_t w)
{
    return (float)((int32_t)w) / 2147483647.0f;
}

// ------------------------------------------------------------------------------------
// initPopulation, ocl_program.cl:46-66
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_population(float *__restrict__ values,
                                                         float *__restrict__ steps,
                                                         float *__restrict__ fitness, PopDims pd,
                                                         uint32_t chunk)
{
    const uint32_t total = pd.p * pd.d;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const uint32_t i = e / pd.d, g = e - i * pd.d;
        const U4 r = philox4x32_10(pd.gid_base + i, chunk, g >> 2, kTagInit, pd.seed_lo, pd.seed_hi);
        const float u = draw_unit(u4_at(r, g & 3u));
        steps[e] = 0.1f;
        values[e] = (u < 0.0f) ? -u : u;
        if (g == 0) fitness[i] = 0.0f;
    }
}

// ------------------------------------------------------------------------------------
// recombinePopulation, ocl_program.cl:73-149.  Block b copies parent block b % NPB and
// moves gene g of local individual l to local individual (l + g*(b+1)) mod B (:130-137).
// Out of place (current half -> other half): the reference's in-place version lets
// offspring blocks read parent rows that parent blocks are overwriting (:104-147).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t recombine_source(uint32_t i, uint32_t g, const PopDims &pd)
{
    // (g < 16 and b + 1 <= 2^26, so g (b + 1) fits 32 bits; the branches are uniform: pd is a kernel argument)
    uint32_t b, l;
    if (pd.block_shift != kNoPow2) {
        const uint32_t mask = pd.block - 1u;
        b = i >> pd.block_shift;
        l = ((i & mask) - g * (b + 1u)) & mask;
    } else {
        b = i / pd.block;
        const uint32_t dl = i - b * pd.block, shift = (g * (b + 1u)) % pd.block;
        l = (dl + pd.block - shift) % pd.block;
    }
    const uint32_t pb = pd.npb_mask != kNoPow2 ? (b & pd.npb_mask) : b % pd.npb;
    return (pb * pd.block + l) * pd.d + g;
}

__global__ __launch_bounds__(256) void k_recombine(const float *__restrict__ vin,
                                                   const float *__restrict__ sin,
                                                   float *__restrict__ vout,
                                                   float *__restrict__ sout, PopDims pd)
{
    const uint32_t total = pd.p * pd.d;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const uint32_t i = e / pd.d, g = e - i * pd.d;
        const uint32_t src = recombine_source(i, g, pd);
        vout[e] = vin[src];
        sout[e] = sin[src];
    }
}

// ------------------------------------------------------------------------------------
// mutatePopulation, ocl_program.cl:155-190.  13 draws per gene = Philox blocks 4g..4g+3.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void mutate_gene(float &x, float &s, uint32_t gid, uint32_t g,
                                            uint32_t generation, const PopDims &pd,
                                            const MutateConsts &mc)
{
    const U4 r0 = philox4x32_10(gid, generation, 4u * g + 0u, kTagMutate, pd.seed_lo, pd.seed_hi);
    const U4 r1 = philox4x32_10(gid, generation, 4u * g + 1u, kTagMutate, pd.seed_lo, pd.seed_hi);
    const U4 r2 = philox4x32_10(gid, generation, 4u * g + 2u, kTagMutate, pd.seed_lo, pd.seed_hi);
    const U4 r3 = philox4x32_10(gid, generation, 4u * g + 3u, kTagMutate, pd.seed_lo, pd.seed_hi);
    const bool even = (r0.x % 2u) == 0u;
    const float ek = even ? mc.alpha : mc.one_over_alpha;             // :168
    const float pw = even ? mc.pow_alpha_beta : mc.pow_inv_alpha_beta; // pow(Ek, BETA), :185
    float sum = 0.0f;                                                  // gauss_rand, :21-31
    sum += draw_unit(r0.y); sum += draw_unit(r0.z); sum += draw_unit(r0.w);
    sum += draw_unit(r1.x); sum += draw_unit(r1.y); sum += draw_unit(r1.z); sum += draw_unit(r1.w);
    sum += draw_unit(r2.x); sum += draw_unit(r2.y); sum += draw_unit(r2.z); sum += draw_unit(r2.w);
    sum += draw_unit(r3.x);
    sum /= 12.0f;
    float gauss = sum;
    float new_x = x + ek * s * gauss;                                  // :174
    if (new_x < 0.0f || new_x > 1.0f) {                                // :176-182
        gauss = gauss * -0.5f;
        new_x = x + ek * s * gauss;
    }
    const float es = expf(fabsf(gauss) - mc.root_two_over_pi);         // :184
    s *= pw * powf(es, mc.beta_scale);                                 // :185
    x = new_x;
}

__global__ __launch_bounds__(256) void k_mutate(float *__restrict__ values, float *__restrict__ steps,
                                                PopDims pd, MutateConsts mc, uint32_t generation)
{
    const uint32_t total = pd.p * pd.d;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const uint32_t i = e / pd.d, g = e - i * pd.d;
        float x = values[e], s = steps[e];
        mutate_gene(x, s, pd.gid_base + i, g, generation, pd, mc);
        values[e] = x;
        steps[e] = s;
    }
}

__global__ __launch_bounds__(256) void k_recombine_mutate(const float *__restrict__ vin,
                                                          const float *__restrict__ sin,
                                                          float *__restrict__ vout,
                                                          float *__restrict__ sout, PopDims pd,
                                                          MutateConsts mc, uint32_t generation)
{
    const uint32_t total = pd.p * pd.d;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const uint32_t i = e / pd.d, g = e - i * pd.d;
        const uint32_t src = recombine_source(i, g, pd);
        float x = vin[src], s = sin[src];
        mutate_gene(x, s, pd.gid_base + i, g, generation, pd, mc);
        vout[e] = x;
        sout[e] = s;
    }
}

// Audio rows are read exactly once per generation: non-temporal loads keep them from
// displacing other data in L2 / Infinity Cache (P = 131072: 151 -> 132 us; neutral at 65536).
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4 *p)
{
    const v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
#define SOTS_ROW_LOAD(p) nt_load4(p)
// ------------------------------------------------------------------------------------
// synthesisePopulation{,DoubleSeries,TripleParallel}, ocl_program.cl:280-443 /
// Objective::synthesiseAudio*, Evolutionary_Strategy.hpp:368-495.
//
// The oscillator phases are fp32 running sums with a conditional wrap per sample, so a
// voice is serial in the sample index and cannot be split over lanes without changing the
// rounding (and with it table indices and spectra).  The 32768-entry wavetable (128 KiB)
// lives in LDS, one workgroup per CU, which leaves 32 KiB of LDS.  One kernel:
//   k_synth      every voice, one lane per individual.  A voice is J parallel chains of OPS
//                operators in series; operator s of block k-s runs in loop trip k (software
//                pipeline over blocks of 8 samples, two alternating register sets, no copies),
//                so no table read is waited for in the trip that issues it.  Finished samples
//                are parked in a swizzled LDS tile and leave as whole 128-byte lines.
//                For small populations a series chain is cut into two wavefronts (SPLIT, below).
// With one wavefront per SIMD every vector instruction costs four cycles, whatever it does, so
// the instruction count per sample is what the loop time follows (in-kernel s_memtime stamps,
// -DSOTS_STAMP): branch-free wraps (below), packed v_pk_* arithmetic for the per-sample
// mul/add/mul outside the recurrences.  Every per-sample operation is the reference's, in
// its order, unfused, so the audio is bit-identical to the serial loop.
// ------------------------------------------------------------------------------------
#ifndef SOTS_SYNTH_UNROLL
#define SOTS_SYNTH_UNROLL 8
#endif
constexpr int kSynthUnroll = SOTS_SYNTH_UNROLL; // samples per pipeline block
#ifndef SOTS_SYNTH_UNROLL_CUT
#define SOTS_SYNTH_UNROLL_CUT 16
#endif
// ... and of a chain cut over several wavefronts: a trip there ends in a workgroup barrier and starts with the read of
// the block handed over (together about 400 cycles), so longer blocks halve the number of trips
constexpr int kSynthUnrollCut = SOTS_SYNTH_UNROLL_CUT;
constexpr float kWf = (float)kWavetableSize;
typedef float v2f_t __attribute__((ext_vector_type(2)));

// table[clamp((int)pos, 0, W-1)]; CLAMP = false only where the phase is known to be in [0, W)
template <bool CLAMP = true>
__device__ __forceinline__ float tab_at(const float *tab, float pos)
{
    if constexpr (CLAMP) {
        // (round 4: v_cvt_u32_f32 + v_min_u32 - a saturating conversion and a two-operand minimum, as k_synth_ol does it - measures
        // SLOWER here: 52.2 against 51.3 us at P = 65 536, 99 against 95 at 131 072; the asm statement costs the scheduler more than
        // the cheaper minimum brings)
        int i = (int)pos; // == (unsigned)pos for every in-range phase
        i = min(max(i, 0), (int)kWavetableSize - 1);
        return tab[i];
    } else {
        return tab[(uint32_t)pos];
    }
}
// The reference's two conditional wraps, `if (p >= W) p -= W;` and `if (p < 0) p += W;`, as
// subtract/add and an UNSIGNED INTEGER minimum of the bit patterns.  Non-negative floats order
// like their bit patterns and every negative float is a larger unsigned number than every
// non-negative one.  The subtraction and addition are the reference's own fp32 operations, so
// the result carries its rounding.
//   wrap_hi (first wrap only, 2 instructions instead of 3): p >= W -> 0 <= p-W < p picks p-W;
//     0 <= p < W -> p-W < 0 picks p;  p < 0 -> p-W is further from zero than p, picks p.
//   wrap_both (both wraps, 3 instructions instead of 6, and the phase recurrence becomes
//     add -> {sub, add} -> v_min3_u32):  p >= W -> the reference takes p-W, which is >= 0 so its
//     second wrap does nothing, and p-W is the smallest pattern;  0 <= p < W -> it keeps p, p-W
//     is negative and p+W larger;  p < 0 -> it takes p+W, which is non-negative or closer to
//     zero than p and p-W.
// The one bit pattern that would differ is p == -0.0f in wrap_both (the reference keeps it, this
// gives W): a phase starts at +0.0f and x + y is -0.0f only when both are, so it never occurs.
__device__ __forceinline__ void wrap_hi(float &p)
{
    p = __uint_as_float(min(__float_as_uint(p - kWf), __float_as_uint(p)));
}
__device__ __forceinline__ void wrap_both(float &p)
{
    const v2f_t bc = v2f_t{p, p} + v2f_t{-kWf, kWf}; // one v_pk_add_f32
    p = __uint_as_float(min(min(__float_as_uint(bc.x), __float_as_uint(bc.y)), __float_as_uint(p)));
}

__device__ __forceinline__ void request_wavetable(float *__restrict__ tab, const float *__restrict__ wavetable)
{
    // global -> LDS directly (global_load_lds_dwordx4): one wavefront instruction lands 1 KiB at a
    // wavefront-uniform LDS base + lane * 16 B, no registers in between, all 128 in flight at once
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x & (kWave - 1);
    const uint32_t waves = blockDim.x / kWave;
    constexpr uint32_t kChunk = kWave * 4; // floats per instruction
    for (uint32_t ch = wave; ch < kWavetableSize / kChunk; ch += waves)
        __builtin_amdgcn_global_load_lds(wavetable + ch * kChunk + lane * 4u, (lds_ptr_t)(tab + ch * kChunk), 16, 0, 0);
}
// ... and the wait for it, after whatever else the kernel can start meanwhile
__device__ __forceinline__ void wavetable_ready()
{
    __builtin_amdgcn_s_waitcnt(0); // vmcnt(0): the copies have landed
    __syncthreads();
}

#if defined(SOTS_STAMP) || defined(SOTS_STAMP_ENDS)
// Diagnostic builds only (never the shipped library): per-wavefront shader cycles and 100 MHz
// ticks, read back with sots_debug_stamps().
__device__ unsigned long long g_stamps[2 * 16384];
#endif
#ifdef SOTS_STAMP
struct StampScope {
    unsigned long long t0, r0;
    uint32_t slot;
    __device__ StampScope(uint32_t s) : slot(s)
    {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    __device__ ~StampScope()
    {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if ((threadIdx.x & 63) == 0 && slot < 16384) {
            g_stamps[2 * slot] = t1 - t0;
            g_stamps[2 * slot + 1] = r1 - r0;
        }
    }
};
#define SOTS_STAMP_SCOPE(slot) StampScope stamp_scope_(slot)
// phase stamps of the selection kernels: shader cycles since the workgroup started, 16 phases per workgroup,
// kept behind the synthesis stamps (slots 8192..)
#define SOTS_PHASE_BEGIN() const unsigned long long phase_t0_ = __builtin_amdgcn_s_memtime()
#define SOTS_PHASE(n)                                                                                              \
    do {                                                                                                           \
        if (threadIdx.x == 0 && blockIdx.x < 512)                                                                  \
            g_stamps[2 * 8192 + blockIdx.x * 16 + (n)] = __builtin_amdgcn_s_memtime() - phase_t0_;                 \
    } while (0)
// absolute clock of any one lane (diagnostics of wavefront start skew inside a workgroup)
// the chip-wide 100 MHz clock of one lane (when workgroups start and end relative to each other)
#define SOTS_PHASE_REAL(n)                                                                                         \
    do {                                                                                                           \
        if (threadIdx.x == 0 && blockIdx.x < 512) g_stamps[2 * 8192 + blockIdx.x * 16 + (n)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define SOTS_PHASE_ABS(n, cond)                                                                                   \
    do {                                                                                                           \
        if ((cond) && blockIdx.x < 512) g_stamps[2 * 8192 + blockIdx.x * 16 + (n)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define SOTS_STAMP_SCOPE(slot)
#define SOTS_PHASE_BEGIN()
#define SOTS_PHASE(n)
#define SOTS_PHASE_REAL(n)
#define SOTS_PHASE_ABS(n, cond)
#endif

// ---- every voice, one lane per individual, whole-line stores through LDS -------------------
// Each wavefront owns an 8 KiB tile [64 rows][8 chunks of 16 B], XOR-swizzled by row so that
// both the row-wise writes (lane = row) and the transposed reads (lane = 8 rows x 8 chunks) are
// bank-conflict free.  Every 32 samples the tile is read back transposed and written with
// stores in which 8 neighbouring lanes cover one whole 128-byte line of one row.  (The
// row-per-lane 16-byte store it replaces touches 64 lines per instruction and 8 L2 requests
// per line: 91 us instead of 57 us for the 2-operator voice at P = 65536.)
constexpr int kSynthWaves = 4;  // at most: one per SIMD, 4 x 8 KiB of tiles beside the table
constexpr int kStageChunks = 8; // 16-byte chunks per row per flush = 32 samples

template <int KIND> struct VoiceShape;
template <> struct VoiceShape<SOTS_SYNTH_2OP> { static constexpr int J = 1, OPS = 2, D = 4; };
template <> struct VoiceShape<SOTS_SYNTH_3OP_SERIES> { static constexpr int J = 1, OPS = 3, D = 6; };
template <> struct VoiceShape<SOTS_SYNTH_4OP_SERIES> { static constexpr int J = 1, OPS = 4, D = 8; };
template <> struct VoiceShape<SOTS_SYNTH_TRIPLE_PAR> { static constexpr int J = 3, OPS = 2, D = 12; };

template <int V> using ic = std::integral_constant<int, V>;

// SPLIT > 0 (series voices, at most two wavefronts' worth of individuals per CU): the chain is cut in
// front of operator SPLIT and runs in TWO wavefronts per 64 individuals, so that all four SIMDs of
// a CU work when the population is small.  The FRONT wavefront runs operators 0..SPLIT-1 and hands
// c * (t * mul + off) - operator SPLIT's phase increments, the same unfused arithmetic - over
// through LDS, one 8-sample block per trip; the BACK wavefront runs operators SPLIT..OPS-1 and
// the tile.  Both execute the same sequence of trips with one workgroup barrier per trip: the
// block handed over in trip k is consumed in trip k+1, exactly when the one-wavefront pipeline
// would read it from registers, and the two hand-over buffers alternate with the block parity.
// HELP (fused generation loop, 4-gene voice, one or two tiles per workgroup): the workgroup starts with four times its
// wavefronts (sixteen for a full tile); each thread makes one gene of the tile's individuals (recombination source, 13 Philox draws, exp, pow), so
// the variation runs once across 1024 lanes at four wavefronts per SIMD instead of four times in a row in each of
// 256 lanes at one; the twelve extra wavefronts then leave and the usual four synthesise.
// SPLIT2 > SPLIT: a second cut in front of operator SPLIT2 - three wavefronts per 64 individuals (stages of operators
// [0, SPLIT), [SPLIT, SPLIT2), [SPLIT2, OPS)), two hand-over links; for the 4-operator voice at <= 128 individuals per CU
// where a series chain of OPS operators is cut, as compile-time facts
// SPLIT3 > SPLIT2: a third cut - with SPLIT = 1, 2, 3 every operator of the 4-operator voice has a wavefront of its own,
// four wavefronts per 64 individuals and three links: eight wavefronts for the 128 individuals per CU of BASELINE configs[3]'s
// shard, two per SIMD (one wavefront alone on a SIMD issues an instruction every ~4.5 cycles, two sharing it every ~2.3).
// The 32 KiB beside the table then hold 2 tiles of 4 KiB (16 samples, leaving as half lines) and 6 hand-over buffers of
// 4 KiB (two 8-sample blocks each).
template <int SPLIT, int SPLIT2, int SPLIT3, int OPS> struct CutPlan {
    static constexpr int STAGES = 1 + (SPLIT > 0 ? 1 : 0) + (SPLIT2 > 0 ? 1 : 0) + (SPLIT3 > 0 ? 1 : 0);
    static constexpr int cuts_before(int S) { return (SPLIT > 0 && S >= SPLIT ? 1 : 0) + (SPLIT2 > 0 && S >= SPLIT2 ? 1 : 0) + (SPLIT3 > 0 && S >= SPLIT3 ? 1 : 0); }
    // stage of operator S: 0 = the chain's head ... STAGES-1 = its tail (which also owns the tile)
    static constexpr int stage_of(int S) { return cuts_before(S); }
    // In trip k operator S works on block k - slot(S): one trip behind the operator in front of it, TWO behind it
    // across a cut (the stage in front sends a block in the trip AFTER it read the table for it, when the values
    // have landed, and the block is read in the trip after that); block k - slot(OPS) leaves.
    static constexpr int slot(int S) { return S + cuts_before(S); }
    static constexpr bool is_cut(int S) { return S > 0 && (S == SPLIT || S == SPLIT2 || S == SPLIT3); }
    static constexpr bool consumes(int S) { return SPLIT > 0 && is_cut(S); }
    static constexpr bool produces(int S) { return SPLIT > 0 && S + 1 < OPS && is_cut(S + 1); }
    // samples per pipeline block and 16-byte chunks per tile row: the deepest cut has the least LDS per buffer
    static constexpr int U = SPLIT == 0 ? kSynthUnroll : STAGES >= 4 ? 8 : kSynthUnrollCut;
    static constexpr int CH = STAGES >= 4 ? 4 : kStageChunks;
};

template <int KIND, int SPLIT, bool HELP = false, int SPLIT2 = 0, int SPLIT3 = 0>
__global__ __launch_bounds__((HELP ? (SPLIT ? 8 : 16) : SPLIT3 ? 8 : SPLIT2 ? 6 : 4) * kWave) void k_synth(const float *__restrict__ values,
                                                               const float *__restrict__ wavetable,
                                                               float *__restrict__ audio, SynthParams sp,
                                                               uint32_t p_len, uint32_t n, uint32_t pitch, Variation var)
{
    constexpr int J = VoiceShape<KIND>::J, OPS = VoiceShape<KIND>::OPS, D = VoiceShape<KIND>::D;
    // SPLIT < 0 (the voice of J parallel chains, up to 64 individuals per CU): a wavefront per CHAIN.  Every wavefront runs
    // its chain's operators as the uncut pipeline does; the chains of wavefronts 1 ... J-1 hand gain * (table value) - the
    // reference's product - to wavefront 0 through LDS, one block per trip and one workgroup barrier per trip as in the
    // series cuts, and wavefront 0 adds them to its own in the reference's order and owns the tile.
    constexpr bool PSPLIT = SPLIT < 0;
    using Plan = CutPlan<(PSPLIT ? 0 : SPLIT), SPLIT2, SPLIT3, OPS>;
    constexpr int U = PSPLIT ? kSynthUnrollCut : Plan::U;
    constexpr int CH = PSPLIT ? kStageChunks : Plan::CH; // 16-byte chunks per tile row = 4 CH samples per flush
    constexpr int RPI = kWave / CH;     // rows per transposed read / store instruction (CH of them per flush)
    static_assert(U % 4 == 0 && 4 * CH % U == 0, "whole 16-byte chunks, whole blocks per flush");
    static_assert(SPLIT <= 0 || (J == 1 && SPLIT < OPS), "only a series chain can be cut");
    static_assert(!PSPLIT || (J == 3 && OPS == 2 && !HELP && SPLIT2 == 0), "the parallel split serves the three 2-operator chains");
    static_assert(SPLIT2 == 0 || (SPLIT > 0 && SPLIT2 > SPLIT && SPLIT2 < OPS), "the second cut lies behind the first");
    static_assert(SPLIT3 == 0 || (SPLIT2 > 0 && SPLIT3 > SPLIT2 && SPLIT3 < OPS), "the third cut lies behind the second");
    static_assert(!HELP || SPLIT > 0 || D == 4, "uncut, the helper wavefronts serve the 4-gene voice (128 registers with 16 wavefronts)");
    constexpr uint32_t HT = 4; // HELP: threads per individual while the genes are made (each takes genes t % 4, t % 4 + 4, ...)
    constexpr int STAGES = PSPLIT ? J : Plan::STAGES; // wavefronts per 64 individuals
    __shared__ float tab[kWavetableSize];
    __shared__ float4 stage_all[kSynthWaves * kWave * kStageChunks];
    SOTS_PHASE_BEGIN();
    SOTS_PHASE_REAL(10);
    request_wavetable(tab, wavetable);
    bool table_pending = true; // the first tile's parameters are fetched while the table is on its way
    const float c = (float)kWavetableSize / (float)SOTS_SAMPLE_RATE; // w2srRatio, Evolutionary_Strategy.hpp:203
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    // cut kernels: wavefronts [0, pairs) are the TAIL stage, [pairs, 2 pairs) the stage before it, ... of the same 64 individuals
    const uint32_t pairs = HELP ? blockDim.x / (HT * kWave) : blockDim.x / (STAGES * kWave);
    // HELP: the tiles this workgroup will synthesise (1 or 2: blockIdx, blockIdx + gridDim)
    const uint32_t help_tiles = HELP ? (blockIdx.x * (pairs * kWave) + gridDim.x * (pairs * kWave) < p_len ? 2u : 1u) : 0u;
    if constexpr (HELP) {
        // one gene per thread and tile; values and steps go to the other half, the values also to LDS for the
        // wavefronts that stay (the tile area is free until the first samples are parked; they take the second tile's
        // values into registers before they start on the first)
        float *__restrict__ made = reinterpret_cast<float *>(stage_all);
        const uint32_t t = threadIdx.x, li = t / HT; // the individual inside the tile
        for (uint32_t kt = 0; kt < help_tiles; ++kt) {
            const uint32_t i1 = (blockIdx.x + kt * gridDim.x) * (pairs * kWave) + li;
            if (i1 < p_len) {
                for (uint32_t g1 = t % HT; g1 < (uint32_t)D; g1 += HT) {
                    const uint32_t src = recombine_source(i1, g1, var.pd);
                    float x = var.vin[src], st = var.sin[src];
                    mutate_gene(x, st, var.pd.gid_base + i1, g1, var.generation, var.pd, var.mc);
                    var.vout[(size_t)i1 * D + g1] = x;
                    var.sout[(size_t)i1 * D + g1] = st;
                    made[kt * (pairs * kWave * D) + li * D + g1] = x;
                }
            }
        }
        __syncthreads();
        SOTS_PHASE(12); // the individuals are made
        if (wave_id >= STAGES * pairs) { // (a cut kernel keeps a wavefront per stage)
            __builtin_amdgcn_s_waitcnt(0); // this wavefront's pieces of the table have landed before it leaves
            return;
        }
    }
    const uint32_t rho = SPLIT ? wave_id / pairs : 0u;          // 0: tail stage
    const int my_stage = STAGES - 1 - (int)rho; // (pairing the tail with the head stage on a SIMD instead: no difference, profiles/r03_experiments.md)
    const bool front = my_stage != STAGES - 1;                  // not the tail: no tile, no stores
    const uint32_t wave = wave_id - rho * pairs;                // which 64 individuals of the workgroup's tile
    float4 *__restrict__ stage = stage_all + wave * kWave * CH;
    // hand-over buffers [link][pair][parity][U/4][lane] of 16 bytes behind the tiles of a cut kernel (launch_synth keeps
    // pairs * (tile + links * hand-over buffer) inside stage_all: two pairs with one link, one pair with two, two pairs
    // with three links of half-length blocks and half-length tiles)
    float4 *__restrict__ xbuf0 = stage_all + pairs * kWave * CH + wave * (2 * (U / 4) * kWave);
    float4 *__restrict__ xbuf1 = xbuf0 + pairs * (2 * (U / 4) * kWave);
    float4 *__restrict__ xbuf2 = xbuf1 + pairs * (2 * (U / 4) * kWave);
    // write side: lane = row; chunk q of the row lives in slot q ^ swz(row): row & 7 with eight chunks, (row >> 1) & 3 with four
    auto swz = [](uint32_t row) { return CH == 8 ? (row & 7u) : ((row >> 1) & 3u); };
    float4 *__restrict__ wr = stage + lane * CH;
    const uint32_t l7 = swz(lane);
    // read side: lane = (row within a group of RPI rows, chunk): group `it` adds 64 slots
    const uint32_t r8 = lane / CH, rch = lane % CH;
    const float4 *__restrict__ rd = stage + r8 * CH + (rch ^ swz(r8)); // (RPI is a multiple of what swz looks at: the group index drops out)
    const uint32_t lane_off = r8 * pitch + 4u * rch; // floats, relative to the group's first row
    uint32_t row_off[CH];                           // ... and, in BYTES, of this lane's piece of a line in each of the CH groups of RPI rows
#pragma unroll
    for (uint32_t g = 0; g < (uint32_t)CH; ++g) row_off[g] = (lane_off + g * RPI * pitch) * 4u; // < 2^32: 64 rows of at most 8224 floats

    const uint32_t rows_per_block = pairs * kWave;
    float help_next[HELP ? D : 1]; // HELP: the second tile's values
    for (uint32_t base = blockIdx.x * rows_per_block; base < p_len; base += gridDim.x * rows_per_block) {
        const uint32_t row0 = base + wave * kWave; // first row of this wavefront
        const uint32_t ind = row0 + lane < p_len ? row0 + lane : p_len - 1u;
        const bool full = row0 + kWave <= p_len; // every row of the tile exists
        float p[D];
        if constexpr (HELP) {
            if (base == blockIdx.x * rows_per_block) { // first tile: both tiles' values leave LDS now
                const float *__restrict__ made = reinterpret_cast<const float *>(stage_all);
#pragma unroll
                for (int g = 0; g < D; ++g) {
                    p[g] = made[(wave * kWave + lane) * D + g];
                    help_next[g] = help_tiles > 1 ? made[rows_per_block * D + (wave * kWave + lane) * D + g] : 0.0f;
                }
            } else {
#pragma unroll
                for (int g = 0; g < D; ++g) p[g] = help_next[g];
            }
        } else if (var.vin) {
            // fused generation loop: this lane's individual is made here (k_recombine_mutate's
            // arithmetic, gene by gene) and written to the other half by the wavefront that owns the row
            const bool owner = !front && row0 + lane < p_len;
#pragma unroll 1
            for (int g = 0; g < D; ++g) {
                const uint32_t src = recombine_source(ind, (uint32_t)g, var.pd);
                float x = var.vin[src], st = var.sin[src];
                mutate_gene(x, st, var.pd.gid_base + ind, (uint32_t)g, var.generation, var.pd, var.mc);
                if (owner) {
                    var.vout[(size_t)ind * D + g] = x;
                    var.sout[(size_t)ind * D + g] = st;
                }
                p[g] = x;
            }
        } else {
#pragma unroll
            for (int g = 0; g < D; ++g) p[g] = values[(size_t)ind * D + g];
        }
#pragma unroll
        for (int g = 0; g < D; ++g) {
            // scaleParams: min + v*(max-min), ocl_program.cl:297; the triple voice scales all
            // three 2-op voices by entries 0..3, Evolutionary_Strategy.hpp:453-455
            const int sc = KIND == SOTS_SYNTH_TRIPLE_PAR ? (g & 3) : g;
            p[g] = sp.pmin[sc] + p[g] * (sp.pmax[sc] - sp.pmin[sc]);
        }
        // Operator 0 of chain j runs free at inc0[j]; operator s >= 1 advances by
        // c * (t_prev * mul[s][j] + off[s][j]); the output is gain[j] * (last operator's table value).
        float inc0[J], mul[OPS][J], off[OPS][J], gain[J];
        if constexpr (KIND == SOTS_SYNTH_2OP) { // Evolutionary_Strategy.hpp:372-401
            inc0[0] = c * p[0], mul[1][0] = p[0] * p[1], off[1][0] = p[2], gain[0] = p[3];
        } else if constexpr (KIND == SOTS_SYNTH_TRIPLE_PAR) { // :457-494
#pragma unroll
            for (int j = 0; j < J; ++j)
                inc0[j] = c * p[4 * j], mul[1][j] = p[4 * j] * p[4 * j + 1], off[1][j] = p[4 * j + 2], gain[j] = p[4 * j + 3];
        } else { // series, :407-445 (the 4-operator voice adds one more modulator stage)
            inc0[0] = c * p[1];
#pragma unroll
            for (int o = 1; o < OPS; ++o) mul[o][0] = p[2 * (o - 1)] * p[2 * (o - 1) + 1], off[o][0] = p[2 * (o - 1) + 3];
            gain[0] = p[2 * (OPS - 1)] * p[2 * (OPS - 1) + 1];
        }
        // a free-running phase stays inside [0, W) when 0 <= inc0 < W: its index needs no clamp
        // (round 4: the modulated phases likewise, where every increment they can receive is bounded by c (|mul| + |off|) < W - the
        // unsigned conversion and one two-operand minimum instead of the signed one and v_med3_i32: no difference here, 51.0-51.3
        // against 51.3-51.4 us; k_synth_ol, where the index goes clamp-FREE on that condition, gains 8 %)
        bool in_range = true;
#pragma unroll
        for (int j = 0; j < J; ++j) in_range = in_range && inc0[j] >= 0.0f && inc0[j] < kWf;
        const bool free_unclamped = __all(in_range);
        if (table_pending) {
            wavetable_ready();
            table_pending = false;
            SOTS_PHASE(13); // the table is in LDS
        }
        SOTS_STAMP_SCOPE(blockIdx.x * (blockDim.x / kWave) + wave_id);

        auto run = [&](auto unclamped_tag, auto stage_tag) {
            constexpr bool UNCLAMPED = decltype(unclamped_tag)::value;
            constexpr int MY = decltype(stage_tag)::value; // this wavefront's stage, a compile-time fact in here: its loop holds
                                                          // (and keeps registers for) its own operators only
            constexpr int CHAIN = PSPLIT ? STAGES - 1 - MY : 0; // parallel split: this wavefront's chain (the tail runs chain 0)
            float pos[OPS][J];
#pragma unroll
            for (int o = 0; o < OPS; ++o)
#pragma unroll
                for (int j = 0; j < J; ++j) pos[o][j] = 0.0f;
            float T[OPS][2][J][U]; // table values of operator s, block parity, chain, sample
            v2f_t handed[U / 2];   // cut kernels: the increments fetched for this wavefront's first operator
            v2f_t to_hand[U / 2];  // ... and the ones made for the next stage's
            v2f_t handed2[PSPLIT ? U / 2 : 1]; // parallel split: handed = chain 1's products, handed2 = chain 2's
            // parallel split: this chain's products of its block of parity B (table values read one trip ago) ...
            auto par_make = [&](auto b_tag) {
                constexpr int B = decltype(b_tag)::value;
                if constexpr (PSPLIT && CHAIN != 0) {
#pragma unroll
                    for (int u = 0; u < U; u += 2)
                        to_hand[u / 2] = v2f_t{T[OPS - 1][B][CHAIN][u], T[OPS - 1][B][CHAIN][u + 1]} * gain[CHAIN];
                }
            };
            // ... go to the tail through this chain's link (chain 1: xbuf0, chain 2: xbuf1) ...
            auto par_send = [&](auto b_tag) {
                constexpr int B = decltype(b_tag)::value;
                if constexpr (PSPLIT && CHAIN != 0) {
                    float4 *__restrict__ xout = CHAIN == 1 ? xbuf0 : xbuf1;
#pragma unroll
                    for (int u = 0; u < U; u += 4)
                        xout[(B * (U / 4) + u / 4) * kWave + lane] =
                            make_float4(to_hand[u / 2].x, to_hand[u / 2].y, to_hand[u / 2 + 1].x, to_hand[u / 2 + 1].y);
                }
            };
            // ... where the tail picks both up one trip later
            auto par_fetch = [&](auto b_tag) {
                constexpr int B = decltype(b_tag)::value;
                if constexpr (PSPLIT && CHAIN == 0) {
#pragma unroll
                    for (int u = 0; u < U; u += 4) {
                        const float4 q1 = xbuf0[(B * (U / 4) + u / 4) * kWave + lane], q2 = xbuf1[(B * (U / 4) + u / 4) * kWave + lane];
                        handed[u / 2] = v2f_t{q1.x, q1.y}, handed[u / 2 + 1] = v2f_t{q1.z, q1.w};
                        handed2[u / 2] = v2f_t{q2.x, q2.y}, handed2[u / 2 + 1] = v2f_t{q2.z, q2.w};
                    }
                }
            };

            // Schedule (CutPlan): operator S works on block k - slot(S) in trip k.  A block's parity (which half of T,
            // which hand-over buffer) is its index & 1.
            // link = the producer's stage: stage 0 sends through xbuf0, stage 1 through xbuf1, stage 2 through xbuf2
            auto link_of = [&](int producer) -> float4 * { return Plan::stage_of(producer) == 0 ? xbuf0 : Plan::stage_of(producer) == 1 ? xbuf1 : xbuf2; };

            // the increments of operator S's block (parity B), asked for at the start of the trip
            auto fetch = [&](auto s_tag, auto b_tag) {
                constexpr int S = decltype(s_tag)::value, B = decltype(b_tag)::value;
                if constexpr (Plan::consumes(S) && Plan::stage_of(S) == MY) {
                    const float4 *__restrict__ xin = link_of(S - 1);
#pragma unroll
                    for (int u = 0; u < U; u += 4) {
                        const float4 q = xin[(B * (U / 4) + u / 4) * kWave + lane];
                        handed[u / 2] = v2f_t{q.x, q.y}, handed[u / 2 + 1] = v2f_t{q.z, q.w};
                    }
                }
            };
            // operator S + 1's increments from operator S's block of parity B (read from the table one trip ago):
            // c * (t * mul + off), the reference's mul, add, mul, unfused, two samples per packed instruction
            auto make_handover = [&](auto s_tag, auto b_tag) {
                constexpr int S = decltype(s_tag)::value, B = decltype(b_tag)::value;
                if constexpr (Plan::produces(S) && Plan::stage_of(S) == MY) {
                    constexpr int NX = S + 1 < OPS ? S + 1 : S;
#pragma unroll
                    for (int u = 0; u < U; u += 2)
                        to_hand[u / 2] = (v2f_t{T[S][B][0][u], T[S][B][0][u + 1]} * mul[NX][0] + off[NX][0]) * c;
                }
            };
            auto send_handover = [&](auto s_tag, auto b_tag) {
                constexpr int S = decltype(s_tag)::value, B = decltype(b_tag)::value;
                if constexpr (Plan::produces(S) && Plan::stage_of(S) == MY) {
                    float4 *__restrict__ xout = link_of(S);
#pragma unroll
                    for (int u = 0; u < U; u += 4)
                        xout[(B * (U / 4) + u / 4) * kWave + lane] =
                            make_float4(to_hand[u / 2].x, to_hand[u / 2].y, to_hand[u / 2 + 1].x, to_hand[u / 2 + 1].y);
                }
            };
            // operator S on its block of parity B
            auto op = [&](auto s_tag, auto b_tag) {
                constexpr int S = decltype(s_tag)::value, B = decltype(b_tag)::value;
                if constexpr (PSPLIT || Plan::stage_of(S) == MY) // else: another wavefront's operator
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    if (PSPLIT && j != CHAIN) continue; // (compile-time after unrolling: another wavefront's chain)
                    if constexpr (S == 0) {
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            T[0][B][j][u] = tab_at<!UNCLAMPED>(tab, pos[0][j]);
                            pos[0][j] += inc0[j];
                            wrap_hi(pos[0][j]);
                        }
                    } else {
                        // phase increments, two samples per packed instruction (v_pk_mul_f32,
                        // v_pk_add_f32, v_pk_mul_f32: the reference's mul, add, mul, unfused)
                        v2f_t inc[U / 2];
                        if constexpr (Plan::consumes(S)) { // handed over by the wavefront of the stage in front
#pragma unroll
                            for (int u = 0; u < U; u += 2) inc[u / 2] = handed[u / 2];
                        } else {
#pragma unroll
                            for (int u = 0; u < U; u += 2)
                                inc[u / 2] = (v2f_t{T[S - 1][B][j][u], T[S - 1][B][j][u + 1]} * mul[S][j] + off[S][j]) * c;
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            T[S][B][j][u] = tab_at(tab, pos[S][j]);
                            pos[S][j] += (u & 1) ? inc[u / 2].y : inc[u / 2].x;
                            wrap_both(pos[S][j]);
                        }
                    }
                }
            };
            // a tile on its way out: eight rows x 128-byte lines per store instruction, the address a wavefront-uniform
            // line start plus a per-lane offset that never changes (no vector address arithmetic per flush)
            float pend[CH][4]; // (plain floats: an array of float4 does not leave memory for registers)
            float *__restrict__ pend_line = audio;
            bool pending = false;
            auto store_pending = [&]() {
                if constexpr (MY != STAGES - 1) return;
                if (!pending) return;
                pending = false;
#pragma unroll
                for (int g = 0; g < CH; ++g) {
                    asm volatile("" : "+v"(row_off[g])); // the zero-extension stays next to the store: scalar base + 32-bit lane offset
                    *reinterpret_cast<float4 *>(reinterpret_cast<char *>(pend_line) + row_off[g]) = make_float4(pend[g][0], pend[g][1], pend[g][2], pend[g][3]);
                }
            };
            // samples ip..ip+U-1 (the block of parity B) leave
            auto emit = [&](auto b_tag, uint32_t ip) {
                constexpr int B = decltype(b_tag)::value;
                if constexpr (MY != STAGES - 1) return; // only the tail stage has samples
                float y[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if constexpr (PSPLIT) // chains 1 and 2 arrive as products from their wavefronts: the same sum in the same order
                        y[u] = (T[OPS - 1][B][0][u] * gain[0] + ((u & 1) ? handed[u / 2].y : handed[u / 2].x) +
                                ((u & 1) ? handed2[u / 2].y : handed2[u / 2].x)) / 3.0f;
                    else if constexpr (KIND == SOTS_SYNTH_TRIPLE_PAR)
                        y[u] = (T[OPS - 1][B][0][u] * gain[0] + T[OPS - 1][B][1][u] * gain[1] + T[OPS - 1][B][2][u] * gain[2]) /
                               3.0f; // == (float)(double(sum)/3.0), :493
                    else
                        y[u] = T[OPS - 1][B][0][u] * gain[0];
                }
                store_pending(); // the tile read back one trip ago leaves now: its LDS reads have long landed
                const uint32_t c0 = (ip >> 2) & (CH - 1);
#pragma unroll
                for (int q = 0; q < U / 4; ++q) wr[(c0 + q) ^ l7] = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
#ifdef SOTS_ABL_NOFLUSH
                if (false) {
#else
                if (c0 == CH - U / 4) { // 4 CH samples parked: flush the tile
#endif
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("" ::: "memory");
                    const uint32_t i0 = ip + U - 4 * CH;
                    float *__restrict__ line0 = audio + (size_t)row0 * pitch + i0; // wavefront-uniform
                    if (full) {
                        // read back transposed now (LDS works in order: the next samples are parked behind these
                        // reads), stored in the next trip
                        constexpr int G = RPI * CH; // slots per group of RPI rows (= 64)
#pragma unroll
                        for (int g = 0; g < CH; ++g) {
                            const float4 q = rd[g * G];
                            pend[g][0] = q.x, pend[g][1] = q.y, pend[g][2] = q.z, pend[g][3] = q.w;
                        }
                        pend_line = line0;
                        pending = true;
                    } else { // last, partly filled tile of the population
#pragma unroll 1
                        for (uint32_t it = 0; it < (uint32_t)CH; ++it)
                            if (row0 + RPI * it + r8 < p_len)
                                *reinterpret_cast<float4 *>(line0 + lane_off + (size_t)(RPI * it) * pitch) = rd[it * RPI * CH];
                    }
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("" ::: "memory");
                }
            };
            const uint32_t nb = n / U; // even and >= 32
            // the slot in which a block leaves; parallel split: one trip later than the uncut pipeline (the other chains' products
            // are made in slot OPS and read in slot OPS + 1; the tail's own table values of that block live until the operator
            // of the same parity overwrites them later in the same trip: the samples leave first)
            constexpr int LAST = PSPLIT ? OPS + 1 : Plan::slot(OPS);
            // Trip k of parity Q.  EDGE (the first and last trips): a slot only works while its block index lies in
            // [0, nb).  Cut kernels: what a wavefront waits for from another stage is asked for first, work that does
            // not need it comes next (the hand-over arithmetic of the block read one trip ago, the samples that leave),
            // then the operators; the hand-over is written last, and only a wavefront that wrote one waits for its LDS
            // operations before the barrier (the tail stage keeps its table reads in flight across it).
            auto trip = [&](auto q_tag, auto edge_tag, uint32_t k) {
                constexpr int Q = decltype(q_tag)::value;
                constexpr bool EDGE = decltype(edge_tag)::value;
                auto on = [&](int sl) { return !EDGE || (k >= (uint32_t)sl && k - (uint32_t)sl < nb); };
                auto each_op = [&](auto &&f, auto shift_tag) { // f(S, parity of the block of slot(S) + shift) where that block exists
                    constexpr int SH = decltype(shift_tag)::value;
                    if (on(Plan::slot(0) + SH)) f(ic<0>{}, ic<(Q ^ ((Plan::slot(0) + SH) & 1))>{});
                    if constexpr (OPS > 1) if (on(Plan::slot(1) + SH)) f(ic<1>{}, ic<(Q ^ ((Plan::slot(1) + SH) & 1))>{});
                    if constexpr (OPS > 2) if (on(Plan::slot(2) + SH)) f(ic<2>{}, ic<(Q ^ ((Plan::slot(2) + SH) & 1))>{});
                    if constexpr (OPS > 3) if (on(Plan::slot(3) + SH)) f(ic<3>{}, ic<(Q ^ ((Plan::slot(3) + SH) & 1))>{});
                };
                if constexpr (PSPLIT) {
                    if (on(LAST)) par_fetch(ic<(Q ^ (LAST & 1))>{});
                    if (on(OPS)) par_make(ic<(Q ^ (OPS & 1))>{});
                    if (on(LAST)) emit(ic<(Q ^ (LAST & 1))>{}, (k - LAST) * U);
                    each_op(op, ic<0>{});
                    if (on(OPS)) par_send(ic<(Q ^ (OPS & 1))>{});
                    asm volatile("" ::: "memory");
                    if constexpr (CHAIN != 0) __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the products are in LDS
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                } else if constexpr (SPLIT > 0) {
                    each_op(fetch, ic<0>{});
                    each_op(make_handover, ic<1>{}); // the block this operator read the table for one trip ago
                    if (on(LAST)) emit(ic<(Q ^ (LAST & 1))>{}, (k - LAST) * U);
                    each_op(op, ic<0>{});
                    each_op(send_handover, ic<1>{});
                    asm volatile("" ::: "memory");
#ifndef SOTS_ABL_NOBARRIER
                    if constexpr (MY != STAGES - 1) __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the hand-over is in LDS
                    __builtin_amdgcn_s_barrier();
#endif
                    asm volatile("" ::: "memory");
                } else {
                    each_op(op, ic<0>{});
                    if (on(LAST)) emit(ic<(Q ^ (LAST & 1))>{}, (k - LAST) * U);
                }
            };
            constexpr uint32_t K0 = (LAST + 1) & ~1; // first trip with every slot busy, rounded to even
            for (uint32_t k = 0; k < K0; k += 2) {
                trip(ic<0>{}, std::true_type{}, k);
                trip(ic<1>{}, std::true_type{}, k + 1);
            }
            for (uint32_t k = K0; k < nb; k += 2) {
                trip(ic<0>{}, std::false_type{}, k);
                trip(ic<1>{}, std::false_type{}, k + 1);
            }
            for (uint32_t k = nb; k < nb + K0; k += 2) { // the last block leaves in trip nb - 1 + LAST
                trip(ic<0>{}, std::true_type{}, k);
                trip(ic<1>{}, std::true_type{}, k + 1);
            }
            store_pending();
        };
        auto run_my_stage = [&](auto unclamped_tag) {
            if constexpr (STAGES == 1) run(unclamped_tag, ic<0>{});
            else if (my_stage == 0) run(unclamped_tag, ic<0>{});
            else if (STAGES == 2 || my_stage == 1) run(unclamped_tag, ic<1>{});
            else if (STAGES == 3 || my_stage == 2) run(unclamped_tag, ic<(STAGES > 2 ? 2 : 0)>{});
            else run(unclamped_tag, ic<(STAGES > 3 ? 3 : 0)>{});
        };
        if (free_unclamped) run_my_stage(std::true_type{});
        else run_my_stage(std::false_type{});
        SOTS_PHASE(14); // the tile's samples are stored (issued)
    }
    if (table_pending) wavetable_ready(); // a workgroup without a tile must not end with copies in flight
    SOTS_PHASE_REAL(11);
}

// ------------------------------------------------------------------------------------
// Series voices: the OPERATORS in the lanes (k_synth_ol, round 4).
//
// k_synth gives an individual a lane, so a CU's share of 128 ... 256 individuals is two to four wavefronts - one per SIMD
// at best, and a wavefront alone on its SIMD issues an instruction every ~4.5 cycles where two or more sharing it issue
// one every ~2.3.  The cut kernels buy the second wavefront per SIMD with a wavefront per operator, LDS hand-overs and a
// workgroup barrier per 8 samples (4-op, 128 per CU: 24 of 89 busy LDS cycles per sample and 28 us of barrier).
// Here a LANE is one OPERATOR of one individual: a row of 16 lanes holds R = 8 / 4 / 4 individuals of the 2- / 3- / 4-operator
// voice, operator s of individual i in lane R s + i (3 operators: R (s + 1) + i, the lanes in front make operator 0's constant
// increment), so a wavefront carries 32 / 16 / 16 individuals and the CU's share is twice / four times as many wavefronts,
// every one of them running ONE instruction stream in which all 64 lanes work:
//   C  the lane's operator on its block of 8 samples: table read at the phase, phase += increment, wrap - the reference's
//      operations (operator 0's second wrap adds 0 instead of W: it has only the first, Evolutionary_Strategy.hpp:383,418);
//   A  the block read ONE TRIP AGO (its table values have landed): t * pm (for the last operator: the sample), + po, * c -
//      the next operator's increments c * (t * mul + off), unfused, two samples per packed instruction;
//   B  hand-over to the operator behind: ONE v_mov_b32_dpp row_shr:R per sample, no LDS, no barrier.  The bottom R lanes
//      of a row have no source lane; DPP leaves such lanes alone, so operator 0's registers keep c * p1, the free-running
//      operator's constant increment.
// Operator s works on block k - 2 s in trip k (its increments were handed over in trip k - 1 from table values read in
// trip k - 2).  The last operator's samples are spread over the G lanes of their individual (row_shl) so that every lane parks
// 32 / G bytes per trip in the wavefront's tile with ONE LDS instruction (a write instruction costs the LDS the same 6-13 cycles
// whether 16 lanes carry data or 64); rows = individuals, whole 128-byte lines leave every 32 samples as in k_synth.
// Per sample and lane ~9 vector instructions + 1 LDS gather.  Every sample sees the reference's operations in its order:
// bit-identical to k_synth and the oracle.
// What it reaches (profiles/r04_experiments.md; tools/ubench/ol_loop.hip, lds_gather.hip): 87 cycles per sample for the
// median wavefront and 97 for the slowest at configs[3]'s shard (k_synth<4OP, 1, ., 2, 3>: 112) - not the ~55 the LDS would
// allow (a random 64-lane gather costs it 4.8-5.9 cycles, eight per sample): a wavefront of this instruction mix (v_cvt, v_pk_*,
// three-operand integer minima, DPP moves at ~8 cycles each) needs 47-53 cycles per sample ALONE on its SIMD, a second one
// adds only ~30 % (the SIMD issues oldest-first: 54 / 78 in isolation, the kernel ends with the younger), so only the
// 4-operator voice at 65 ... 128 individuals per CU runs here (launch_synth).
// ------------------------------------------------------------------------------------
#ifndef SOTS_OL_ABL
#define SOTS_OL_ABL 0
#endif
template <int OPS> struct OlShape {
    static constexpr int G = OPS == 2 ? 2 : 4; // lanes per individual (3 operators: a fourth lane in front of operator 0 supplies its constant increment)
    static constexpr int R = 16 / G;         // individuals per row of 16 lanes: 8 or 4
    static constexpr int IPW = 4 * R;        // individuals per wavefront: 32 or 16
    static constexpr int NI = IPW / 8;       // store instructions per flush (8 rows x 128 bytes each)
};
constexpr int kOlTileRows = 240; // individuals per workgroup at most: 240 x 128 bytes of tiles + the table's spare entry in the 32 KiB beside the table
template <int OPS> constexpr int ol_max_waves() { return kOlTileRows / OlShape<OPS>::IPW; } // 7 or 15

template <int KIND>
__global__ __launch_bounds__(ol_max_waves<VoiceShape<KIND>::OPS>() * kWave) void k_synth_ol(const float *__restrict__ values,
                                                               const float *__restrict__ wavetable,
                                                               float *__restrict__ audio, SynthParams sp,
                                                               uint32_t p_len, uint32_t n, uint32_t pitch, Variation var)
{
    constexpr int OPS = VoiceShape<KIND>::OPS, D = VoiceShape<KIND>::D;
    static_assert(VoiceShape<KIND>::J == 1, "series voices");
    constexpr int G = OlShape<OPS>::G, R = OlShape<OPS>::R, IPW = OlShape<OPS>::IPW, NI = OlShape<OPS>::NI;
#ifndef SOTS_OL_U
#define SOTS_OL_U 16
#endif
    // samples per trip: 16 where a lane is one of four per individual (the per-trip work - tile addresses, the flush test, loop and
    // priority bookkeeping - over twice the samples: 157 -> 146-150 us at configs[3]'s shard; 32 would not fit 128 registers), 8 for the 2-operator layout
#ifndef SOTS_OL_U2
#define SOTS_OL_U2 8
#endif
    constexpr int U = G == 4 ? SOTS_OL_U : SOTS_OL_U2, CH = kStageChunks;
    constexpr int LAST = 2 * OPS - 1; // the block whose samples leave in trip k is k - LAST
    __shared__ float tab[kWavetableSize + 64]; // entry W repeats entry W - 1: the clamp-free index below may reach it
    __shared__ float4 stage_all[kOlTileRows * CH + 4]; // the tiles + 64 bytes of progress counters
    request_wavetable(tab, wavetable);
    bool table_pending = true;
    const float c = (float)kWavetableSize / (float)SOTS_SAMPLE_RATE; // w2srRatio, Evolutionary_Strategy.hpp:203
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), waves = blockDim.x / kWave;
    const uint32_t lr = lane & 15u, grp = lr / (uint32_t)R;        // this lane's place among the G lanes of its individual
    const int s = (int)grp - (G - OPS);                            // ... i.e. its operator; -1: the 3-operator voice's constant source
    const uint32_t li = (lane >> 4) * R + (lr - grp * R);          // ... of which individual of the wavefront
    const uint32_t rows_per_block = waves * IPW;
    const uint32_t sharers = (waves + 3u) / 4u; // the wavefronts of a workgroup go to the SIMDs in turn: w, w + 4, ... share one
    float4 *__restrict__ stage = stage_all + wave * (IPW * CH);
    // write side: the last operator's samples are spread over the G lanes of their individual first (row shifts), so that
    // EVERY lane parks 32 / G bytes with one instruction; row = individual, chunk q in slot q ^ (row & 7)
    const uint32_t l7 = li & 7u;
    const uint32_t r8 = lane / CH, rch = lane % CH; // read side: lane = (row within a group of 8, chunk)
    const float4 *__restrict__ rd = stage + r8 * CH + (rch ^ (r8 & 7u));
    const uint32_t lane_off = r8 * pitch + 4u * rch;
    uint32_t row_off[NI];
#pragma unroll
    for (int g = 0; g < NI; ++g) row_off[g] = (lane_off + (uint32_t)g * 8u * pitch) * 4u; // bytes; < 2^32: 32 rows of at most 8224 floats

    // progress counters of the wavefronts (below): 64 bytes behind the workgroup's tiles
    int *__restrict__ progress = reinterpret_cast<int *>(stage_all + rows_per_block * CH);
    const bool feedback = sharers > 1u;
    const uint32_t first_base = blockIdx.x * rows_per_block;
    for (uint32_t base = first_base; base < p_len; base += gridDim.x * rows_per_block) {
        const uint32_t row0 = base + wave * IPW; // first row of this wavefront
        const uint32_t ind = row0 + li < p_len ? row0 + li : p_len - 1u;
        float p[D];
        if (var.vin) {
            // fused generation loop: the workgroup makes its individuals, a thread per gene (k_recombine_mutate's
            // arithmetic), writes them to the other half and leaves the values in LDS for the lanes that need them
            // (the tile area: nothing is parked there yet)
            float *__restrict__ made = reinterpret_cast<float *>(stage_all);
            if (base != first_base) __syncthreads(); // the previous tile's last lines have been read back
            for (uint32_t t = threadIdx.x; t < rows_per_block * D; t += blockDim.x) {
                const uint32_t i1 = base + t / D, g1 = t % D;
                if (i1 < p_len) {
                    const uint32_t src = recombine_source(i1, g1, var.pd);
                    float x = var.vin[src], st = var.sin[src];
                    mutate_gene(x, st, var.pd.gid_base + i1, g1, var.generation, var.pd, var.mc);
                    var.vout[(size_t)i1 * D + g1] = x;
                    var.sout[(size_t)i1 * D + g1] = st;
                    made[t] = x;
                }
            }
            __syncthreads();
#pragma unroll
            for (int g = 0; g < D; ++g) p[g] = made[(ind - base) * D + g];
            __syncthreads(); // everybody has its genes: the tiles may be written
        } else {
#pragma unroll
            for (int g = 0; g < D; ++g) p[g] = values[(size_t)ind * D + g];
        }
#pragma unroll
        for (int g = 0; g < D; ++g) p[g] = sp.pmin[g] + p[g] * (sp.pmax[g] - sp.pmin[g]); // scaleParams, ocl_program.cl:297
        // operator s hands (t * pm + po) * c to operator s + 1; the last operator's t * pm is the sample
        float inc0c, pm = 0.0f, po = 0.0f;
        if constexpr (KIND == SOTS_SYNTH_2OP) { // Evolutionary_Strategy.hpp:372-401
            inc0c = c * p[0];
            pm = s == 0 ? p[0] * p[1] : p[3], po = s == 0 ? p[2] : 0.0f;
        } else { // series, :407-445 (the 4-operator voice adds one more modulator stage)
            inc0c = c * p[1];
#pragma unroll
            for (int o = 0; o < OPS; ++o)
                if (s == o) pm = p[2 * o] * p[2 * o + 1], po = o + 1 < OPS ? p[2 * o + 3] : 0.0f;
        }
        // 3 operators: the lane in front of operator 0 makes (t * 0 + p1) * c = c * p1 for it, sample after sample
        if (G > OPS && s < 0) pm = 0.0f, po = p[1];
        const float wlo = s <= 0 ? 0.0f : kWf; // operator 0 has the first wrap only
        if (table_pending) {
            wavetable_ready();
            if (threadIdx.x == 0) tab[kWavetableSize] = tab[kWavetableSize - 1];
            __syncthreads();
            table_pending = false;
        }
        // The index needs no clamp while every phase stays inside [0, W]: operator 0's increment in [0, W) (first wrap only), and
        // every handed-over increment c (t pm + po) smaller than W in magnitude - |t| <= 1, so c (|pm| + |po|) < W suffices, whatever
        // the table says (the lanes run on past their last block and before their first: only bounds that hold for every table
        // value count).  A phase that the second wrap's addition rounds up to exactly W reads entry W = entry W - 1, which is what
        // the reference's clamp makes of it (tab_at).  The reference's own parameter box qualifies (0.743 (3520 x 8 + 3520) < 32768);
        // any lane out of bounds, or NaN, sends the wavefront down the clamped path.
        const bool hands_over = s >= 0 && s < OPS - 1;
        const bool lane_bounded = (!hands_over || c * (__builtin_fabsf(pm) + __builtin_fabsf(po)) < 0.999f * kWf) && inc0c >= 0.0f && inc0c < kWf;
        const bool unclamped = __all(lane_bounded);

        auto run = [&](auto unclamped_tag) {
        constexpr bool UNCLAMPED = decltype(unclamped_tag)::value;
        float pos = 0.0f;
        float inc[U];  // this lane's increments for the block it works on next (operator 0: the constant, never overwritten)
        float T[2][U]; // table values of the block of this trip's parity / of the trip before
#pragma unroll
        for (int u = 0; u < U; ++u) inc[u] = s == 0 ? inc0c : 0.0f, T[0][u] = 0.0f, T[1][u] = 0.0f;
        float pend[NI][4];
        float *__restrict__ pend_line = audio;
        bool pending = false;
        auto store_pending = [&]() {
            if (!pending) return;
            pending = false;
#pragma unroll
            for (int g = 0; g < NI; ++g) {
                if (row0 + 8u * g + r8 >= p_len) continue;
                asm volatile("" : "+v"(row_off[g]));
                *reinterpret_cast<float4 *>(reinterpret_cast<char *>(pend_line) + row_off[g]) = make_float4(pend[g][0], pend[g][1], pend[g][2], pend[g][3]);
            }
        };
        const uint32_t nb = n / U;
        // PHASE 1: the first trips (an operator idles, increments 0, until its first block arrives; samples leave from trip LAST),
        // 0: the body, 2: the last trips (samples leave while their block exists)
        auto trip = [&](auto q_tag, auto phase_tag, uint32_t k) {
            constexpr int Q = decltype(q_tag)::value, PHASE = decltype(phase_tag)::value;
            // C: this lane's operator on its block
#pragma unroll
            for (int u = 0; u < U; ++u) {
#if SOTS_OL_ABL & 1 // timing ablations (never in the shipped build; the audio is NOT valid): no table reads
                T[Q][u] = pos * 1e-9f;
#elif SOTS_OL_ABL & 16 // conflict-free table reads (no index arithmetic either)
                T[Q][u] = tab[lane + 64 * u];
#else
                if constexpr (UNCLAMPED) {
                    T[Q][u] = tab[(uint32_t)pos]; // pos in [0, W]
                } else {
                    // table[clamp((int)pos, 0, W-1)] (tab_at): v_cvt_u32_f32 saturates - negative and NaN give 0, as the
                    // oracle's tab_at does - then one unsigned minimum; both cheaper to issue than the signed med3
                    uint32_t ti;
                    asm("v_cvt_u32_f32 %0, %1" : "=v"(ti) : "v"(pos));
                    T[Q][u] = tab[min(ti, kWavetableSize - 1u)];
                }
#endif
                pos += inc[u];
#if !(SOTS_OL_ABL & 8) // 8: no wraps
                const v2f_t bc = v2f_t{pos, pos} + v2f_t{-kWf, wlo};
                pos = __uint_as_float(min(min(__float_as_uint(bc.x), __float_as_uint(bc.y)), __float_as_uint(pos)));
#endif
            }
            // (nothing of A may move above this trip's table reads: it would wait for the reads of the trip before with none in
            // flight behind them - the scheduler did, and every trip began with s_waitcnt lgkmcnt(0))
            __builtin_amdgcn_sched_barrier(0);
            // A: the block read one trip ago (its table values have landed): t * pm (the last operator's samples), + po, * c
            v2f_t m[U / 2];
#pragma unroll
            for (int u = 0; u < U; u += 2) {
                m[u / 2] = v2f_t{T[Q ^ 1][u], T[Q ^ 1][u + 1]} * pm;
                const v2f_t o = (m[u / 2] + po) * c;
                // B: to the operator behind (lanes R ... 15 of every row take from R lanes below; lanes 0 ... R-1 have no source lane
                // and keep theirs: operator 0's constant)
#if SOTS_OL_ABL & 2 // no lane shift
                inc[u] = o.x, inc[u + 1] = o.y;
#else
                inc[u] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(inc[u]), __float_as_int(o.x), 0x110 + R, 0xf, 0xf, false));
                inc[u + 1] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(inc[u + 1]), __float_as_int(o.y), 0x110 + R, 0xf, 0xf, false));
#endif
            }
            if constexpr (PHASE == 1) {
                if (s >= 1 && 2u * (uint32_t)s > k + 1u) { // this operator's first block has not arrived yet
#pragma unroll
                    for (int u = 0; u < U; ++u) inc[u] = 0.0f;
                }
            }
            // the samples of block k - LAST leave
#if SOTS_OL_ABL & 4 // no tile, no stores
            if (pm == 12345.678f && m[0].x == 1.0f && m[1].y == 2.0f && m[2].x == 5.0f && m[3].y == 0.1f) {
#else
            if (PHASE == 0 || (k >= (uint32_t)LAST && k - (uint32_t)LAST < nb)) {
#endif
                const uint32_t ip = (k - (uint32_t)LAST) * U;
                store_pending(); // the lines read back one trip ago
                const uint32_t c0 = (ip >> 2) & (CH - 1);
                // the 8 samples sit in the last operator's lanes (the top R lanes of every row): lane group g of the individual takes
                // samples 8 g / G ... from them (row_shl: a lane reads the lane n above it; lanes without a source keep theirs), then
                // all 64 lanes write (a write instruction costs the LDS the same whether 16 lanes carry data or 64: 26 -> 6 cycles per sample and CU)
                auto shl = [&](float keep, float src, auto n_tag) {
                    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(src), 0x100 + decltype(n_tag)::value, 0xf, 0xf, false));
                };
                if constexpr (G == 4 && U == 16) { // 16-sample trips: a whole chunk of 16 bytes per lane
                    float4 w = make_float4(m[6].x, m[6].y, m[7].x, m[7].y);
                    w.x = shl(w.x, m[4].x, ic<R>{}), w.y = shl(w.y, m[4].y, ic<R>{}), w.z = shl(w.z, m[5].x, ic<R>{}), w.w = shl(w.w, m[5].y, ic<R>{});
                    w.x = shl(w.x, m[2].x, ic<2 * R>{}), w.y = shl(w.y, m[2].y, ic<2 * R>{}), w.z = shl(w.z, m[3].x, ic<2 * R>{}), w.w = shl(w.w, m[3].y, ic<2 * R>{});
                    w.x = shl(w.x, m[0].x, ic<3 * R>{}), w.y = shl(w.y, m[0].y, ic<3 * R>{}), w.z = shl(w.z, m[1].x, ic<3 * R>{}), w.w = shl(w.w, m[1].y, ic<3 * R>{});
                    stage[li * CH + ((c0 + grp) ^ l7)] = w;
                } else if constexpr (G == 4) {
                    v2f_t w = m[3];
                    w.x = shl(w.x, m[2].x, ic<R>{}), w.y = shl(w.y, m[2].y, ic<R>{});
                    w.x = shl(w.x, m[1].x, ic<2 * R>{}), w.y = shl(w.y, m[1].y, ic<2 * R>{});
                    w.x = shl(w.x, m[0].x, ic<3 * R>{}), w.y = shl(w.y, m[0].y, ic<3 * R>{});
                    float2 *__restrict__ wr2 = reinterpret_cast<float2 *>(stage + li * CH);
                    wr2[((c0 + (grp >> 1)) ^ l7) * 2u + (grp & 1u)] = make_float2(w.x, w.y);
                } else if constexpr (U == 16) { // two lanes per individual, 16-sample trips: two chunks per lane
                    float4 w0 = make_float4(m[4].x, m[4].y, m[5].x, m[5].y), w1 = make_float4(m[6].x, m[6].y, m[7].x, m[7].y);
                    w0.x = shl(w0.x, m[0].x, ic<R>{}), w0.y = shl(w0.y, m[0].y, ic<R>{}), w0.z = shl(w0.z, m[1].x, ic<R>{}), w0.w = shl(w0.w, m[1].y, ic<R>{});
                    w1.x = shl(w1.x, m[2].x, ic<R>{}), w1.y = shl(w1.y, m[2].y, ic<R>{}), w1.z = shl(w1.z, m[3].x, ic<R>{}), w1.w = shl(w1.w, m[3].y, ic<R>{});
                    stage[li * CH + ((c0 + 2u * grp) ^ l7)] = w0;
                    stage[li * CH + ((c0 + 2u * grp + 1u) ^ l7)] = w1;
                } else {
                    float4 w = make_float4(m[2].x, m[2].y, m[3].x, m[3].y);
                    w.x = shl(w.x, m[0].x, ic<R>{}), w.y = shl(w.y, m[0].y, ic<R>{}), w.z = shl(w.z, m[1].x, ic<R>{}), w.w = shl(w.w, m[1].y, ic<R>{});
                    stage[li * CH + ((c0 + grp) ^ l7)] = w;
                }
                if (c0 == CH - U / 4) { // 32 samples parked: the tile is read back transposed (LDS works in order) and stored in the next trip
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int g = 0; g < NI; ++g) {
                        const float4 q = rd[g * (8 * CH)];
                        pend[g][0] = q.x, pend[g][1] = q.y, pend[g][2] = q.z, pend[g][3] = q.w;
                    }
                    pend_line = audio + (size_t)row0 * pitch + (ip + U - 4 * CH); // wavefront-uniform
                    pending = true;
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("" ::: "memory");
                }
            }
        };
        if (feedback && lane == 0) progress[wave] = 0; // (a stale count of the tile before only costs a priority)
        constexpr uint32_t K0 = 2 * OPS; // first trip (even) in which every operator works and samples leave
        SOTS_STAMP_SCOPE(blockIdx.x * (blockDim.x / kWave) + wave); // (diagnostic builds: this wavefront's cycles over all trips)
        for (uint32_t k = 0; k < K0; k += 2) {
            trip(ic<0>{}, ic<1>{}, k);
            trip(ic<1>{}, ic<1>{}, k + 1);
        }
        for (uint32_t k = K0; k < nb; k += 2) {
#ifndef SOTS_OL_NO_FEEDBACK
            // The SIMD issues for its OLDEST wavefront first: of two that share it the older runs as if alone (54 cycles per sample in
            // isolation) and the younger takes what is left (78) - and the kernel ends with the younger.  Every 16 trips a wavefront
            // publishes its trip count in LDS and takes priority = the number of wavefronts sharing its SIMD (w, w + 4, ... of the
            // workgroup) that are AHEAD of it, so the laggard issues first: 70 / 71 in isolation (tools/ubench/ol_loop.hip).
            // (Priority TURNS - a trip each, or 32 - even the two out at the slow end; s_setprio takes an immediate, hence the branches.)
            if (feedback && (k & 15u) == 0u) {
                if (lane == 0) __hip_atomic_store(&progress[wave], (int)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                uint32_t ahead = 0;
                for (uint32_t q = 1; q < sharers; ++q) {
                    const uint32_t other = (wave + 4u * q) % (4u * sharers);
                    ahead += other < waves && __hip_atomic_load(&progress[other], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > (int)k ? 1u : 0u;
                }
                ahead = __builtin_amdgcn_readfirstlane(ahead);
                asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 10f\n\ts_cmp_eq_u32 %0, 1\n\ts_cbranch_scc1 11f\n\ts_cmp_eq_u32 %0, 2\n\ts_cbranch_scc1 12f\n\t"
                             "s_setprio 3\n\ts_branch 19f\n10:\ts_setprio 0\n\ts_branch 19f\n11:\ts_setprio 1\n\ts_branch 19f\n12:\ts_setprio 2\n19:" ::"s"(ahead));
            }
#endif
            trip(ic<0>{}, ic<0>{}, k);
            trip(ic<1>{}, ic<0>{}, k + 1);
        }
        for (uint32_t k = nb; k < nb + K0; k += 2) { // the last block leaves in trip nb - 1 + LAST
            trip(ic<0>{}, ic<2>{}, k);
            trip(ic<1>{}, ic<2>{}, k + 1);
        }
        store_pending();
        };
#ifdef SOTS_OL_CLAMPED // (experiment: the clamped index for every wavefront)
        run(std::false_type{});
#else
        if (unclamped) run(std::true_type{});
        else run(std::false_type{});
#endif
    }
    if (table_pending) wavetable_ready(); // a workgroup without a tile must not end with copies in flight
}

// ------------------------------------------------------------------------------------
// SMALL populations (a few individuals per CU), every voice: the time axis in the lanes.
//
// k_synth above puts an individual in a lane and pays its ten to twenty vector instructions per sample and operator whether
// one lane is alive or sixty-four; with a few individuals per CU almost all of that is arithmetic on dead lanes.  But only the
// PHASE of an operator is a recurrence over the samples - pos[n+1] = wrap(pos[n] + inc[n]), three dependent instructions -
// while everything else (the table read at pos[n], t mul + off, times c, times the gain) is independent per sample.  So
// operator s has TWO wavefronts here, and per block of 64 samples
//   * its SCAN wavefront, lane = individual, reads the 64 increments of its individual (operator 0: a constant) and writes
//     the 64 phases in their place - the reference's additions and wraps, in its order;
//   * its EVALUATION wavefront, lane = SAMPLE, takes individual after individual: table value at the phase, then either the
//     next operator's increment c (t mul + off) into that operator's buffer or gain t to the audio row (a whole 256-byte
//     piece of a row per instruction).
// A pipeline of 2 OPS stages, one block and one workgroup barrier per tick, three buffers per operator in rotation (the
// evaluation of block b reads a buffer in the tick in which block b + 2's increments are written).  Every sample sees exactly the
// reference's operations, unfused: bit-identical to k_synth and the oracle.  What is left per sample is the latency of the
// three dependent instructions of a scan (~30 cycles; k_synth: ~90): profiles/r03_experiments.md.
// ------------------------------------------------------------------------------------
constexpr int kTpRow = 64 + 4; // floats per (individual, block) row: the scan's 16-byte accesses of neighbouring lanes on different banks
// individuals per workgroup: three rows per operator each (and, for the voice of three parallel chains, two rows for each of
// the two products that wait for the third), in the 31 KiB beside the table that the genes leave: 19 / 12 / 9 for 2, 3, 4
// operators in series (populations up to 4864 / 3072 / 2304 on 256 CUs), 5 for the triple voice (1280)
template <int KIND> constexpr int tp_max_individuals()
{
    return (31 * 1024) / ((VoiceShape<KIND>::J * VoiceShape<KIND>::OPS * 3 + (VoiceShape<KIND>::J > 1 ? 2 * (VoiceShape<KIND>::J - 1) : 0)) * kTpRow * 4);
}

// evaluating wavefronts per operator: two for the series voices (from about eight individuals per CU the evaluation of a
// block takes longer than its scan: the two take every other batch of four), one for the triple voice (six operators)
template <int KIND> constexpr int tp_eval_waves() { return VoiceShape<KIND>::J == 1 ? 2 : 1; }
template <int KIND> constexpr int tp_waves() { return VoiceShape<KIND>::J * VoiceShape<KIND>::OPS * (1 + tp_eval_waves<KIND>()); }

template <int KIND>
__global__ __launch_bounds__(tp_waves<KIND>() * kWave) void k_synth_tp(const float *__restrict__ values,
                                                                                const float *__restrict__ wavetable,
                                                                                float *__restrict__ audio, SynthParams sp,
                                                                                uint32_t p_len, uint32_t n, uint32_t pitch,
                                                                                uint32_t per_group, Variation var)
{
    // J > 1 (the voice of three 2-operator chains, averaged): every chain has its own pipeline; chains 1 ... J-1 leave gain t
    // in LDS and chain 0, which runs one tick behind them, adds the products in the reference's order and divides
    constexpr int J = VoiceShape<KIND>::J, OPS = VoiceShape<KIND>::OPS, D = VoiceShape<KIND>::D, IMAX = tp_max_individuals<KIND>();
    constexpr int NE = tp_eval_waves<KIND>();
    constexpr uint32_t THREADS = tp_waves<KIND>() * kWave;
    __shared__ float tab[kWavetableSize];
    __shared__ __attribute__((aligned(16))) float buf[J * OPS][3][IMAX][kTpRow]; // [chain, operator][block mod 3][individual][sample]: increments, then phases
    __shared__ float prod[J > 1 ? J - 1 : 1][2][J > 1 ? IMAX : 1][J > 1 ? kTpRow : 1]; // [chain - 1][block parity]: gain t of chains 1 ... J-1
    __shared__ float made[IMAX * D];
    request_wavetable(tab, wavetable);
    const float c = (float)kWavetableSize / (float)SOTS_SAMPLE_RATE; // w2srRatio, Evolutionary_Strategy.hpp:203
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const int wave = (int)__builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const bool scans = wave < J * OPS; // the first J OPS wavefronts scan, the others evaluate (NE per operator)
    const int jo = scans ? wave : (wave - J * OPS) / NE, jc = jo / OPS, s = jo % OPS; // this wavefront's chain and operator
    const uint32_t ev = scans ? 0u : (uint32_t)((wave - J * OPS) % NE);                // ... and which of its evaluators
    const uint32_t first = blockIdx.x * per_group;
    const uint32_t count = first >= p_len ? 0u : (p_len - first < per_group ? p_len - first : per_group); // individuals here (<= IMAX)

    // the individuals: read, or made here (fused generation loop: one gene per thread)
    if (var.vin) {
        for (uint32_t t = threadIdx.x; t < count * D; t += THREADS) {
            const uint32_t i1 = first + t / D, g1 = t % D;
            const uint32_t src = recombine_source(i1, g1, var.pd);
            float x = var.vin[src], st = var.sin[src];
            mutate_gene(x, st, var.pd.gid_base + i1, g1, var.generation, var.pd, var.mc);
            var.vout[(size_t)i1 * D + g1] = x;
            var.sout[(size_t)i1 * D + g1] = st;
            made[t] = x;
        }
    } else {
        for (uint32_t t = threadIdx.x; t < count * D; t += THREADS) made[t] = values[(size_t)first * D + t];
    }
    __syncthreads();
    // lane i < count: individual first + i.  scaleParams and the per-operator constants as in k_synth
    float pr[D];
#pragma unroll
    for (int g = 0; g < D; ++g) {
        pr[g] = lane < count ? made[lane * D + g] : 0.0f;
        const int sc = KIND == SOTS_SYNTH_TRIPLE_PAR ? (g & 3) : g; // the triple voice scales all three chains by entries 0..3, Evolutionary_Strategy.hpp:453-455
        pr[g] = sp.pmin[sc] + pr[g] * (sp.pmax[sc] - sp.pmin[sc]); // min + v*(max-min), ocl_program.cl:297
    }
    float inc0, mul_next = 0.0f, off_next = 0.0f, gain; // mul / off: what the evaluation of operator s applies for operator s + 1
    if constexpr (KIND == SOTS_SYNTH_2OP) { // Evolutionary_Strategy.hpp:372-401
        inc0 = c * pr[0];
        if (s == 0) mul_next = pr[0] * pr[1], off_next = pr[2];
        gain = pr[3];
    } else if constexpr (KIND == SOTS_SYNTH_TRIPLE_PAR) { // :457-494
        inc0 = 0.0f, gain = 0.0f;
#pragma unroll
        for (int jj = 0; jj < J; ++jj)
            if (jc == jj) {
                inc0 = c * pr[4 * jj], gain = pr[4 * jj + 3];
                if (s == 0) mul_next = pr[4 * jj] * pr[4 * jj + 1], off_next = pr[4 * jj + 2];
            }
    } else { // series, :407-445 (the 4-operator voice adds one more modulator stage)
        inc0 = c * pr[1];
#pragma unroll
        for (int o = 1; o < OPS; ++o)
            if (s == o - 1) mul_next = pr[2 * (o - 1)] * pr[2 * (o - 1) + 1], off_next = pr[2 * (o - 1) + 3];
        gain = pr[2 * (OPS - 1)] * pr[2 * (OPS - 1) + 1];
    }
    wavetable_ready();

    const uint32_t blocks = n / kWave;
    float pos = 0.0f; // (scan wavefronts) this operator's phase of individual `lane`
    constexpr uint32_t LAG0 = J > 1 ? 1u : 0u; // chain 0 of a parallel voice: one tick behind the others
    for (uint32_t tick = 0; tick < blocks + 2 * OPS - 1 + LAG0; ++tick) {
        // scan of operator s: block tick - 2 s; its evaluation: one tick later
        const uint32_t k = tick - 2u * (uint32_t)s - (scans ? 0u : 1u) - (jc == 0 ? LAG0 : 0u);
        if (k < blocks) {
            float(*rows)[kTpRow] = buf[jo][k % 3u];
            if (scans) {
                if (lane < count) { // lane = individual: phases in place of increments, four samples per access
                    float4 *row = reinterpret_cast<float4 *>(rows[lane]);
                    if (s == 0) {
#pragma unroll
                        for (int q = 0; q < kWave / 4; ++q) {
                            float4 o;
                            o.x = pos, pos += inc0, wrap_hi(pos);
                            o.y = pos, pos += inc0, wrap_hi(pos);
                            o.z = pos, pos += inc0, wrap_hi(pos);
                            o.w = pos, pos += inc0, wrap_hi(pos);
                            row[q] = o;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < kWave / 4; ++q) {
                            const float4 v = row[q];
                            float4 o;
                            o.x = pos, pos += v.x, wrap_both(pos);
                            o.y = pos, pos += v.y, wrap_both(pos);
                            o.z = pos, pos += v.z, wrap_both(pos);
                            o.w = pos, pos += v.w, wrap_both(pos);
                            row[q] = o;
                        }
                    }
                }
            } else { // lane = sample
                const uint32_t sample = k * kWave + lane;
                // four individuals at a time: their phases, then their table values, are asked for together
                float(*next)[kTpRow] = buf[s + 1 < OPS ? jo + 1 : 0][k % 3u];
                for (uint32_t i0 = 4u * ev; i0 < count; i0 += 4u * NE) {
                    float ph[4], t[4];
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) ph[j] = rows[i0 + j < count ? i0 + j : i0][lane];
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) t[j] = tab_at<true>(tab, ph[j]);
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        const uint32_t i = i0 + j;
                        if (i >= count) break;
                        if (s < OPS - 1) {
                            const float m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mul_next), (int)i));
                            const float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(off_next), (int)i));
                            next[i][lane] = c * (t[j] * m + o); // the next operator's increment, unfused
                        } else {
                            const float g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gain), (int)i));
                            if constexpr (J == 1) audio[(size_t)(first + i) * pitch + sample] = g * t[j];
                            else if (jc != 0) prod[jc - 1][k & 1u][i][lane] = g * t[j];
                            else // the reference's sum, in its order, and its division (== (float)(double(sum) / 3.0), :493)
                                audio[(size_t)(first + i) * pitch + sample] = (g * t[j] + prod[0][k & 1u][i][lane] + prod[J > 2 ? 1 : 0][k & 1u][i][lane]) / 3.0f;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// applyWindowPopulation, ocl_program.cl:566-586.  The table is the reference's double
// window (Evolutionary_Strategy.hpp:308-317) rounded once to fp32.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_window(float *__restrict__ audio, const float *__restrict__ window,
                                                size_t total4, uint32_t log2n4, uint32_t pitch4)
{
    float4 *a4 = reinterpret_cast<float4 *>(audio);
    const float4 *w4 = reinterpret_cast<const float4 *>(window);
    const uint32_t mask = (1u << log2n4) - 1u;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4;
         e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e >> log2n4;
        const uint32_t col = (uint32_t)e & mask;
        float4 v = a4[row * pitch4 + col];
        const float4 w = w4[col];
        v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w;
        a4[row * pitch4 + col] = v;
    }
}

// ------------------------------------------------------------------------------------
// Batched real FFT (replaces clFFT, Evolutionary_Strategy_OpenCL.hpp:156-192,555-561)
// and fitnessPopulation (ocl_program.cl:594-659 with the CPU bin range k < N/2,
// Evolutionary_Strategy_CPU.hpp:235).
//
// One wavefront transforms one individual: the N real samples are read as M = N/2 complex
// points, E = M/64 per lane, and go through Stockham autosort passes of radix 8 or 4 (three
// or two radix-2 butterfly layers done in registers), exchanging through a padded LDS
// buffer between passes.  A final split step turns Z[k], Z[M-k] into the real-input bins
// X[k], X[M-k].
// ------------------------------------------------------------------------------------
// The transform is not bit-matched to anything (the oracle's FFT is fp64), so fused
// multiply-adds are allowed here and only here; "on" contracts within one expression, which
// keeps k_fft<.,1> and k_fitness on identical arithmetic.
#pragma clang fp contract(on)
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// ---- complex values as 2-vectors (register pairs): v_pk_add/mul/fma_f32 do a complex add, or half a complex
// multiply, per instruction, and every wave64 vector instruction holds the SIMD for 4 cycles whatever it does
__device__ __forceinline__ v2f_t xv(float2 a) { return v2f_t{a.x, a.y}; }
// a * w: (a.x, a.y) * w.x + (-a.y, a.x) * w.y - a packed multiply and a packed fma (operand selects and negations are
// instruction modifiers)
// The packed instructions select the low or high half of each source per result half (op_sel, op_sel_hi) and negate
// per half (neg_lo, neg_hi); the compiler uses the selects but flips signs of single halves with v_xor and copies, so
// the few shapes the transform needs are written out.
#ifndef SOTS_XC_ONE_ASM
#define SOTS_XC_ONE_ASM 1 // (0: a statement per instruction, rounds 2-3; profiles/r04_experiments.md)
#endif
__device__ __forceinline__ v2f_t xc_mul(v2f_t a, v2f_t w)
{
#if SOTS_XC_ONE_ASM // both halves in ONE statement: around a statement the compiler pads wait states it cannot rule out (s_nop)
    v2f_t t;
    asm("v_pk_mul_f32 %1, %0, %2 op_sel:[0,0] op_sel_hi:[1,0]\n\t"                                              // (a.x w.x, a.y w.x)
        "v_pk_fma_f32 %0, %0, %2, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "+v"(a), "=&v"(t) : "v"(w)); // (-a.y w.y, a.x w.y) + t
    return a;
#else
    v2f_t t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));                       // (a.x w.x, a.y w.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t)); // (-a.y w.y, a.x w.y) + t
    return r;
#endif
}
__device__ __forceinline__ v2f_t xc_mul_neg_i(v2f_t a) { return v2f_t{a.y, -a.x}; }
// (-i a) * w = (a.y w.x + a.x w.y, a.y w.y - a.x w.x)
__device__ __forceinline__ v2f_t xc_mul_negi_w(v2f_t a, v2f_t w)
{
#if SOTS_XC_ONE_ASM
    v2f_t t;
    asm("v_pk_mul_f32 %1, %0, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"                                  // (a.y w.x, -a.x w.x)
        "v_pk_fma_f32 %0, %0, %2, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a), "=&v"(t) : "v"(w));               // (a.x w.y, a.y w.y) + t
    return a;
#else
    v2f_t t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));           // (a.y w.x, -a.x w.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));        // (a.x w.y, a.y w.y) + t
    return r;
#endif
}
// a + conj(b), a - conj(b)
__device__ __forceinline__ v2f_t xc_add_conj(v2f_t a, v2f_t b)
{
    v2f_t r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f_t xc_sub_conj(v2f_t a, v2f_t b)
{
    v2f_t r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}


// a * e^{-i pi/4} and a * e^{-3 i pi/4} for a = (x, y): ((x + y) s, (y - x) s) and ((y - x) s, -(x + y) s), s = sqrt(1/2) -
// one packed add with swapped and negated halves and one packed multiply each (the same sums and products as the scalar
// form, so the same bits)
__device__ __forceinline__ float2 rot_m45(float2 a)
{
    const v2f_t av = xv(a);
    v2f_t t;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(t) : "v"(av)); // (x + y, y - x)
    t = t * v2f_t{0.70710678118654752440f, 0.70710678118654752440f};
    return make_float2(t.x, t.y);
}
__device__ __forceinline__ float2 rot_m135(float2 a)
{
    const v2f_t av = xv(a);
    v2f_t t;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(t) : "v"(av)); // (y - x, -x - y)
    t = t * v2f_t{0.70710678118654752440f, 0.70710678118654752440f};
    return make_float2(t.x, t.y);
}

// a + (-i) d = (a.x + d.y, a.y - d.x) and a - (-i) d = (a.x - d.y, a.y + d.x) in ONE packed add each (half selects and negations
// are instruction modifiers): a butterfly's rotation by -i never becomes a register shuffle
__device__ __forceinline__ float2 add_negi(float2 a, float2 d)
{
    v2f_t r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(xv(a)), "v"(xv(d)));
    return make_float2(r.x, r.y);
}
__device__ __forceinline__ float2 sub_negi(float2 a, float2 d)
{
    v2f_t r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(xv(a)), "v"(xv(d)));
    return make_float2(r.x, r.y);
}

template <int R> struct Dft;
template <> struct Dft<2> {
    static __device__ __forceinline__ void run(float2 *v)
    {
        const float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};
template <> struct Dft<4> {
    static __device__ __forceinline__ void run(float2 *v)
    {
        const float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
        const float2 t2 = cadd(v[1], v[3]), d = csub(v[1], v[3]);
        v[0] = cadd(t0, t2);
        v[1] = add_negi(t1, d); // t1 + (-i) d
        v[2] = csub(t0, t2);
        v[3] = sub_negi(t1, d);
    }
};
template <> struct Dft<8> {
    static __device__ __forceinline__ void run(float2 *v)
    {
        float2 e[4] = {v[0], v[2], v[4], v[6]};
        float2 o[4] = {v[1], v[3], v[5], v[7]};
        Dft<4>::run(e);
        Dft<4>::run(o);
        // o[k] *= exp(-2 pi i k / 8); k = 2 (a rotation by -i) folded into its butterfly
        o[1] = rot_m45(o[1]);
        o[3] = rot_m135(o[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = k == 2 ? add_negi(e[k], o[k]) : cadd(e[k], o[k]);
            v[k + 4] = k == 2 ? sub_negi(e[k], o[k]) : csub(e[k], o[k]);
        }
    }
};

// LDS index padding: one extra complex slot per 8 keeps the stride-8 / stride-64 writes of
// the first two passes and the unit-stride reads off each other's banks.
__device__ __forceinline__ int lds_pad(int i) { return i + (i >> 3); }

// One Stockham pass of radix R with NS = product of the radices already applied.
// Register slot s holds element lane + 64 s of the pass input; butterfly b uses slots
// b + t*(E/R), t < R.  Output element t of butterfly j goes to (j-k)*R + k + t*NS with
// k = j mod NS.  The twiddles e^{-2 pi i t k / (NS R)} depend only on the lane, so they are
// loop-invariant per kernel: twr (when non-null) holds them in registers, B*(R-1) values in
// (b, t) order; otherwise they come from the N-entry table e^{-2 pi i q / N}, N = 2M.
template <int M, int R, int NS>
__device__ __forceinline__ void load_pass_twiddles(float2 *twr, const float2 *__restrict__ tw, int lane)
{
    constexpr int E = M / kWave, B = E / R, stride = (2 * M) / (NS * R);
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int k = (lane + kWave * b) & (NS - 1);
#pragma unroll
        for (int t = 1; t < R; ++t) twr[b * (R - 1) + t - 1] = tw[t * k * stride];
    }
}

// LAST: the pass's outputs stay in registers instead of going to LDS.  For the last pass
// (NS * R == M, so k == j and j0 == j) output t of butterfly b is element lane + 64 (b + B t): slot
// b + B t of x, i.e. afterwards x[s] = Z[lane + 64 s].
template <int M, int R, int NS, bool LAST = false>
__device__ __forceinline__ void fft_pass(float2 (&x)[M / kWave], float2 *__restrict__ lds,
                                         const float2 *__restrict__ tw, const float2 *twr, int lane)
{
    constexpr int E = M / kWave, B = E / R;
    static_assert(!LAST || NS * R == M, "only the final pass can stay in registers");
    float2 out[LAST ? E : 1];
    static_assert(E % R == 0, "radix must divide the per-lane element count");
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int j = lane + kWave * b;
        const int k = j & (NS - 1);
        float2 v[R];
#pragma unroll
        for (int t = 0; t < R; ++t) v[t] = x[b + t * B];
        if constexpr (NS > 1) {
            constexpr int stride = (2 * M) / (NS * R);
#pragma unroll
            for (int t = 1; t < R; ++t) v[t] = cmul(v[t], twr ? twr[b * (R - 1) + t - 1] : tw[t * k * stride]);
        }
        Dft<R>::run(v);
        if constexpr (LAST) {
#pragma unroll
            for (int t = 0; t < R; ++t) out[b + t * B] = v[t];
        } else {
            const int j0 = (j - k) * R + k;
#pragma unroll
            for (int t = 0; t < R; ++t) lds[lds_pad(j0 + t * NS)] = v[t];
        }
    }
    if constexpr (LAST) {
#pragma unroll
        for (int sl = 0; sl < E; ++sl) x[sl] = out[sl];
    }
}

template <int M>
__device__ __forceinline__ void lds_reload(float2 (&x)[M / kWave], const float2 *__restrict__ lds, int lane)
{
#pragma unroll
    for (int s = 0; s < M / kWave; ++s) x[s] = lds[lds_pad(lane + kWave * s)];
}

// First pass (NS = 1, no twiddles) straight from the 16-byte loads of the row.  Lane l holds
// the float4 = two complex points at pair index l + 64 h, i.e. complex elements
// e0 = 2l + 128 h and e0 + 1.  With element e = j + (M/R) t:
//   B = E/R >= 2: the lane already owns every t of butterflies j = 2l + 64 bb (+1), bb even;
//   B == 1      : lanes l and l+32 hold the even-t and odd-t halves of butterflies 2l, 2l+1;
//                 one v_permlane32_swap per register gives lane l all of 2l and lane l+32
//                 all of 2l+1.
template <int M, int R>
__device__ __forceinline__ void fft_first_pass(const float4 (&q)[M / kWave / 2], float2 *__restrict__ lds, int lane)
{
    constexpr int E = M / kWave;
    static_assert(E == R, "one first-pass butterfly per lane (N = 512: radix 4, N = 1024: radix 8)");
    float2 v[R];
#pragma unroll
    for (int h = 0; h < R / 2; ++h) {
        const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(q[h].x), __float_as_uint(q[h].z), false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(q[h].y), __float_as_uint(q[h].w), false, false);
        v[2 * h] = make_float2(__uint_as_float(sx[0]), __uint_as_float(sy[0]));
        v[2 * h + 1] = make_float2(__uint_as_float(sx[1]), __uint_as_float(sy[1]));
    }
    Dft<R>::run(v);
    const int j = 2 * (lane & 31) + (lane >> 5);
#pragma unroll
    for (int t = 0; t < R; ++t) lds[lds_pad(j * R + t)] = v[t];
}

// All passes for M complex points (N = 512: 4.4.4.4, N = 1024: 8.8.8), starting from the row as
// loaded; the last pass stays in registers: x[s] = Z[lane + 64 s].  The per-lane pass twiddles
// (all passes after the first) are loop-invariant and live in registers.
template <int M> constexpr int tw_count() { return M == 256 ? 9 : 14; }

template <int M>
__device__ __forceinline__ void preload_twiddles(float2 (&twr)[tw_count<M>()], const float2 *__restrict__ tw, int lane)
{
    static_assert(M == 256 || M == 512, "wavefront-per-row FFT is for N <= 1024");
    if constexpr (M == 256) {
        load_pass_twiddles<M, 4, 4>(&twr[0], tw, lane);
        load_pass_twiddles<M, 4, 16>(&twr[3], tw, lane);
        load_pass_twiddles<M, 4, 64>(&twr[6], tw, lane);
    } else {
        load_pass_twiddles<M, 8, 8>(&twr[0], tw, lane);
        load_pass_twiddles<M, 8, 64>(&twr[7], tw, lane);
    }
}

// Between the passes of ONE wavefront's transform.  A workgroup of one wavefront: __syncthreads(), of which the compiler
// drops the s_barrier and keeps the fence (s_waitcnt lgkmcnt(0)).  A workgroup of several wavefronts, each with its own
// piece of LDS (W > 1 below): the fence written out - a barrier would tie the wavefronts together for nothing.  (The LDS
// executes a wavefront's instructions in order; ordering them by a compiler barrier alone measures the same,
// profiles/r03_experiments.md.)
template <bool ALONE>
__device__ __forceinline__ void wave_lds_sync()
{
    if constexpr (ALONE) __syncthreads();
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int M, bool ALONE = true>
__device__ __forceinline__ void fft_forward(const float4 (&q)[M / kWave / 2], float2 *__restrict__ lds,
                                            const float2 *__restrict__ tw, const float2 (&twr)[tw_count<M>()], int lane,
                                            float2 (&x)[M / kWave])
{
#define SOTS_SYNC() wave_lds_sync<ALONE>()
#define SOTS_FIRST(R)                        \
    fft_first_pass<M, R>(q, lds, lane);      \
    SOTS_SYNC();
#define SOTS_PASS(R, NS, OFF)                             \
    fft_pass<M, R, NS>(x, lds, tw, &twr[OFF], lane);      \
    SOTS_SYNC();
#define SOTS_NEXT()                          \
    lds_reload<M>(x, lds, lane);             \
    SOTS_SYNC();
#define SOTS_LAST(R, NS, OFF) fft_pass<M, R, NS, true>(x, lds, tw, &twr[OFF], lane);
    if constexpr (M == 256) {
        SOTS_FIRST(4) SOTS_NEXT() SOTS_PASS(4, 4, 0) SOTS_NEXT() SOTS_PASS(4, 16, 3) SOTS_NEXT() SOTS_LAST(4, 64, 6)
    } else {
        SOTS_FIRST(8) SOTS_NEXT() SOTS_PASS(8, 8, 0) SOTS_NEXT() SOTS_LAST(8, 64, 7)
    }
#undef SOTS_SYNC
#undef SOTS_FIRST
#undef SOTS_PASS
#undef SOTS_NEXT
#undef SOTS_LAST
}

// Real-input split for the pair (k, M-k), 0 <= k < M/2:
//   Ee = (Z[k] + conj Z[M-k]) / 2,  Oo = -i (Z[k] - conj Z[M-k]) / 2,  T = e^{-2 pi i k/N} Oo
//   X[k] = Ee + T,  X[M-k] = conj(Ee - T)
// With Z[M] read as Z[0] the same formula gives X[0] = Re Z0 + Im Z0 and the Nyquist bin
// X[M] = Re Z0 - Im Z0 for k = 0, so no lane takes a different path.  Bin M/2, which no pair
// covers, is X[M/2] = conj Z[M/2].
// After the last pass lane l holds z[s] = Z[l + 64 s].  For k = l + 64 q the partner
// Z[M-k] = Z[(64-l) + 64 (E-1-q)] is slot E-1-q of lane 64-l: one ds_bpermute per dword through
// the LDS crossbar, no LDS memory and no bank conflicts (this replaces a write of the whole
// transform to LDS and two reads of it).  Lane 0 pairs with itself one slot further:
// Z[M - 64 q] = its own slot E-q, and Z[M] = Z[0] for q = 0.
template <int M>
__device__ __forceinline__ float2 split_partner(const float2 (&z)[M / kWave], int q, int lane, int partner_addr)
{
    constexpr int E = M / kWave;
    const float2 mine = z[q == 0 ? 0 : E - q];                    // what lane 0 needs
    const float2 send = z[E - 1 - q];                             // what lane 64-l needs from this lane
    const float px = __int_as_float(__builtin_amdgcn_ds_bpermute(partner_addr, __float_as_int(send.x)));
    const float py = __int_as_float(__builtin_amdgcn_ds_bpermute(partner_addr, __float_as_int(send.y)));
    return lane == 0 ? mine : make_float2(px, py);
}

// In packed form, without the two 1/2 factors: 2 X[k] and 2 conj X[M-k] (a factor of two is exact in fp32; the fitness
// folds it into its magnitude scale and takes magnitudes, the spectrum writer halves and conjugates as it stores) - six
// packed instructions for two bins
__device__ __forceinline__ void split_pair_2x(float2 a, float2 bz, float2 w, v2f_t &xa2, v2f_t &xbc2)
{
    const v2f_t av = xv(a), bv = xv(bz);
    const v2f_t ee = xc_add_conj(av, bv), dd = xc_sub_conj(av, bv);
    const v2f_t t = xc_mul_negi_w(dd, xv(w));
    xa2 = ee + t;
    xbc2 = ee - t;
}

// (|X| * scale - target)^2 with scale = 1 / N / windowFactor, Evolutionary_Strategy.hpp:517-519 /
// ocl_program.cl:608-611 (one combined factor: 1/N is a power of two and the window factor is 1 to an ulp).
// v_sqrt_f32 (1 ulp) instead of the correctly rounded sequence: the transform feeding it is
// fp32 against the oracle's fp64 anyway.  The fused kernels pass 2 X and scale / 2: the same value bit for bit.
__device__ __forceinline__ float bin_error(float2 x, float target, float scale)
{
    const float raw = __builtin_amdgcn_sqrtf(x.x * x.x + x.y * x.y);
    const float e = raw * scale - target;
    return e * e;
}

// Two bins at once, each with its own running sum: the magnitudes come out of v_sqrt_f32 one by one, the scale, the
// subtraction and the squared accumulation are packed (acc2 = (sum over the bins k, sum over the bins M - k); k_fft<., 1> and
// k_fitness add the two halves in the same order, so both paths still give the same fp32 sum)
__device__ __forceinline__ void bin_error2(v2f_t &acc2, float2 xa, float2 xb, float ta, float tb, float scale)
{
    const v2f_t raw = v2f_t{__builtin_amdgcn_sqrtf(xa.x * xa.x + xa.y * xa.y), __builtin_amdgcn_sqrtf(xb.x * xb.x + xb.y * xb.y)};
    const v2f_t e = raw * v2f_t{scale, scale} - v2f_t{ta, tb};
    acc2 = acc2 + e * e;
}

// Wavefront sum without LDS traffic: DPP swaps inside each row of 16 lanes (every lane of a
// row ends with the row total), then the four row totals are added in row order.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    const int m = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
    return v + __int_as_float(m);
}
__device__ __forceinline__ float wave_sum(float v)
{
    v = dpp_add<0xB1>(v);  // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);  // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v); // row_half_mirror
    v = dpp_add<0x140>(v); // row_mirror
    const int iv = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(iv, 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(iv, 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
    return ((r0 + r1) + r2) + r3;
}

// MODE 0: write spectrum rows; MODE 1: accumulate the fitness directly
// WIN: multiply by the fp32 window while loading (the generation loop then skips the window
// pass; the product is the same single fp32 rounding either way).
// Rows of N <= 1024 only (longer rows: k_fft_x).
// W: wavefronts per workgroup, each transforming rows of its own.  W = 1: workgroup b takes rows b, b + grid, ... - small
// populations, a wavefront wherever there is room.  W = 12 (N = 1024: what the registers let a CU hold, ONE workgroup per
// CU): the workgroup's rows b + t grid are dealt to its wavefronts as they ask (an LDS counter; every grid-th row, so that
// the whole GPU reads one moving window of the audio: a contiguous block of rows per workgroup is 8 % slower).  The SIMD issues for its
// oldest wavefront first: with a fixed deal the three wavefronts of a SIMD finish their equal shares one after the
// other and the last one runs alone, far below the issue rate (k_fft_x below has the numbers).
template <int LOG2N> constexpr int fft_wide_waves() { return LOG2N == 10 ? 12 : 16; }
template <int LOG2N, int MODE, bool WIN, int W = 1>
__global__ __launch_bounds__(W *kWave) void k_fft(const float *__restrict__ audio, float *__restrict__ spectrum,
                                                   const float *__restrict__ target, float *__restrict__ fitness,
                                                   const float2 *__restrict__ tw, const float *__restrict__ window,
                                                   uint32_t p_len, float inv_n, float inv_wf, uint32_t pitch)
{
    constexpr int N = 1 << LOG2N, M = N / 2, E = M / kWave, H = E / 2;
    static_assert(LOG2N == 9 || LOG2N == 10, "wavefront-per-row FFT is for N <= 1024");
    __shared__ float2 lds_all[W][M + M / 8 + 1];
    __shared__ uint32_t next_s;
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    float2 *const lds = lds_all[wave];
    const float half_scale = 0.5f * (inv_n * inv_wf); // the split below leaves a factor of two in
    const int partner_addr = ((kWave - lane) & (kWave - 1)) * 4; // ds_bpermute byte address of lane 64-l
    // buffer b0 starts with take `wave`, b1 with take W + wave; take t is row blockIdx.x + t grid
    const uint32_t grid = gridDim.x;
    uint32_t ind = blockIdx.x + wave * grid;
    if (W == 1 && ind >= p_len) return;
    if (threadIdx.x == 0) next_s = 2 * W;

    // per-lane constants of the split / fitness step, loaded once (nothing but the audio
    // prefetch is in flight inside the loop, so its waits never drain the prefetch)
    float2 w_split[H];
#pragma unroll
    for (int q = 0; q < H; ++q) w_split[q] = tw[lane + kWave * q];
    // the target spectrum sits in LDS (2N bytes): eight registers fewer than holding this lane's
    // bins, which is what keeps N = 1024 at three wavefronts per SIMD with two rows in flight
    __shared__ float tgt_s[MODE == 1 ? M + 1 : 1];
    if constexpr (MODE == 1) {
        for (int k = threadIdx.x; k < M; k += W * kWave) tgt_s[k] = target[k];
    }
    if constexpr (MODE == 1 || W > 1) __syncthreads(); // (the only workgroup barrier: target and row counter are there)
    if (ind >= p_len) return;

    // rows are read 16 bytes per lane: pair index lane + 64 h holds complex points 2(lane+64h), +1
    constexpr int Q = E / 2;
    float4 wv[WIN ? Q : 1];
    if constexpr (WIN) {
#pragma unroll
        for (int h = 0; h < Q; ++h) wv[h] = reinterpret_cast<const float4 *>(window)[lane + kWave * h];
    }
    float2 twr[tw_count<M>()];
    preload_twiddles<M>(twr, tw, lane);
    // Three row buffers rotate (no copies): two rows are in flight beside the one being
    // transformed - a row's transform is shorter than the loaded memory latency, and the registers
    // are there (168 = three wavefronts per SIMD).
    auto request = [&](float4 (&dst)[Q], uint32_t r) { // rows past the end re-read the current one
#ifdef SOTS_ABL_FFT_CACHED_ROWS
        const float4 *__restrict__ in = reinterpret_cast<const float4 *>(audio + (size_t)((r < p_len ? r : ind) & 511u) * pitch); // timing ablation: rows from L2
#else
        const float4 *__restrict__ in = reinterpret_cast<const float4 *>(audio + (size_t)(r < p_len ? r : ind) * pitch);
#endif
#pragma unroll
        for (int h = 0; h < Q; ++h) dst[h] = SOTS_ROW_LOAD(in + lane + kWave * h);
    };
    // transforms the row in `cur` (individual `ind`) after requesting row ind + 2 grid into `fill`
#ifdef SOTS_STAMP
    unsigned long long st_wait = 0, st_fft = 0, st_tail = 0, st_rows = 0;
    const uint32_t st_slot = blockIdx.x * W + wave;
    const unsigned long long st_begin = __builtin_amdgcn_s_memrealtime(); // 100 MHz, common to the whole chip
    const unsigned long long st_begin_clk = __builtin_amdgcn_s_memtime();
#define SOTS_FFT_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define SOTS_FFT_T(var)
#endif
#ifdef SOTS_STAMP_ENDS
    // light stamps (no per-phase timers: the register allocation stays the product's): when every wavefront starts its
    // first row and ends, and how many rows it took (tools/fft_ends_probe.py)
    const unsigned long long se_begin = __builtin_amdgcn_s_memrealtime();
    uint32_t se_rows = 0;
#endif
    uint32_t pend = 0, dealt = 2; // the take on its way (lane 0) / W = 1: takes so far
    auto next_take = [&]() {
        if constexpr (W == 1) return dealt++;
        else {
            const uint32_t t = __builtin_amdgcn_readfirstlane(pend);
            if (lane == 0) pend = atomicAdd(&next_s, 1u); // answered during this row's transform
            return t;
        }
    };
    if constexpr (W > 1) {
        if (lane == 0) pend = atomicAdd(&next_s, 1u);
    }
    // transforms the row in `cur` (row `ind`) after requesting this wavefront's next take into `fill`; returns that row
    auto process = [&](float4 (&cur)[Q], float4 (&fill)[Q]) -> uint32_t {
        SOTS_FFT_T(t0);
#ifdef SOTS_STAMP
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Q) : "memory"); // the row in `cur` has landed (the newer request may still fly)
#endif
        SOTS_FFT_T(t1);
        if constexpr (WIN) {
#pragma unroll
            for (int h = 0; h < Q; ++h) { // two packed multiplies per 16 bytes
                const v2f_t lo = v2f_t{cur[h].x, cur[h].y} * v2f_t{wv[h].x, wv[h].y}, hi = v2f_t{cur[h].z, cur[h].w} * v2f_t{wv[h].z, wv[h].w};
                cur[h] = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
        }
        const uint32_t filled = blockIdx.x + next_take() * grid;
        request(fill, filled);
#ifdef SOTS_STAMP_ENDS
        ++se_rows;
#endif
        float2 z[E]; // z[s] = Z[lane + 64 s]
        fft_forward<M, W == 1>(cur, lds, tw, twr, lane, z);
        SOTS_FFT_T(t2);
        // bin M/2 = conj Z[M/2]; Z[M/2] = Z[0 + 64 (E/2)] is lane 0's slot E/2, and only lane 0 (k = 0) uses it
        const float2 x_half = make_float2(z[E / 2].x, -z[E / 2].y);
        if constexpr (MODE == 0) {
            float2 *__restrict__ row = reinterpret_cast<float2 *>(spectrum + (size_t)ind * (N + 8));
#pragma unroll
            for (int q = 0; q < H; ++q) {
                const int k = lane + kWave * q;
                v2f_t xa2, xbc2;
                split_pair_2x(z[q], split_partner<M>(z, q, lane, partner_addr), w_split[q], xa2, xbc2);
                row[k] = make_float2(0.5f * xa2.x, 0.5f * xa2.y);
                row[M - k] = make_float2(0.5f * xbc2.x, -0.5f * xbc2.y); // k = 0 lands on the Nyquist bin M
            }
            if (lane == 0) row[M / 2] = x_half;
        } else {
            v2f_t acc2 = v2f_t{0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < H; ++q) {
                const int k = lane + kWave * q;
                v2f_t xa2, xbc2;
                split_pair_2x(z[q], split_partner<M>(z, q, lane, partner_addr), w_split[q], xa2, xbc2); // 2 X[k], 2 conj X[M-k]
                if (k == 0) xbc2 = v2f_t{2.0f * x_half.x, 2.0f * x_half.y}; // the fitness skips the Nyquist bin and needs bin M/2
                bin_error2(acc2, make_float2(xa2.x, xa2.y), make_float2(xbc2.x, xbc2.y), tgt_s[k], tgt_s[k == 0 ? M / 2 : M - k], half_scale);
            }
            float acc = wave_sum(acc2.x + acc2.y);
            if (lane == 0) fitness[ind] = acc;
        }
        wave_lds_sync<W == 1>(); // orders this row's LDS reads before the next row's writes
#ifdef SOTS_STAMP
        {
            const unsigned long long t3 = __builtin_amdgcn_s_memtime();
            if (st_rows == 0 && lane == 0 && st_slot < 512) g_stamps[3 * 8192 + st_slot * 16 + 7] = t0 - st_begin_clk; // prologue
            st_wait += t1 - t0, st_fft += t2 - t1, st_tail += t3 - t2, st_rows += 1;
            if (lane == 0 && st_slot < 512) {
                g_stamps[3 * 8192 + st_slot * 16 + 0] = st_wait;
                g_stamps[3 * 8192 + st_slot * 16 + 1] = st_fft;
                g_stamps[3 * 8192 + st_slot * 16 + 2] = st_tail;
                g_stamps[3 * 8192 + st_slot * 16 + 3] = st_rows;
                g_stamps[3 * 8192 + st_slot * 16 + 4] = st_begin;
                g_stamps[3 * 8192 + st_slot * 16 + 5] = __builtin_amdgcn_s_memrealtime();
                g_stamps[3 * 8192 + st_slot * 16 + 6] = __builtin_amdgcn_s_memtime() - st_begin_clk;
            }
        }
#endif
        return filled;
    };
    float4 b0[Q], b1[Q], b2[Q];
    uint32_t r0 = ind, r1 = blockIdx.x + (W + wave) * grid, r2; // the rows in the three buffers
    request(b0, r0);
    asm volatile("" ::: "memory"); // keep the two requests in this order (the loop's waits count on it)
    request(b1, r1);
    while (true) { // (takes only grow: a wavefront whose next row is past the end has no later one either)
        ind = r0, r2 = process(b0, b2);
        if (r1 >= p_len) break;
        ind = r1, r0 = process(b1, b0);
        if (r2 >= p_len) break;
        ind = r2, r1 = process(b2, b1);
        if (r0 >= p_len) break;
    }
#ifdef SOTS_STAMP_ENDS
    if (lane == 0 && blockIdx.x * W + wave < 4096) {
        unsigned long long *o = g_stamps + (blockIdx.x * W + wave) * 4;
        o[0] = se_begin, o[1] = __builtin_amdgcn_s_memrealtime(), o[2] = se_rows, o[3] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | ((16 - 1) << 11)); // HW_ID: wave, SIMD, CU, SH, SE
    }
#endif
}

// fitnessPopulation on materialised spectrum rows; same bin -> lane assignment and
// summation order as k_fft<.., 1>, so both paths give the same fp32 sum.
template <int LOG2N>
__global__ __launch_bounds__(kWave) void k_fitness(const float *__restrict__ spectrum,
                                                   const float *__restrict__ target,
                                                   float *__restrict__ fitness, uint32_t p_len, float inv_n,
                                                   float inv_wf)
{
    constexpr int N = 1 << LOG2N, M = N / 2, E = M / kWave;
    const int lane = threadIdx.x;
    for (uint32_t ind = blockIdx.x; ind < p_len; ind += gridDim.x) {
        const float2 *__restrict__ row = reinterpret_cast<const float2 *>(spectrum + (size_t)ind * (N + 8));
        v2f_t acc2 = v2f_t{0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < E / 2; ++q) {
            const int k = lane + kWave * q;
            const int kb = k == 0 ? M / 2 : M - k;
            bin_error2(acc2, row[k], row[kb], target[k], target[kb], inv_n * inv_wf);
        }
        float acc = wave_sum(acc2.x + acc2.y);
        if (lane == 0) fitness[ind] = acc;
    }
}

#pragma clang fp contract(off)

// ------------------------------------------------------------------------------------
// sortPopulation, ocl_program.cl:664-711 (rank sort) / Population::bubbleSortPopulation,
// Evolutionary_Strategy.hpp:108-124.  Ascending by fitness, equal fitness keeps the lower
// original index first (the CPU's stable order), NaN after every number.  Implemented as a
// bitonic network over unique 64-bit keys (order-preserving fitness bits << 32 | index).
// ------------------------------------------------------------------------------------
// ---- island exchange inside the kernels that move the sorted rows (SortExchange, sots_kernels.h) -------------
// a destination row that belongs to an immigrant: the local row that sorted there is NOT written
__device__ __forceinline__ bool ex_immigrant_row(const SortExchange &ex, uint32_t dst)
{
    return ex.imm != nullptr && dst - ex.imm_first < ex.imm_rows;
}
// element `c` (0..d-1 values, d..2d-1 steps, 2d fitness) of destination row dst also goes to the packed elite rows
__device__ __forceinline__ void ex_sink(const SortExchange &ex, uint32_t dst, uint32_t c, uint32_t d, float v)
{
    if (dst < ex.sink_rows) ex.sink[(size_t)dst * (2 * d + 1) + (c == 2 * d ? 0u : c + 1u)] = v;
}
// the immigrant rows themselves (k_unpack_rows' copy), by the threads of ONE workgroup
__device__ __forceinline__ void ex_unpack(const SortExchange &ex, float *__restrict__ vout, float *__restrict__ sout,
                                          float *__restrict__ fout, uint32_t d, uint32_t tid, uint32_t threads)
{
    if (ex.imm == nullptr) return;
    const uint32_t w = 2 * d + 1, total = ex.imm_rows * w;
    for (uint32_t e = tid; e < total; e += threads) {
        const uint32_t r = e / w, c = e - r * w, dst = ex.imm_first + r;
        const uint32_t sr = r < ex.skip_first ? r : r + ex.skip_count;
        const float v = ex.imm[(size_t)sr * w + c]; // packed rows: [fitness, values.., steps..]
        if (c == 0) fout[dst] = v;
        else if (c <= d) vout[(size_t)dst * d + (c - 1)] = v;
        else sout[(size_t)dst * d + (c - 1 - d)] = v;
        if (dst < ex.sink_rows) ex.sink[(size_t)dst * w + c] = v;
    }
}

constexpr int kSortThreads = 1024;
constexpr uint32_t kSortTile = 4096; // keys per LDS tile (32 KiB)
constexpr uint32_t kSortSmall = 1024; // populations sorted by one workgroup in one launch (k_sort_small)

__device__ __forceinline__ uint64_t make_key(float f, uint32_t idx)
{
    uint32_t u;
    if (f != f) {
        u = 0xFFFFFFFFu;
    } else {
        u = __float_as_uint(f == 0.0f ? 0.0f : f);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    return ((uint64_t)u << 32) | idx;
}

// Padding slot g >= P of the power-of-two key array: after every real key (NaN included) and
// unique, which the counting merges rely on.
__device__ __forceinline__ uint64_t pad_key(uint32_t g) { return 0xFFFFFFFF00000000ull | g; }

__device__ __forceinline__ void cmp_swap(uint64_t &a, uint64_t &b, bool ascending)
{
    if ((a > b) == ascending) {
        const uint64_t t = a;
        a = b;
        b = t;
    }
}

// Builds the keys of one tile and sorts it completely (all steps with k <= tile).
__global__ __launch_bounds__(kSortThreads) void k_sort_tiles(const float *__restrict__ fitness,
                                                             uint64_t *__restrict__ keys, uint32_t p_len,
                                                             uint32_t tile, uint32_t alternate)
{
    __shared__ uint64_t s[kSortTile];
    const uint32_t base = blockIdx.x * tile;
    // alternate != 0: odd tiles end up descending, ready for further global bitonic merges;
    // alternate == 0: every tile ascending (rank merge)
    const uint32_t dir_base = alternate ? base : 0u;
    for (uint32_t i = threadIdx.x; i < tile; i += blockDim.x) {
        const uint32_t g = base + i;
        s[i] = g < p_len ? make_key(fitness[g], g) : pad_key(g);
    }
    __syncthreads();
    for (uint32_t k = 2; k <= tile; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < tile / 2; t += blockDim.x) {
                const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t hi = lo | j;
                const bool asc = (((dir_base + lo) & k) == 0);
                uint64_t a = s[lo], b = s[hi];
                cmp_swap(a, b, asc);
                s[lo] = a;
                s[hi] = b;
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < tile; i += blockDim.x) keys[base + i] = s[i];
}

// 1024-key tile for the rank merge, without workgroup barriers in the sorting network: each
// of the four wavefronts bitonic-sorts its own run of 256 keys in its private quarter of the
// LDS buffer (the LDS operations of one wavefront execute in order, so the 36 steps need no
// s_barrier), then every key's place in the tile is its index in its own run plus its lower
// bound in the other three runs (24 LDS reads), the same counting argument as the tile merge.
constexpr uint32_t kRunKeys = 256;

__global__ __launch_bounds__(256) void k_sort_tiles_1k(const float *__restrict__ fitness,
                                                       uint64_t *__restrict__ keys, uint32_t p_len)
{
    __shared__ uint64_t runs[4 * kRunKeys];
    __shared__ uint64_t merged[4 * kRunKeys];
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t base = blockIdx.x * 4 * kRunKeys + wave * kRunKeys;
    uint64_t *__restrict__ run = runs + wave * kRunKeys;
#pragma unroll
    for (uint32_t r = 0; r < kRunKeys / kWave; ++r) {
        const uint32_t i = lane + kWave * r, g = base + i;
        run[i] = g < p_len ? make_key(fitness[g], g) : pad_key(g);
    }
    for (uint32_t k = 2; k <= kRunKeys; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (uint32_t r = 0; r < kRunKeys / 2 / kWave; ++r) {
                const uint32_t t = lane + kWave * r;
                const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t hi = lo | j;
                uint64_t a = run[lo], b = run[hi];
                cmp_swap(a, b, (lo & k) == 0);
                run[lo] = a;
                run[hi] = b;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < kRunKeys / kWave; ++r) {
        const uint32_t i = lane + kWave * r;
        const uint64_t x = run[i];
        uint32_t rank = i;
#pragma unroll
        for (uint32_t o = 1; o < 4; ++o) {
            const uint64_t *__restrict__ other = runs + ((wave + o) & 3u) * kRunKeys;
            uint32_t pos = 0;
#pragma unroll
            for (uint32_t step = kRunKeys >> 1; step >= 1; step >>= 1) pos += (other[pos + step - 1] < x) ? step : 0u;
            rank += pos + ((other[pos] < x) ? 1u : 0u);
        }
        merged[rank] = x;
    }
    __syncthreads();
    uint64_t *__restrict__ out = keys + (size_t)blockIdx.x * 4 * kRunKeys;
    for (uint32_t i = threadIdx.x; i < 4 * kRunKeys; i += 256) out[i] = merged[i];
}

// One compare-exchange step (k, j) with j >= tile, over the whole padded array.
__global__ __launch_bounds__(256) void k_sort_global_step(uint64_t *__restrict__ keys, uint32_t n_pad,
                                                          uint32_t k, uint32_t j)
{
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_pad / 2; t += gridDim.x * blockDim.x) {
        const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const uint32_t hi = lo | j;
        const bool asc = ((lo & k) == 0);
        uint64_t a = keys[lo], b = keys[hi];
        cmp_swap(a, b, asc);
        keys[lo] = a;
        keys[hi] = b;
    }
}

// The remaining steps j = tile/2 .. 1 of merge size k, tile-local.
__global__ __launch_bounds__(kSortThreads) void k_sort_tile_merge(uint64_t *__restrict__ keys, uint32_t k,
                                                                  uint32_t tile)
{
    __shared__ uint64_t s[kSortTile];
    const uint32_t base = blockIdx.x * tile;
    for (uint32_t i = threadIdx.x; i < tile; i += blockDim.x) s[i] = keys[base + i];
    __syncthreads();
    const bool asc = ((base & k) == 0); // k > tile: the direction is uniform over the tile
    for (uint32_t j = tile >> 1; j > 0; j >>= 1) {
        for (uint32_t t = threadIdx.x; t < tile / 2; t += blockDim.x) {
            const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
            const uint32_t hi = lo | j;
            uint64_t a = s[lo], b = s[hi];
            cmp_swap(a, b, asc);
            s[lo] = a;
            s[hi] = b;
        }
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < tile; i += blockDim.x) keys[base + i] = s[i];
}

// ---- rank merge of sorted tiles ------------------------------------------------------
// With T sorted tiles of unique keys, the final position of a key is the number of keys
// below it = its index in its own tile + sum over the other tiles of lower_bound(tile, key).
// Workgroup (a, b) stages tile b in LDS and binary-searches every key of tile a in it;
// the T partial counts per key are then summed by the scatter kernel, which moves the
// (2D+1)-float row straight to its final place.  Two launches instead of the ~15 of the
// global bitonic steps, and every CU is busy.
constexpr int kRankThreads = 256;
constexpr uint32_t kRankGroup = 8;      // tiles staged in LDS per workgroup (tile = 1024: 64 KiB)
constexpr uint32_t kRankLdsKeys = 8192; // LDS capacity in keys

// Workgroup (a, g): keys of tile a against tiles [g*GROUP, (g+1)*GROUP); writes one partial
// count per key of a.  Keys are unique, so against its own tile a key's lower bound is its
// sorted index and no tile needs a special case.  The tiles are staged by LDS-DMA (all pieces
// in flight at once) and the GROUP x 4 binary searches of a thread advance together, one level
// per trip, so a trip has GROUP x 4 independent LDS reads in flight instead of 4.
// MERGE (large populations, first of two levels): the grid has one workgroup per tile, which ranks its
// keys inside its OWN group of GROUP tiles only and writes them to their place in the group's
// sorted run of GROUP * tile keys (merged[]); the second level then ranks runs instead of tiles.
template <uint32_t GROUP, bool MERGE = false>
__global__ __launch_bounds__(kRankThreads) void k_sort_rank_pairs(const uint64_t *__restrict__ keys,
                                                                  uint16_t *__restrict__ partial,
                                                                  uint32_t n_pad, uint32_t tile,
                                                                  uint64_t *__restrict__ merged = nullptr)
{
    __shared__ uint64_t s[kRankLdsKeys];
    const uint32_t a = blockIdx.x, g = MERGE ? blockIdx.x / GROUP : blockIdx.y;
    const uint64_t *__restrict__ kb = keys + (size_t)g * GROUP * tile;
    {
        typedef __attribute__((address_space(3))) void *lds_ptr_t;
        const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x & (kWave - 1);
        constexpr uint32_t kChunk = kWave * 2; // keys per wavefront instruction (16 B per lane)
        for (uint32_t ch = wave; ch < GROUP * tile / kChunk; ch += kRankThreads / kWave)
            __builtin_amdgcn_global_load_lds(kb + ch * kChunk + lane * 2u, (lds_ptr_t)(s + ch * kChunk), 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0); // the copies have landed
    }
    __syncthreads();
    const uint64_t *__restrict__ ka = keys + (size_t)a * tile;
    uint16_t *__restrict__ out = partial + (size_t)g * n_pad + (size_t)a * tile;
    for (uint32_t i0 = threadIdx.x; i0 < tile; i0 += 4 * kRankThreads) {
        uint64_t x[4];
        uint32_t pos[GROUP][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t i = i0 + q * kRankThreads;
            x[q] = i < tile ? ka[i] : 0ull;
#pragma unroll
            for (uint32_t bb = 0; bb < GROUP; ++bb) pos[bb][q] = 0;
        }
        for (uint32_t step = tile >> 1; step >= 1; step >>= 1) {
#pragma unroll
            for (uint32_t bb = 0; bb < GROUP; ++bb)
#pragma unroll
                for (int q = 0; q < 4; ++q) pos[bb][q] += (s[bb * tile + pos[bb][q] + step - 1] < x[q]) ? step : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t total = 0;
#pragma unroll
            for (uint32_t bb = 0; bb < GROUP; ++bb) total += pos[bb][q] + ((s[bb * tile + pos[bb][q]] < x[q]) ? 1u : 0u);
            const uint32_t i = i0 + q * kRankThreads;
            if constexpr (MERGE) {
                if (i < tile) merged[(size_t)g * GROUP * tile + total] = x[q];
            } else {
                if (i < tile) out[i] = (uint16_t)total;
            }
        }
    }
}

// 64 consecutive slots of the tile-sorted key array per workgroup: four quarter-sums of the
// partial counts per slot (all 256 lanes load), then the rows move to their final places.
__global__ __launch_bounds__(kRankThreads) void k_sort_rank_scatter(const uint64_t *__restrict__ keys,
                                                                    const uint16_t *__restrict__ partial,
                                                                    const float *__restrict__ vin,
                                                                    const float *__restrict__ sin,
                                                                    const float *__restrict__ fin,
                                                                    float *__restrict__ vout, float *__restrict__ sout,
                                                                    float *__restrict__ fout, uint32_t n_pad,
                                                                    uint32_t groups, uint32_t p_len, uint32_t d,
                                                                    uint32_t first_row, SortExchange ex)
{
    constexpr uint32_t kSlots = 64;
    if (blockIdx.x == 0) ex_unpack(ex, vout, sout, fout, d, threadIdx.x, kRankThreads);
    __shared__ uint32_t part[4][kSlots];
    __shared__ uint32_t src_row[kSlots];
    const uint32_t j = threadIdx.x & (kSlots - 1), quarter = threadIdx.x / kSlots;
    const uint32_t e = blockIdx.x * kSlots + j;
    uint32_t sum = 0;
    for (uint32_t g = quarter; g < groups; g += 4) sum += partial[(size_t)g * n_pad + e];
    part[quarter][j] = sum;
    if (quarter == 0) src_row[j] = (uint32_t)keys[e]; // >= P for padding keys
    __syncthreads();
    const uint32_t w = 2 * d + 1;
    for (uint32_t t = threadIdx.x; t < kSlots * w; t += kRankThreads) {
        const uint32_t jj = t / w, c = t - jj * w;
        const uint32_t src = src_row[jj];
        if (src >= p_len) continue;
        const uint32_t dst = part[0][jj] + part[1][jj] + part[2][jj] + part[3][jj];
        if (dst < first_row) continue; // rows the selection kernels have already placed
        if (ex_immigrant_row(ex, dst)) continue;
        const float v = c < d ? vin[(size_t)src * d + c] : c < 2 * d ? sin[(size_t)src * d + (c - d)] : fin[src];
        if (c < d) vout[(size_t)dst * d + c] = v;
        else if (c < 2 * d) sout[(size_t)dst * d + (c - d)] = v;
        else fout[dst] = v;
        ex_sink(ex, dst, c, d, v);
    }
}

// Row r of the new half <- row (keys[r] & 0xffffffff) of the old half.
__global__ __launch_bounds__(256) void k_sort_gather(const uint64_t *__restrict__ keys,
                                                     const float *__restrict__ vin,
                                                     const float *__restrict__ sin,
                                                     const float *__restrict__ fin, float *__restrict__ vout,
                                                     float *__restrict__ sout, float *__restrict__ fout,
                                                     uint32_t p_len, uint32_t d, uint32_t first_row, SortExchange ex)
{
    const uint32_t w = 2 * d + 1;
    const uint32_t total = p_len * w;
    if (blockIdx.x == 0) ex_unpack(ex, vout, sout, fout, d, threadIdx.x, blockDim.x);
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x + first_row * w; e < total; e += gridDim.x * blockDim.x) {
        const uint32_t r = e / w, c = e - r * w;
        if (ex_immigrant_row(ex, r)) continue;
        const uint32_t src = (uint32_t)keys[r];
        const float v = c < d ? vin[(size_t)src * d + c] : c < 2 * d ? sin[(size_t)src * d + (c - d)] : fin[src];
        if (c < d) vout[(size_t)r * d + c] = v;
        else if (c < 2 * d) sout[(size_t)r * d + (c - d)] = v;
        else fout[r] = v;
        ex_sink(ex, r, c, d, v);
    }
}

// ------------------------------------------------------------------------------------
// Selection for the fused generation loop: the best `need` rows IN ORDER and nothing else.
// The next recombination reads whole parent blocks only (recombine_source above,
// ocl_program.cl:99-112), so inside sots_execute_generations the order of the other rows is
// never looked at; the full order is produced (by launch_sort, from the untouched unsorted half)
// when somebody reads the population.  Rows 0..need-1 are bit-identical to the full sort's.
//
// Two launches.  k_sel_tiles sorts tiles of 1024 (fitness bits, index) keys: one key per lane,
// 16 wavefronts sort 64 keys each in registers (bitonic network over lane exchanges, no LDS
// traffic, no barrier), then every key's place in the tile is its lane plus its lower bound in
// the other 15 runs.  It writes the sorted fitness bits and indices as two u32 arrays and, per
// tile, the four keys at positions 255, 511, 767, 1023 ("samples").
// k_sel_rank_scatter (one workgroup per CU) finds v* = the ceil(need/256)-th smallest sample:
// at least need keys are <= v*, and every key <= v* lies in its tile's first
// 256 * (samples_t <= v*) + 256 positions.  Only those prefixes (need + 256 * tiles keys when
// there are no ties: 128 KiB at P = 65536) are staged in LDS as 4-byte fitness bits; a staged key's
// rank is the sum of its lower bounds in every staged prefix, and keys whose rank is below `need`
// move their rows.  Tiles are contiguous index ranges, so "equal fitness, lower index first" needs
// no index in LDS: against an EARLIER tile an equal key counts (<=), against a LATER one it does
// not (<).  Keys above v* get ranks >= need (every key <= v* is staged and there are >= need of
// them), so no check can fail and there is no fallback path; prefixes that exceed the LDS are staged
// in several passes.
// ------------------------------------------------------------------------------------
constexpr uint32_t kSelTile = 1024, kSelSamples = 4, kSelQuantum = kSelTile / kSelSamples;
constexpr uint32_t kSelThreads = 1024, kSelLanesPerKey = 8, kSelKeysPerRound = kSelThreads / kSelLanesPerKey;
constexpr uint32_t kSelCap = 35328;                 // staged keys per pass: 138 KiB of LDS
constexpr uint32_t kSelSkew = 4;                    // consecutive tiles start 4 more words (16 B) off the 1 KiB grid, see below
// the rank kernel handles 16..64 tiles of 1024 keys (P <= 65536) and, for P = 131072, 32 tiles of 4096 keys merged four by
// four (k_sel_merge4).  Same-box A/B of the un-instrumented loop (profiles/r03_experiments.md): at P = 131072 the selection
// beats the two-level full sort (237 against 244 us per generation), at 262144 it loses (496 against 482), so it stops at 128 tiles
constexpr uint32_t kSelMinTiles = 16, kSelMaxTiles = 128, kSelMaxOwn = 512; // kSelMaxTiles: tiles the rank kernel handles (of either size)

// order-preserving bits of a fitness: every number (at most 0xFF800000, +inf) below NaN (0xFFFFFFFD), NaN
// below the padding key (0xFFFFFFFE); 0xFFFFFFFF stays free so that "bits + 1" never wraps
constexpr uint32_t kSelPadBits = 0xFFFFFFFEu;
__device__ __forceinline__ uint32_t order_bits(float f)
{
    if (f != f) return 0xFFFFFFFDu;
    const uint32_t u = __float_as_uint(f == 0.0f ? 0.0f : f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// value of lane (lane ^ J) for J = 1, 2 (DPP quad permutes), 4, 8, 16 (ds_swizzle through the LDS crossbar,
// no LDS memory), 32 (v_permlane32_swap)
template <uint32_t J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v)
{
    if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true); // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true); // quad_perm [2,3,0,1]
    else if constexpr (J == 32) {
        const auto sw = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (threadIdx.x & 32u) ? sw[0] : sw[1]; // lanes 32-63 get lane-32's value in [0], lanes 0-31 lane+32's in [1]
    } else return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (int)((J << 10) | 0x1Fu)); // bit mode: xor J, and 0x1f
}

template <uint32_t K, uint32_t J>
__device__ __forceinline__ void bitonic_step(uint32_t &b, uint32_t &i, uint32_t lane)
{
    const uint32_t pb = lane_xor<J>(b), pi = lane_xor<J>(i);
    const bool keep_min = ((lane & J) == 0) == ((lane & K) == 0);
    const bool mine_less = b < pb || (b == pb && i < pi);
    if (keep_min != mine_less) {
        b = pb;
        i = pi;
    }
}
template <uint32_t K, uint32_t J>
__device__ __forceinline__ void bitonic_merge(uint32_t &b, uint32_t &i, uint32_t lane)
{
    bitonic_step<K, J>(b, i, lane);
    if constexpr (J > 1) bitonic_merge<K, J / 2>(b, i, lane);
}

// SPLIT workgroups per tile (blockIdx = tile + part * tiles): a tile is 16 wavefronts on ONE CU, and with the 64 tiles of
// P = 65536 three quarters of the GPU idle while each of those CUs answers 1440 wavefront-wide LDS reads of the searches.
// Every workgroup of a tile sorts all 16 runs (the searches need them; the register network is cheap), then only the
// wavefronts of ITS runs - a quarter, one per SIMD - search and write.
#ifndef SOTS_SEL_SPLIT
#define SOTS_SEL_SPLIT 4
#endif
template <uint32_t SPLIT>
__global__ __launch_bounds__(kSelTile) void k_sel_tiles(const float *__restrict__ fitness, uint32_t *__restrict__ kbits,
                                                         uint32_t *__restrict__ kidx, uint32_t *__restrict__ samples,
                                                         uint32_t p_len, uint32_t tiles)
{
    constexpr uint32_t kRuns = kSelTile / kWave;
    __shared__ uint32_t runs[kSelTile];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const uint32_t tile = SPLIT == 1 ? blockIdx.x : blockIdx.x % tiles, part = SPLIT == 1 ? 0u : blockIdx.x / tiles;
    const uint32_t g = tile * kSelTile + tid;
    SOTS_PHASE_BEGIN();
    uint32_t b = g < p_len ? order_bits(fitness[g]) : kSelPadBits, i = g; // padding keys sort last, by index
#ifdef SOTS_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the stamp below then includes the fitness load
#endif
    SOTS_PHASE(8);
    bitonic_merge<2, 1>(b, i, lane);
    bitonic_merge<4, 2>(b, i, lane);
    bitonic_merge<8, 4>(b, i, lane);
    bitonic_merge<16, 8>(b, i, lane);
    bitonic_merge<32, 16>(b, i, lane);
    bitonic_merge<64, 32>(b, i, lane);
    runs[tid] = b;
    __syncthreads();
    SOTS_PHASE(9);
    if (SPLIT > 1 && wave / (kRuns / SPLIT) != part) return;
    // place in the tile = lane + lower bounds in the other 15 runs, all 15 searches advanced level by level so
    // that the LDS reads of a level are in flight together.  A run of an earlier wavefront holds lower indices,
    // so its equal keys come first (count <=, i.e. < b + 1), a later run's do not (<).
    uint32_t pos[kRuns], thr[kRuns]; // absolute positions in runs[]; "entries below thr" is what is counted
#pragma unroll
    for (uint32_t w = 0; w < kRuns; ++w) {
        pos[w] = w * kWave;
        thr[w] = w == wave ? 0u : b + (w < wave ? 1u : 0u); // own run: nothing counts (the lane is added below)
    }
#pragma unroll
    for (uint32_t step = kWave / 2; step >= 1; step >>= 1) {
#pragma unroll
        for (uint32_t w = 0; w < kRuns; ++w) pos[w] += runs[pos[w] + step - 1] < thr[w] ? step : 0u;
    }
    uint32_t rank = lane;
#pragma unroll
    for (uint32_t w = 0; w < kRuns; ++w) rank += pos[w] - w * kWave + (runs[pos[w]] < thr[w] ? 1u : 0u);
    SOTS_PHASE(10);
    const size_t o = (size_t)tile * kSelTile + rank;
    kbits[o] = b;
    kidx[o] = i;
    if ((rank & (kSelQuantum - 1)) == kSelQuantum - 1) {
        samples[tile * kSelSamples + rank / kSelQuantum] = b;
        samples[tiles * kSelSamples + tile * kSelSamples + rank / kSelQuantum] = i; // indices behind all the bits
    }
    SOTS_PHASE(11);
}

// A whole population of at most 1024 rows in ONE launch and one workgroup (the reference's default sizes are this
// small, and there a launch costs as much as the sort): k_sel_tiles' network - one key per lane, 64-key runs sorted in
// registers, a key's place = its lane + its lower bounds in the other runs - and then the workgroup moves the rows,
// a lane per output element.  Same order as the full sort: fitness, equal fitness by index, NaN last.
template <uint32_t RUNS>
__global__ __launch_bounds__(RUNS *kWave) void k_sort_small(const float *__restrict__ vin, const float *__restrict__ sin,
                                                            const float *__restrict__ fin, float *__restrict__ vout,
                                                            float *__restrict__ sout, float *__restrict__ fout,
                                                            uint32_t p_len, uint32_t d, uint32_t first_row, SortExchange ex)
{
    __shared__ uint32_t runs[RUNS * kWave], from[RUNS * kWave];
    ex_unpack(ex, vout, sout, fout, d, threadIdx.x, RUNS * kWave);
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    uint32_t b = tid < p_len ? order_bits(fin[tid]) : kSelPadBits, i = tid;
    bitonic_merge<2, 1>(b, i, lane);
    bitonic_merge<4, 2>(b, i, lane);
    bitonic_merge<8, 4>(b, i, lane);
    bitonic_merge<16, 8>(b, i, lane);
    bitonic_merge<32, 16>(b, i, lane);
    bitonic_merge<64, 32>(b, i, lane);
    uint32_t rank = lane;
    if constexpr (RUNS > 1) {
        runs[tid] = b;
        __syncthreads();
        uint32_t pos[RUNS], thr[RUNS];
#pragma unroll
        for (uint32_t w = 0; w < RUNS; ++w) {
            pos[w] = w * kWave;
            thr[w] = w == wave ? 0u : b + (w < wave ? 1u : 0u);
        }
#pragma unroll
        for (uint32_t step = kWave / 2; step >= 1; step >>= 1) {
#pragma unroll
            for (uint32_t w = 0; w < RUNS; ++w) pos[w] += runs[pos[w] + step - 1] < thr[w] ? step : 0u;
        }
#pragma unroll
        for (uint32_t w = 0; w < RUNS; ++w) rank += pos[w] - w * kWave + (runs[pos[w]] < thr[w] ? 1u : 0u);
    }
    // from[place] = the row that sorted there (padding keys sort behind every row).  The rows then move with lane =
    // OUTPUT element: a wavefront's stores are consecutive addresses and its loads runs of d consecutive floats - one
    // workgroup is one CU's address path, and a lane-per-row move (64 cache lines per instruction, 2 d + 1
    // instructions each way) took twice as long as the sort itself at d = 12.
    from[rank] = i;
    __syncthreads();
    constexpr uint32_t T = RUNS * kWave;
    if (tid < p_len && tid >= first_row && !ex_immigrant_row(ex, tid)) {
        const float f = fin[from[tid]];
        fout[tid] = f;
        ex_sink(ex, tid, 2 * d, d, f);
    }
    const uint32_t qd = T / d, rd = T - qd * d; // uniform: one scalar division per launch
    uint32_t r = tid / d, c = tid - r * d;
    while (r < p_len) { // four elements per trip, their loads in flight together
        uint32_t dst[4], row[4];
        float v[4], s[4];
        bool ok[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            row[q] = r, dst[q] = r * d + c;
            ok[q] = r < p_len && r >= first_row && !ex_immigrant_row(ex, r);
            if (ok[q]) {
                const uint32_t src = from[r] * d + c;
                v[q] = vin[src], s[q] = sin[src];
            }
            r += qd, c += rd;
            if (c >= d) c -= d, ++r;
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q)
            if (ok[q]) {
                const uint32_t cq = dst[q] - row[q] * d;
                vout[dst[q]] = v[q], sout[dst[q]] = s[q];
                ex_sink(ex, row[q], cq, d, v[q]), ex_sink(ex, row[q], d + cq, d, s[q]);
            }
    }
}

// Four neighbouring sorted tiles of 1024 keys -> one sorted tile of 4096 with a sample every 512 keys.  For
// populations of 128 k and more the rank kernel's work (staged keys x tiles x search depth) is quartered by
// having a quarter of the tiles.  One workgroup per group of four: the 4096 fitness-bit words sit in LDS, a key's
// place = its position in its own tile + its lower bounds in the three siblings (earlier tile: <=, later: <).
constexpr uint32_t kSelBigTile = 4 * kSelTile, kSelBigQuantum = 512, kSelBigSamples = kSelBigTile / kSelBigQuantum;

__global__ __launch_bounds__(kSelTile) void k_sel_merge4(const uint32_t *__restrict__ kbits, const uint32_t *__restrict__ kidx,
                                                          uint32_t *__restrict__ obits, uint32_t *__restrict__ oidx,
                                                          uint32_t *__restrict__ samples)
{
    __shared__ uint32_t s[kSelBigTile];
    const uint32_t tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * kSelBigTile;
    uint32_t b[4], i[4];
#pragma unroll
    for (uint32_t t = 0; t < 4; ++t) { // thread `tid` holds position `tid` of each of the four tiles
        b[t] = kbits[base + t * kSelTile + tid];
        i[t] = kidx[base + t * kSelTile + tid];
        s[t * kSelTile + tid] = b[t];
    }
    __syncthreads();
    uint32_t pos[4][4], thr[4][4]; // [own tile][sibling]: absolute search positions in s[]
#pragma unroll
    for (uint32_t t = 0; t < 4; ++t)
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            pos[t][u] = u * kSelTile;
            thr[t][u] = u == t ? 0u : b[t] + (u < t ? 1u : 0u); // bits <= 0xFFFFFFFE: no wrap
        }
#pragma unroll
    for (uint32_t step = kSelTile / 2; step >= 1; step >>= 1) {
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t)
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u)
                if (u != t) pos[t][u] += s[pos[t][u] + step - 1] < thr[t][u] ? step : 0u;
    }
    const uint32_t ns = gridDim.x * kSelBigSamples;
#pragma unroll
    for (uint32_t t = 0; t < 4; ++t) {
        uint32_t rank = tid;
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u)
            if (u != t) rank += pos[t][u] - u * kSelTile + (s[pos[t][u]] < thr[t][u] ? 1u : 0u);
        obits[base + rank] = b[t];
        oidx[base + rank] = i[t];
        if ((rank & (kSelBigQuantum - 1)) == kSelBigQuantum - 1) {
            samples[blockIdx.x * kSelBigSamples + rank / kSelBigQuantum] = b[t];
            samples[ns + blockIdx.x * kSelBigSamples + rank / kSelBigQuantum] = i[t];
        }
    }
}

constexpr uint32_t kSelChains = 8; // tiles searched together by one lane (independent LDS reads in flight)

// TILE keys per sorted tile, a sample every QUANT keys (1024 / 256 straight from k_sel_tiles; 4096 / 512 after
// k_sel_merge4 for populations whose 1024-key tiles would be too many); R samples per lane = tiles * (TILE / QUANT) / 64
// LPK lanes share a key's searches, each kSelChains tiles at a time: 8 for the 64 tiles of the 1024-key layout, 4 for
// the <= 20 big tiles a pass stages (with 8, three quarters of the search chains would run empty)
template <uint32_t R, uint32_t TILE, uint32_t QUANT, uint32_t LPK>
__global__ __launch_bounds__(kSelThreads) void k_sel_rank_scatter(const uint32_t *__restrict__ kbits,
                                                                  const uint32_t *__restrict__ kidx,
                                                                  const uint32_t *__restrict__ samples,
                                                                  const float *__restrict__ vin,
                                                                  const float *__restrict__ sin,
                                                                  const float *__restrict__ fin,
                                                                  float *__restrict__ vout, float *__restrict__ sout,
                                                                  float *__restrict__ fout, uint32_t tiles, uint32_t need,
                                                                  uint32_t p_len, uint32_t d, SortExchange ex, uint32_t whole_tiles)
{
    constexpr uint32_t kWaves = kSelThreads / kWave;
    if (blockIdx.x == 0) ex_unpack(ex, vout, sout, fout, d, threadIdx.x, kSelThreads);
    // the names of the 1024 / 256 layout, shadowed by this instantiation's values
    constexpr uint32_t kSelTile = TILE, kSelQuantum = QUANT, kSelSamples = TILE / QUANT;
    // a tile whose prefix STARTS inside the window fits whole, skew included (at most window / QUANT tiles per pass)
    constexpr uint32_t kSelWindow = TILE == 1024 ? 33536 : 30720;
    static_assert(kSelWindow + kSelTile + kSelSkew * (kSelWindow / kSelQuantum) <= kSelCap, "staging buffer too small");
    static_assert(R * kWave <= kSelMaxTiles * sots::kSelSamples, "sample store too small");
    __shared__ __attribute__((aligned(16))) uint32_t staged[kSelCap];
    __shared__ __attribute__((aligned(16))) uint32_t smp[kSelMaxTiles * sots::kSelSamples], smi[kSelMaxTiles * sots::kSelSamples];
    __shared__ uint32_t off[kSelMaxTiles + 1]; // first staged position of each tile, all passes concatenated
    __shared__ uint16_t chunk_tile[kSelMaxTiles * sots::kSelSamples]; // tile of every QUANT staged positions
    // LDS place of tile t's prefix in its pass: off[t] - w0 + kSelSkew * (t - ta).  Prefix lengths are multiples
    // of 256 words, so without the skew every prefix would start on bank 0 and the first binary-search levels
    // (positions 511, 255, 127, 63, 31 of eight different tiles in one instruction) would all hit bank 31.
    __shared__ uint32_t own_bits[kSelMaxOwn], own_tile[kSelMaxOwn], own_pos[kSelMaxOwn], own_rank[kSelMaxOwn], own_src[kSelMaxOwn];
    __shared__ uint32_t wave_tot[kWaves];
    __shared__ unsigned long long vstar_s;
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const uint32_t ns = tiles * kSelSamples;
    SOTS_PHASE_BEGIN();

    // ---- v* = the k-th smallest sample, k = ceil(need / 256), in (fitness bits, index) order: with the
    // index in the comparison, equal fitness values (a converged population is full of them) cannot
    // inflate the staged prefixes.  Every lane holds R = ns / 64 samples in registers; wavefront w ranks the
    // samples held by lanes 4w .. 4w+3: one sample at a time is broadcast as a SCALAR and compared with all
    // ns samples by R 64-bit vector compares whose lane masks are counted by s_bcnt1 - 64 comparisons per
    // vector instruction and no per-lane counters (a lane-per-pair count costs several times the vector work).
    // whole_tiles > 0 (a population of at most 8 real tiles, padded to 16): the real tiles are staged whole and the tiles of
    // padding not at all - no samples to fetch, no v* to rank (2.6 us in front of the copies), and every key's rank is exact
    if (whole_tiles == 0)
        for (uint32_t e = tid; e < ns; e += kSelThreads) {
            smp[e] = samples[e];
            smi[e] = samples[ns + e];
        }
    const uint32_t kth = (need + kSelQuantum - 1) / kSelQuantum;
    if (whole_tiles != 0) {
        if (tid == 0) vstar_s = ~0ull;
    } else if (kth > ns) {
        if (tid == 0) vstar_s = ~0ull; // more rows wanted than the samples can vouch for: stage everything
    } else {
        unsigned long long key[R]; // (bits << 32) | index: unique, so the ranks are a permutation
#pragma unroll
        for (uint32_t r = 0; r < R; ++r)
            key[r] = ((unsigned long long)samples[lane + kWave * r] << 32) | samples[ns + lane + kWave * r];
#pragma unroll
        for (uint32_t r = 0; r < R; ++r) {
#pragma unroll
            for (uint32_t l = 0; l < 4; ++l) {
                const uint32_t src_lane = 4u * wave + l;
                const unsigned long long v = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(key[r] >> 32), (int)src_lane) << 32) |
                                             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key[r], (int)src_lane);
                uint32_t below = 0;
#pragma unroll
                for (uint32_t r2 = 0; r2 < R; ++r2) below += (uint32_t)__popcll(__ballot(key[r2] < v));
                if (below == kth - 1 && lane == 0) vstar_s = v; // exactly one sample has this rank
            }
        }
    }
    __syncthreads();
    SOTS_PHASE(1);
    const uint32_t vstar_b = (uint32_t)(vstar_s >> 32), vstar_i = (uint32_t)vstar_s;

    // ---- staged prefix of every tile (256, 512, 768 or 1024 keys), exclusive scan -> off[] ---------
    {
        uint32_t quanta = 0; // prefix length / 256
        if (whole_tiles != 0) {
            quanta = tid < whole_tiles ? kSelSamples : 0u;
        } else if (tid < tiles) {
            uint32_t m = 0;
#pragma unroll
            for (uint32_t j = 0; j < kSelSamples; ++j) {
                const uint32_t sb = smp[tid * kSelSamples + j], si = smi[tid * kSelSamples + j];
                m += (sb < vstar_b || (sb == vstar_b && si <= vstar_i)) ? 1u : 0u;
            }
            quanta = m >= kSelSamples ? kSelSamples : m + 1u;
        }
        // prefix sum inside the wavefront from four ballots (quanta is 0..4), across wavefronts through LDS
        uint32_t before_lane = 0, wave_sum = 0;
#pragma unroll
        for (uint32_t q = 1; q <= kSelSamples; ++q) {
            const uint64_t mask = __ballot(quanta >= q);
            before_lane += __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            wave_sum += (uint32_t)__popcll(mask);
        }
        if (lane == 0) wave_tot[wave] = wave_sum;
        __syncthreads();
        uint32_t before = before_lane;
        for (uint32_t w = 0; w < wave; ++w) before += wave_tot[w];
        if (tid < tiles) {
            off[tid + 1] = (before + quanta) * kSelQuantum;
            for (uint32_t q = 0; q < quanta; ++q) chunk_tile[before + q] = (uint16_t)tid;
        }
        if (tid == 0) off[0] = 0;
    }
    __syncthreads();
    SOTS_PHASE(2);
    const uint32_t total = off[tiles];

    // ---- this workgroup's keys: an equal share of the staged positions.  Their fitness bits and row
    // indices are requested now and land while the prefixes are staged -------------------------------
    const uint32_t share = (total + gridDim.x - 1) / gridDim.x; // <= kSelMaxOwn by the launcher's grid
    uint32_t my_tile = 0xFFFFFFFFu, my_pos = 0, my_bits = 0xFFFFFFFFu, my_src = 0; // slot `tid`
    {
        const uint32_t g = blockIdx.x * share + tid;
        if (tid < share && g < total) {
            my_tile = chunk_tile[g / kSelQuantum];
            my_pos = g - off[my_tile];
            my_bits = kbits[(size_t)my_tile * kSelTile + my_pos];
            my_src = kidx[(size_t)my_tile * kSelTile + my_pos];
        }
    }
    SOTS_PHASE(3);

    // ---- passes over the staged prefixes ---------------------------------------------------------
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    const uint32_t sub = tid & (kSelLanesPerKey - 1);
    constexpr uint32_t kRowRounds = kSelMaxOwn / kSelKeysPerRound, kColsPerLane = (2 * SOTS_MAX_DIMS + 1 + kSelLanesPerKey - 1) / kSelLanesPerKey;
    float val[kRowRounds][kColsPerLane];
    const uint32_t passes = off[tiles - 1] / kSelWindow + 1;
    uint32_t ta = 0;
    for (uint32_t pass = 0; pass < passes; ++pass) {
        const uint32_t w0 = pass * kSelWindow;
        // first tile that starts at or beyond the end of this window: tiles [ta, tb) are staged
        uint32_t tb = tiles;
        if (w0 + kSelWindow < total) {
            const uint32_t t = chunk_tile[(w0 + kSelWindow) / kSelQuantum]; // the tile that owns that position
            tb = off[t] == w0 + kSelWindow ? t : t + 1;
        }
        if (pass) __syncthreads(); // the previous pass's searches are done with the staging buffer
        // wavefront w copies tiles ta + w, ta + w + 16, ...: lane m fetches the bounds of the m-th of them, then
        // the copies are issued back to back with wavefront-uniform (readlane) arguments
        {
            const uint32_t mine = ta + wave + lane * kWaves;
            const uint32_t b0 = mine < tb ? off[mine] : 0u, b1 = mine < tb ? off[mine + 1] : 0u;
            uint32_t m = 0;
            for (uint32_t tt = ta + wave; tt < tb; tt += kWaves, ++m) {
                const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, (int)m), o1 = (uint32_t)__builtin_amdgcn_readlane((int)b1, (int)m);
                const uint32_t *__restrict__ src = kbits + (size_t)tt * kSelTile;
                for (uint32_t e = 0; e < o1 - o0; e += 4 * kWave) // a multiple of 256: whole wavefront instructions
                    __builtin_amdgcn_global_load_lds(src + e + 4 * lane, (lds_ptr_t)(staged + (o0 - w0) + kSelSkew * (tt - ta) + e), 16, 0, 0);
            }
        }
        if (pass == 0) { // the own-key loads were issued before the copies, so they have landed too
            const bool live = my_tile != 0xFFFFFFFFu && my_bits != kSelPadBits; // padding keys move nothing
            if (tid < kSelMaxOwn) {
                own_tile[tid] = live ? my_tile : 0xFFFFFFFFu;
                own_pos[tid] = my_pos;
                own_bits[tid] = my_bits;
                own_src[tid] = my_src;
                own_rank[tid] = 0;
            }
        }
        SOTS_PHASE(4);
        __builtin_amdgcn_s_waitcnt(0); // vmcnt(0): this wavefront's copies have landed
        __syncthreads();
        SOTS_PHASE(5);
        if (pass == 0) {
            // the rows of this workgroup's keys are requested now, before it is known which of them made it (about
            // half do): the loads fly while the ranks are computed, eight lanes per key, lane `sub` holding row
            // elements sub, sub + 8, ...
#pragma unroll
            for (uint32_t r = 0; r < kRowRounds; ++r) {
                const uint32_t slot = r * kSelKeysPerRound + tid / kSelLanesPerKey;
                if (slot >= share || own_tile[slot] == 0xFFFFFFFFu) continue;
                const uint32_t src = own_src[slot];
#pragma unroll
                for (uint32_t q = 0; q < kColsPerLane; ++q) {
                    const uint32_t c = sub + q * kSelLanesPerKey;
                    if (c < d) val[r][q] = vin[(size_t)src * d + c];
                    else if (c < 2 * d) val[r][q] = sin[(size_t)src * d + (c - d)];
                    else if (c < 2 * d + 1) val[r][q] = fin[src];
                }
            }
        }
        for (uint32_t r0 = 0; r0 < share; r0 += kSelThreads / LPK) {
            const uint32_t slot = r0 + tid / LPK, ssub = tid % LPK;
            const uint32_t t_own = slot < share ? own_tile[slot] : 0xFFFFFFFFu;
            const bool live = t_own != 0xFFFFFFFFu;
            const uint32_t xb = live ? own_bits[slot] : 0u, pos_own = live ? own_pos[slot] : 0u;
            uint32_t count = 0;
            // kSelChains tiles per lane at a time, their binary searches advanced level by level.  A search keeps
            // the ABSOLUTE position in staged[] (one add less per step).
            for (uint32_t t0 = ta + ssub; t0 < tb; t0 += LPK * kSelChains) {
                uint32_t base[kSelChains], n[kSelChains], thr[kSelChains], p[kSelChains];
#pragma unroll
                for (uint32_t m = 0; m < kSelChains; ++m) {
                    const uint32_t tt = t0 + m * LPK;
                    const bool in = tt < tb && live;
                    const uint32_t o0 = in ? off[tt] : w0, o1 = in ? off[tt + 1] : w0;
                    base[m] = o0 - w0 + (in ? kSelSkew * (tt - ta) : 0u);
                    n[m] = o1 - o0;
                    thr[m] = in ? xb + (tt < t_own ? 1u : 0u) : 0u; // xb <= 0xFFFFFFFD for a live key: no wrap; never 0
                    p[m] = base[m];
                }
                // levels of a quantum or more: only while the step stays inside the prefix (prefix lengths are
                // multiples of the quantum, not powers of two)
#pragma unroll
                for (uint32_t step = kSelTile / 2; step >= kSelQuantum; step >>= 1) {
#pragma unroll
                    for (uint32_t m = 0; m < kSelChains; ++m)
                        p[m] += (p[m] + step <= base[m] + n[m] && staged[p[m] + step - 1] < thr[m]) ? step : 0u;
                }
                // a search that has reached the end of its prefix stops (threshold 0: nothing compares below it)
#pragma unroll
                for (uint32_t m = 0; m < kSelChains; ++m) thr[m] = p[m] - base[m] >= n[m] ? 0u : thr[m];
#pragma unroll
                for (uint32_t step = kSelQuantum / 2; step >= 1; step >>= 1) {
#pragma unroll
                    for (uint32_t m = 0; m < kSelChains; ++m) p[m] += staged[p[m] + step - 1] < thr[m] ? step : 0u;
                }
#pragma unroll
                for (uint32_t m = 0; m < kSelChains; ++m) {
                    const uint32_t tt = t0 + m * LPK;
                    const uint32_t lb = p[m] - base[m] + (staged[p[m]] < thr[m] ? 1u : 0u);
                    count += (tt == t_own && tt < tb) ? pos_own : lb; // the own tile counts in the pass that stages it
                }
            }
            count += lane_xor<1>(count);
            if constexpr (LPK >= 4) count += lane_xor<2>(count);
            if constexpr (LPK == 8) count += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)count, 0x141, 0xf, 0xf, false); // row_half_mirror
            static_assert(LPK == 2 || LPK == 4 || LPK == 8, "the partial counts are added over 2, 4 or 8 neighbouring lanes");
            if (ssub == 0 && live) own_rank[slot] += count; // one writer per slot
        }
        ta = tb;
    }
    __syncthreads();
    SOTS_PHASE(6);

    // ---- rows of the keys that made it (requested above) -------------------------------------------
    const uint32_t width = 2 * d + 1;
#pragma unroll
    for (uint32_t r = 0; r < kRowRounds; ++r) {
        const uint32_t slot = r * kSelKeysPerRound + tid / kSelLanesPerKey;
        if (slot >= share || own_tile[slot] == 0xFFFFFFFFu) continue;
        const uint32_t dst = own_rank[slot];
        if (dst >= need || ex_immigrant_row(ex, dst)) continue;
#pragma unroll
        for (uint32_t q = 0; q < kColsPerLane; ++q) {
            const uint32_t c = sub + q * kSelLanesPerKey;
            if (c < d) vout[(size_t)dst * d + c] = val[r][q];
            else if (c < 2 * d) sout[(size_t)dst * d + (c - d)] = val[r][q];
            else if (c < width) fout[dst] = val[r][q];
            if (c < width) ex_sink(ex, dst, c, d, val[r][q]);
        }
    }
    SOTS_PHASE(7);
}

// ------------------------------------------------------------------------------------
// Batched real FFT + fitness for N = 4096 (and 2048) WITHOUT LDS exchanges: one wavefront per row,
// E = N/128 complex points per lane in registers, the 64-point sub-transforms ACROSS lanes.
//
// The workgroup-per-row kernel above makes four dependent LDS round trips per row behind workgroup
// barriers, with 8 wavefronts per CU (0.40 of the HBM roofline at N = 4096).  Here a row never leaves
// the registers of its wavefront.  With M = 64 E complex points z[n], n = l + 64 j (lane l, register j):
//   Z[q + E p] = sum_l W_64^{l p} ( W_M^{l q} sum_j z[l + 64 j] W_E^{j q} )
//   1. an E-point DFT over j inside every lane (radix-2 DIF in registers; register r ends with q = bitrev(r));
//   2. the twiddle W_M^{l q};
//   3. for every register a 64-point DFT over the LANES: six radix-2 DIF stages whose partner lane is l ^ h,
//      h = 32, 16, 8, 4, 2, 1 - v_permlane32_swap, ds_swizzle (LDS crossbar, no LDS memory), DPP row rotate,
//      DPP quad permutes - a lane with bit h clear keeps a + b, the other one (a - b) W_2h^{l mod h}
//      (both as fma(own, +-1, partner) times a per-lane constant).  Lane l ends with p = bitrev6(l).
// So lane l, register r holds Z[k], k = bitrev(r) + E bitrev6(l): E consecutive bins per lane.  The real-input
// split needs Z[M - k]: register bitrev(E - q) of lane l ^ 63 (q >= 1) - one ds_bpermute per dword - and every
// lane turns E bins into squared errors: for half of its registers BOTH bins of the pair (k, M - k), its own and its
// partner lane's (round 4, pair2x: the pair's sum and twiddled difference are worked out once, not once in each of the two
// lanes).  Window, step-2 twiddles, split twiddles and target are laid out per (lane, register) in LDS once per workgroup
// (61 KiB, read as 16-byte words).
// ------------------------------------------------------------------------------------
#pragma clang fp contract(on)
constexpr int x_bitrev(int v, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}
template <int LOG2N> constexpr int x_points() { return (1 << LOG2N) / 2 / kWave; }
#ifndef SOTS_X_WAVES12
#define SOTS_X_WAVES12 16
#endif
#ifndef SOTS_X_SADDR
#define SOTS_X_SADDR 1
#endif
#ifndef SOTS_X_PAIR_SPLIT
#define SOTS_X_PAIR_SPLIT 1 // both bins of a pair (k, M - k) in one lane (0: every lane its own E bins, rounds 2-3)
#endif
// wavefronts (independent rows) per workgroup: what the registers allow (MODE 0, the spectrum writer of the stage-separated
// path, needs a few more than the fused kernel)
template <int LOG2N, int MODE = 1> constexpr int x_waves() { return LOG2N >= 13 ? 8 : LOG2N == 12 ? (MODE == 0 ? 12 : SOTS_X_WAVES12) : 16; }
template <int LOG2N> constexpr bool x_applies() { return LOG2N >= 10 && LOG2N <= 13; }

// target bin of entry (lane l, register r) of k_fft_x's target table.  With SOTS_X_PAIR_SPLIT a lane turns out its own bin
// for the first register of a pair (RR, R2 = bitrev(E - bitrev(RR))) and its PARTNER lane's bin M - k for the second, so the
// second register's entry holds that bin's target: the partner is lane l ^ 63 (p' = 63 - p), its bin at r is q + E (63 - p).
template <int E, int EB>
__device__ __forceinline__ uint32_t x_target_bin(uint32_t l, uint32_t r)
{
    const uint32_t q = __brev(r) >> (32 - EB), pp = __brev(l) >> 26;
#if SOTS_X_PAIR_SPLIT
    const uint32_t r2 = q == 0 ? 0u : __brev((uint32_t)E - q) >> (32 - EB);
    return r2 < r ? q + E * (63u - pp) : q + E * pp;
#else
    return q + E * pp;
#endif
}
// f(ic<I>{}) for I = FIRST .. LAST-1 with I a compile-time constant inside f (register arrays are indexed with it)
template <int FIRST, int LAST, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (FIRST < LAST) {
        f(ic<FIRST>{});
        static_for<FIRST + 1, LAST>(f);
    }
}

// Complex values are 2-vectors here (register pairs): the transform is VALU-bound (85 % busy at 2400 one-float
// instructions per row in its first form), and v_pk_add/mul/fma_f32 do a complex add, or half a complex multiply, per
// instruction.
#ifndef SOTS_X_NT
#define SOTS_X_NT 1
#endif
__device__ __forceinline__ v2f_t x_row_load(const float2 *p)
{
#if SOTS_X_NT
    return __builtin_nontemporal_load(reinterpret_cast<const v2f_t *>(p)); // rows are read once
#else
    return *reinterpret_cast<const v2f_t *>(p);
#endif
}
// one radix-2 DIF stage of the E-point transform over the registers: pairs (a, a + H) inside groups of 2 H
template <int H, int E, int N>
__device__ __forceinline__ void x_reg_stage(v2f_t (&x)[E], const float2 *__restrict__ tw)
{
    static_for<0, E / 2>([&](auto i_tag) {
        constexpr int I = decltype(i_tag)::value, A0 = (I / H) * 2 * H, A = A0 + I % H;
        constexpr int IDX = (A - A0) * (E / (2 * H)); // the twiddle is W_E^IDX
        const v2f_t u = x[A] + x[A + H], d = x[A] - x[A + H];
        x[A] = u;
        if constexpr (IDX == 0) x[A + H] = d;
        else if constexpr (IDX == E / 4) x[A + H] = xc_mul_neg_i(d);
        else x[A + H] = xc_mul(d, xv(tw[IDX * (N / E)]));
    });
    if constexpr (H > 1) x_reg_stage<H / 2, E, N>(x, tw);
}

template <uint32_t J> __device__ __forceinline__ float lane_xor_f(float v)
{
    if constexpr (J == 8) return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xf, 0xf, true)); // row_ror:8
    else return __uint_as_float(lane_xor<J>(__float_as_uint(v)));
}

// one radix-2 DIF stage over the lanes, partner l ^ H, on every register: own * (+-1) + partner, times the lane's
// constant (1 where bit H is clear)
template <uint32_t H, int E>
__device__ __forceinline__ void x_lane_stage(v2f_t (&x)[E], float sgn, v2f_t wl)
{
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const v2f_t p = v2f_t{lane_xor_f<H>(x[r].x), lane_xor_f<H>(x[r].y)};
#if SOTS_XC_ONE_ASM
        if constexpr (H != 1) { // the butterfly and its twiddle as one statement (see xc_mul)
            v2f_t t, u = x[r];
            asm("v_pk_fma_f32 %0, %0, %2, %3\n\t"
                "v_pk_mul_f32 %1, %0, %4 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
                "v_pk_fma_f32 %0, %0, %4, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
                : "+v"(u), "=&v"(t) : "v"(v2f_t{sgn, sgn}), "v"(p), "v"(wl));
            x[r] = u;
        } else
#endif
        {
        const v2f_t t = x[r] * v2f_t{sgn, sgn} + p;
        x[r] = H == 1 ? t : xc_mul(t, wl);
        }
        if (r % 4 == 3) __builtin_amdgcn_sched_barrier(0); // (four registers' partner fetches in flight at a time)
    }
}

// The stages with partner l ^ 32 and l ^ 16 take the registers two at a time: v_permlane32_swap / v_permlane16_swap
// (new in gfx950) exchange the upper half (odd rows of 16) of one register with the lower half (even rows) of the other, so
// after swapping A with B the lanes with bit H clear hold BOTH inputs of register A's butterfly and the others both
// inputs of B's; every lane computes one whole butterfly (sum, difference times W_2H^{l mod H}) and a second pair of
// swaps puts the results where they belong.  Four swaps and four packed operations per pair of complex values, no
// per-lane signs, no select.
template <uint32_t H>
__device__ __forceinline__ void x_swap2(v2f_t &a, v2f_t &b)
{
    if constexpr (H == 32) {
        const auto r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
        a = v2f_t{__uint_as_float(r0[0]), __uint_as_float(r1[0])}, b = v2f_t{__uint_as_float(r0[1]), __uint_as_float(r1[1])};
    } else {
        const auto r0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
        a = v2f_t{__uint_as_float(r0[0]), __uint_as_float(r1[0])}, b = v2f_t{__uint_as_float(r0[1]), __uint_as_float(r1[1])};
    }
}
template <uint32_t H, int E>
__device__ __forceinline__ void x_lane_stage_pairs(v2f_t (&x)[E], v2f_t w)
{
#pragma unroll
    for (int r = 0; r < E; r += 2) {
        x_swap2<H>(x[r], x[r + 1]);
        v2f_t sum = x[r] + x[r + 1], dif = xc_mul(x[r] - x[r + 1], w);
        x_swap2<H>(sum, dif);
        x[r] = sum, x[r + 1] = dif;
        if (r % 4 == 2) __builtin_amdgcn_sched_barrier(0);
    }
}

// WG: wavefronts per workgroup.  The default fills a CU with one workgroup (the tables are made once per workgroup); a SMALL
// population - fewer such workgroups than CUs - runs workgroups of four instead, a wavefront per SIMD on four times as many
// CUs: a row is then transformed at a lone wavefront's pace, not at a quarter of it (1024 rows of N = 4096: 24.6 -> ... us).
// floats of the per-(lane, register) tables as they lie in LDS: step-2 twiddles, split twiddles, window (float2 each), target
template <int LOG2N> constexpr int x_table_floats() { return 3 * 2 * kWave * (x_points<LOG2N>() + 2) + kWave * (x_points<LOG2N>() + 4); }

// `image` (fused kernel with window only; may be null): the four tables as they lie in LDS, made once per target by
// k_x_tables - the workgroups then copy them in by LDS-DMA, one round trip, instead of four dependent-index loads per entry
#ifndef SOTS_X_MIN_WAVES // (experiment: minimum wavefronts per SIMD the register allocation must allow, e.g. 5 with two workgroups of ten)
#define SOTS_X_MIN_WAVES 1
#endif
template <int LOG2N, int MODE, bool WIN, int WG = x_waves<LOG2N, MODE>()>
__global__ __launch_bounds__((WG * kWave), (LOG2N == 12 && MODE == 1 ? SOTS_X_MIN_WAVES : 1)) void k_fft_x(const float *__restrict__ audio, float *__restrict__ spectrum,
                                                                   const float *__restrict__ target, float *__restrict__ fitness,
                                                                   const float2 *__restrict__ tw, const float *__restrict__ window,
                                                                   uint32_t p_len, float inv_n, float inv_wf, uint32_t pitch,
                                                                   const float *__restrict__ image)
{
    constexpr int N = 1 << LOG2N, M = N / 2, E = x_points<LOG2N>(), EB = (E == 2 ? 1 : E == 4 ? 2 : E == 8 ? 3 : E == 16 ? 4 : E == 32 ? 5 : 6), W = WG;
    constexpr int S2 = E + 2, S1 = E + 4; // lane strides of the float2 / float tables (16-byte reads, spread over the banks)
    __shared__ __attribute__((aligned(16))) float xt_s[x_table_floats<LOG2N>()]; // (the variants without window or target leave theirs unused)
    float2 *const tw2_s = reinterpret_cast<float2 *>(xt_s), *const tws_s = tw2_s + kWave * S2, *const win_s = tws_s + kWave * S2;
    float *const tgt_s = reinterpret_cast<float *>(win_s + kWave * S2);
    __shared__ uint32_t next_s; // the workgroup's rows are dealt out as its wavefronts ask for them (below)
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    if (tid == 0) next_s = W;
#ifdef SOTS_STAMP
    const unsigned long long xs_begin = __builtin_amdgcn_s_memrealtime(), xs_begin_clk = __builtin_amdgcn_s_memtime();
    unsigned long long xs_wait = 0, xs_rows = 0, xs_split = 0;
#endif
    if (MODE == 1 && WIN && image != nullptr) { // (launch_x_tables makes images only where the tables are whole 1 KiB pieces: N >= 2048)
        typedef __attribute__((address_space(3))) void *lds_ptr_t;
        for (uint32_t ch = wave; ch < (uint32_t)x_table_floats<LOG2N>() / 256u; ch += W)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t *>(image) + ch * 256u + lane * 4u, (lds_ptr_t)(xt_s + ch * 256u), 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0); // vmcnt(0): this wavefront's pieces have landed (the barrier below covers the others')
    } else
    for (uint32_t e = tid; e < kWave * E; e += W * kWave) {
        const uint32_t l = e / E, r = e % E;
        const uint32_t q = __brev(r) >> (32 - EB), pp = __brev(l) >> 26, k = q + E * pp;
        tw2_s[l * S2 + r] = tw[2u * l * q]; // W_M^{l q} = W_N^{2 l q}
        tws_s[l * S2 + r] = tw[k];          // W_N^k
        if constexpr (WIN) win_s[l * S2 + r] = reinterpret_cast<const float2 *>(window)[l + kWave * r]; // r = input register j here
        if constexpr (MODE == 1) tgt_s[l * S1 + r] = target[x_target_bin<E, EB>(l, r)];
    }
    // per-lane constants of the six lane stages
    const float sg8 = (lane & 8u) ? -1.0f : 1.0f;
    const float sg4 = (lane & 4u) ? -1.0f : 1.0f, sg2 = (lane & 2u) ? -1.0f : 1.0f, sg1 = (lane & 1u) ? -1.0f : 1.0f;
    const v2f_t one = v2f_t{1.0f, 0.0f};
    const v2f_t w32 = xv(tw[(lane & 31u) * (N / 64)]), w16 = xv(tw[(lane & 15u) * (N / 32)]); // (every lane makes a difference there)
    const v2f_t w8 = (lane & 8u) ? xv(tw[(lane & 7u) * (N / 16)]) : one, w4 = (lane & 4u) ? xv(tw[(lane & 3u) * (N / 8)]) : one;
    const v2f_t w2 = (lane & 2u) ? xv(tw[(lane & 1u) * (N / 4)]) : one;
    const uint32_t pp = __brev(lane) >> 26;
    const int addr_flip = (int)((lane ^ 63u) * 4u);                        // partner lane of the registers with q >= 1
    const int addr_zero = (int)((__brev((64u - pp) & 63u) >> 26) * 4u);    // ... and of register 0 (q = 0): Z[E (64 - p)]
    const float half_scale = 0.5f * (inv_n * inv_wf);
    __syncthreads();

    // Rows: workgroup b owns rows b W + w + i grid W (w < W, i = 0, 1, ...).  Its wavefronts start with i = 0, wavefront w
    // on row b W + w, and then take the workgroup's remaining rows in that order AS THEY FINISH: the SIMD issues for its
    // oldest wavefront first, so with a fixed deal of 8 rows each the wavefronts of one launch end anywhere between 77
    // and 148 us (N = 4096, P = 32768) and the last ones run alone, far below the issue rate (tools/fftx_probe.py).
    // Which wavefront transforms a row changes nothing in its result.
    uint32_t row = blockIdx.x * W + wave;
    if (row >= p_len) return;
#ifdef SOTS_STAMP
    const unsigned long long xs_tables = __builtin_amdgcn_s_memrealtime();
#endif
    v2f_t x[E];
    {
        const float2 *__restrict__ in = reinterpret_cast<const float2 *>(audio + (size_t)row * pitch);
#pragma unroll
        for (int j = 0; j < E; ++j) x[j] = x_row_load(in + lane + kWave * j);
    }
    while (true) {
        uint32_t take = 0;
        if (lane == 0) take = atomicAdd(&next_s, 1u); // (answered long before the split below needs it)
        take = __builtin_amdgcn_readfirstlane(take);
        const uint32_t nxt = blockIdx.x * W + take % W + (take / W) * gridDim.x * W;
        const bool more = nxt < p_len; // (rows past the end: every later take is past it too)
        __builtin_amdgcn_sched_barrier(0);
#ifdef SOTS_STAMP
        {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            xs_wait += __builtin_amdgcn_s_memtime() - t0, xs_rows += 1;
        }
#endif
        if constexpr (WIN) {
#pragma unroll
            for (int j = 0; j < E; j += 2) {
                const float4 wv = *reinterpret_cast<const float4 *>(&win_s[lane * S2 + j]);
                x[j] = x[j] * v2f_t{wv.x, wv.y};
                x[j + 1] = x[j + 1] * v2f_t{wv.z, wv.w};
                if (j % 8 == 6) __builtin_amdgcn_sched_barrier(0); // (table reads a few at a time: the row already fills 2 E registers)
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // 1. E-point DIF over the registers
        x_reg_stage<E / 2, E, N>(x, tw);
        __builtin_amdgcn_sched_barrier(0);
        // 2. W_M^{l q}
#pragma unroll
        for (int r = 0; r < E; r += 2) {
            const float4 t = *reinterpret_cast<const float4 *>(&tw2_s[lane * S2 + r]);
            if (r) x[r] = xc_mul(x[r], v2f_t{t.x, t.y}); // register 0: q = 0
            x[r + 1] = xc_mul(x[r + 1], v2f_t{t.z, t.w});
            if (r % 8 == 6) __builtin_amdgcn_sched_barrier(0);
        }
        // 3. 64-point DIF over the lanes
        x_lane_stage_pairs<32>(x, w32);
        __builtin_amdgcn_sched_barrier(0);
        x_lane_stage_pairs<16>(x, w16);
        __builtin_amdgcn_sched_barrier(0);
        x_lane_stage<8>(x, sg8, w8);
        __builtin_amdgcn_sched_barrier(0);
        x_lane_stage<4>(x, sg4, w4);
        __builtin_amdgcn_sched_barrier(0);
        x_lane_stage<2>(x, sg2, w2);
        __builtin_amdgcn_sched_barrier(0);
        x_lane_stage<1>(x, sg1, one);
        __builtin_amdgcn_sched_barrier(0);
        // split: this lane's E bins k = bitrev(r) + E p
#ifdef SOTS_STAMP
        const unsigned long long xs_t2 = __builtin_amdgcn_s_memtime();
#endif
        float acc = 0.0f;
#if !SOTS_X_PAIR_SPLIT
        v2f_t out_even = v2f_t{0.f, 0.f}; // MODE 0: bins leave two at a time, in bin order
        float2 *__restrict__ dst = MODE == 0 ? reinterpret_cast<float2 *>(spectrum + (size_t)row * (N + 8)) + E * pp : nullptr;
#endif
        // one bin: register RR of this lane with its partner Z[M - k]
        auto bin2x = [&](auto r_tag) -> v2f_t {
            constexpr int RR = decltype(r_tag)::value, Q = x_bitrev(RR, EB), R2 = Q == 0 ? 0 : x_bitrev(E - Q, EB);
            const int addr = Q == 0 ? addr_zero : addr_flip;
            const v2f_t zm = v2f_t{__int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(x[R2].x))),
                                   __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(x[R2].y)))};
            const v2f_t w = xv(tws_s[lane * S2 + RR]);
            const v2f_t ee = xc_add_conj(x[RR], zm), dd = xc_sub_conj(x[RR], zm);
            return ee + xc_mul_negi_w(dd, w); // 2 X[k] = (Z[k] + conj Z[M-k]) + W_N^k (-i) (Z[k] - conj Z[M-k])
        };
#if SOTS_X_PAIR_SPLIT
        // BOTH bins of the pair (k, M - k) from this lane's Z[k] (register RR) and the partner lane's Z[M - k] (its register
        // R2): ee and t = W_N^k (-i) dd once, 2 X[k] = ee + t and 2 conj X[M - k] = ee - t.  The partner lane does the same with
        // ITS register RR and this lane's R2, so every lane still turns out E bins - half of them its partner's - with half
        // the exchanges and a packed subtraction in place of a second add / subtract / multiply chain.
        auto pair2x = [&](auto r_tag, v2f_t &xa, v2f_t &xb) {
            constexpr int RR = decltype(r_tag)::value, Q = x_bitrev(RR, EB), R2 = x_bitrev(E - Q, EB);
            static_assert(Q != 0 && R2 != RR, "registers 0 and bitrev(E / 2) pair with themselves: bin2x");
            const v2f_t zm = v2f_t{__int_as_float(__builtin_amdgcn_ds_bpermute(addr_flip, __float_as_int(x[R2].x))),
                                   __int_as_float(__builtin_amdgcn_ds_bpermute(addr_flip, __float_as_int(x[R2].y)))};
            const v2f_t w = xv(tws_s[lane * S2 + RR]);
            const v2f_t ee = xc_add_conj(x[RR], zm), t = xc_mul_negi_w(xc_sub_conj(x[RR], zm), w);
            xa = ee + t, xb = ee - t;
        };
#endif
        if constexpr (MODE == 1) {
            // Registers RR and R2 = bitrev(E - bitrev(RR)) need each other and nobody else: the bins are taken in such
            // pairs (this is the summation order, k_fitness_x repeats it), and a pair that is done is free - the NEXT
            // row's loads into those two registers go out at once, so by the end of the split most of the next row is on
            // its way without a second set of registers.
            // (a wavefront's last row has no successor: it re-reads row 0, which every wavefront of the launch then finds in
            // L2 - loads under `if (more)` cost the compiler its register allocation)
#if SOTS_X_SADDR
            // the next row's address as a uniform base (scalar registers: audio + row offset) plus the lane's byte offset, instead
            // of the loop-invariant 64-bit pointer audio + lane offset plus a uniform row offset: with the pair split that pointer
            // was the one value too many for the 128 registers (an 8-byte spill, reloaded once per row; 130 against 127 us)
            // (the builtin returns a SIGNED int - the halves are widened as unsigned: the first attempt sign-extended the low
            // half and faulted on rows whose address has bit 31 set)
            const uint64_t in_next_u = reinterpret_cast<uint64_t>(audio + (size_t)(more ? nxt : 0u) * pitch);
            const uint32_t in_next_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)in_next_u);
            const uint32_t in_next_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(in_next_u >> 32));
            typedef const __attribute__((address_space(1))) char *x_gptr_t; // global, not generic: global_load, not flat_load
            const x_gptr_t in_next_base = (x_gptr_t)(((uint64_t)in_next_hi << 32) | (uint64_t)in_next_lo);
            const uint32_t lane_bytes = lane * 8u;
#else
            const float2 *__restrict__ in_next = reinterpret_cast<const float2 *>(audio + (size_t)(more ? nxt : 0u) * pitch);
#endif
            static_for<0, E>([&](auto r_tag) {
                constexpr int RR = decltype(r_tag)::value, Q = x_bitrev(RR, EB), R2 = Q == 0 ? 0 : x_bitrev(E - Q, EB);
                if constexpr (R2 >= RR) {
#if SOTS_X_PAIR_SPLIT
                    if constexpr (R2 != RR) {
                        v2f_t xa, xb;
                        pair2x(ic<RR>{}, xa, xb);
                        acc += bin_error(make_float2(xa.x, xa.y), tgt_s[lane * S1 + RR], half_scale);
                        acc += bin_error(make_float2(xb.x, xb.y), tgt_s[lane * S1 + R2], half_scale); // the partner lane's bin M - k: x_target_bin
                    } else {
                        const v2f_t xa = bin2x(ic<RR>{});
                        acc += bin_error(make_float2(xa.x, xa.y), tgt_s[lane * S1 + RR], half_scale);
                    }
#else
                    const v2f_t xa = bin2x(ic<RR>{});
                    acc += bin_error(make_float2(xa.x, xa.y), tgt_s[lane * S1 + RR], half_scale);
                    if constexpr (R2 != RR) {
                        const v2f_t xb = bin2x(ic<R2>{});
                        acc += bin_error(make_float2(xb.x, xb.y), tgt_s[lane * S1 + R2], half_scale);
                    }
#endif
#ifndef SOTS_X_RECYCLE
#define SOTS_X_RECYCLE 1
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (SOTS_X_RECYCLE != 0) {
#if SOTS_X_SADDR
                        typedef const __attribute__((address_space(1))) v2f_t *x_gv2_t;
                        x[RR] = __builtin_nontemporal_load((x_gv2_t)(in_next_base + (lane_bytes + 8u * kWave * RR)));
                        if constexpr (R2 != RR) x[R2] = __builtin_nontemporal_load((x_gv2_t)(in_next_base + (lane_bytes + 8u * kWave * R2)));
#else
                        x[RR] = x_row_load(in_next + lane + kWave * RR);
                        if constexpr (R2 != RR) x[R2] = x_row_load(in_next + lane + kWave * R2);
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            });
        } else
#if SOTS_X_PAIR_SPLIT
        static_for<0, E>([&](auto r_tag) { // MODE 0, the fused kernel's pairs: the same values, written where they belong
            constexpr int RR = decltype(r_tag)::value, Q = x_bitrev(RR, EB), R2 = Q == 0 ? 0 : x_bitrev(E - Q, EB);
            if constexpr (R2 >= RR) {
                float2 *__restrict__ bins = reinterpret_cast<float2 *>(spectrum + (size_t)row * (N + 8));
                const uint32_t k = E * pp + Q;
                if constexpr (R2 != RR) {
                    v2f_t xa, xb;
                    pair2x(ic<RR>{}, xa, xb);
                    bins[k] = make_float2(0.5f * xa.x, 0.5f * xa.y);
                    bins[M - k] = make_float2(0.5f * xb.x, -0.5f * xb.y);
                } else {
                    const v2f_t xa = bin2x(ic<RR>{});
                    bins[k] = make_float2(0.5f * xa.x, 0.5f * xa.y);
                }
            }
            if constexpr (RR % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        });
#else
        static_for<0, E>([&](auto i_tag) {
            // MODE 0 walks the bins in order
            constexpr int I = decltype(i_tag)::value, RR = MODE == 0 ? x_bitrev(I, EB) : I;
            constexpr int Q = x_bitrev(RR, EB), R2 = Q == 0 ? 0 : x_bitrev(E - Q, EB);
            const int addr = Q == 0 ? addr_zero : addr_flip;
            const v2f_t zm = v2f_t{__int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(x[R2].x))),
                                   __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(x[R2].y)))};
            const v2f_t w = xv(tws_s[lane * S2 + RR]);
            // 2 X[k] = (Z[k] + conj Z[M-k]) + W_N^k (-i) (Z[k] - conj Z[M-k])
            const v2f_t ee = xc_add_conj(x[RR], zm), dd = xc_sub_conj(x[RR], zm);
            const v2f_t x2 = ee + xc_mul_negi_w(dd, w);
            if constexpr (MODE == 0) {
                if constexpr (Q % 2 == 0) out_even = x2 * v2f_t{0.5f, 0.5f};
                else *reinterpret_cast<float4 *>(dst + Q - 1) = make_float4(out_even.x, out_even.y, 0.5f * x2.x, 0.5f * x2.y);
            } else {
                acc += bin_error(make_float2(x2.x, x2.y), tgt_s[lane * S1 + RR], half_scale);
            }
            if constexpr (I % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        });
#endif
        if constexpr (MODE == 0) {
            // the Nyquist bin X[M] = Re Z0 - Im Z0, from Z0 itself: lane 0 still holds it in register 0
            if (lane == 0) reinterpret_cast<float2 *>(spectrum + (size_t)row * (N + 8))[M] = make_float2(x[0].x - x[0].y, 0.0f);
        } else {
            acc = wave_sum(acc);
            if (lane == 0) fitness[row] = acc;
        }
#ifdef SOTS_STAMP
        xs_split += __builtin_amdgcn_s_memtime() - xs_t2;
        if (lane == 0 && blockIdx.x * W + wave < 2048) { // per wavefront: 8 words
            unsigned long long *o = g_stamps + (blockIdx.x * W + wave) * 8;
            o[0] = xs_begin, o[1] = xs_tables, o[2] = __builtin_amdgcn_s_memrealtime(), o[3] = xs_rows;
            o[4] = xs_wait, o[5] = xs_split, o[6] = __builtin_amdgcn_s_memtime() - xs_begin_clk;
        }
#endif
        if (!more) break;
        if constexpr (MODE == 0 || SOTS_X_RECYCLE == 0) { // (MODE 1 has asked for the row already, pair by pair)
            const float2 *__restrict__ in = reinterpret_cast<const float2 *>(audio + (size_t)nxt * pitch);
            static_for<0, E>([&](auto j_tag) { x[decltype(j_tag)::value] = x_row_load(in + lane + kWave * decltype(j_tag)::value); });
        }
        row = nxt;
    }
}

// ------------------------------------------------------------------------------------
// Rows of N = 16384 and 32768 samples (round 4: the reference takes any audioLengthLog2, main.cpp:90, and renders 2^14
// samples of its best match, main.cpp:273): a WORKGROUP per row.  Such a row is 128 or 256 registers per lane of one
// wavefront, more than k_fft_x can hold; here its M = N / 2 complex points live in LDS (64 or 128 KiB), 512 threads run an
// in-place radix-2 DIF (natural order in, bit-reversed out) with one workgroup barrier per stage, and the real-input split
// reads Z[k] and Z[M - k] at their bit-reversed places.  Coverage, not speed: nothing of BASELINE's configs runs here.
//   MODE 0: spectrum row (bins 0 .. M) to memory; MODE 1: squared error against the target, bins 0 .. M-1
//   (Evolutionary_Strategy_CPU.hpp:235).  Thread t takes bins t, t + 512, ... in rising order, the wavefront totals are added
//   in wavefront order; k_fitness_big repeats map and order on a materialised row (bit-identical sums).
// ------------------------------------------------------------------------------------
constexpr int kBigThreads = 512;
template <int LOG2N> constexpr bool big_applies() { return LOG2N == 14 || LOG2N == 15; }

__device__ __forceinline__ float big_block_sum(float acc, float *__restrict__ red)
{
    acc = wave_sum(acc);
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    float total = 0.0f;
    if (threadIdx.x == 0)
        for (int w = 0; w < kBigThreads / kWave; ++w) total += red[w];
    return total; // (thread 0's)
}

template <int LOG2N, int MODE, bool WIN>
__global__ __launch_bounds__(kBigThreads) void k_fft_big(const float *__restrict__ audio, float *__restrict__ spectrum,
                                                         const float *__restrict__ target, float *__restrict__ fitness,
                                                         const float2 *__restrict__ tw, const float *__restrict__ window,
                                                         uint32_t p_len, float inv_n, float inv_wf, uint32_t pitch)
{
    constexpr uint32_t N = 1u << LOG2N, M = N / 2, LM = LOG2N - 1, T = kBigThreads;
    extern __shared__ float2 big_z[]; // M complex points (the launch asks for M * 8 + 64 bytes)
    float *__restrict__ red = reinterpret_cast<float *>(big_z + M);
    const uint32_t t = threadIdx.x;
    for (uint32_t row = blockIdx.x; row < p_len; row += gridDim.x) {
        const float2 *__restrict__ in = reinterpret_cast<const float2 *>(audio + (size_t)row * pitch);
        for (uint32_t i = t; i < M; i += T) {
            float2 v = in[i];
            if constexpr (WIN) {
                const float2 w = reinterpret_cast<const float2 *>(window)[i];
                v.x *= w.x, v.y *= w.y;
            }
            big_z[i] = v;
        }
        __syncthreads();
        // in-place radix-2 DIF: stage s pairs i and i + half, half = M >> (s + 1); the difference takes W_{2 half}^{j} = W_N^{j N / (2 half)}
        for (uint32_t st = 0; st < LM; ++st) {
            const uint32_t half = M >> (st + 1), tw_step = N / (2u * half);
            for (uint32_t b = t; b < M / 2; b += T) {
                const uint32_t j = b & (half - 1), i0 = ((b - j) << 1) + j, i1 = i0 + half;
                const float2 a = big_z[i0], c = big_z[i1];
                big_z[i0] = cadd(a, c);
                const float2 d = csub(a, c);
                big_z[i1] = j == 0 ? d : cmul(d, tw[j * tw_step]);
            }
            __syncthreads();
        }
        // Z[k] now lies at bitrev(k).  X[k] = (Z[k] + conj Z[M-k]) / 2 + W_N^k (-i) (Z[k] - conj Z[M-k]) / 2, Z[M] read as Z[0]
        auto bin = [&](uint32_t k) -> float2 {
            const uint32_t km = (M - k) & (M - 1);
            const float2 a = big_z[__brev(k) >> (32 - LM)], bz = big_z[__brev(km) >> (32 - LM)];
            const float2 ee = make_float2(0.5f * (a.x + bz.x), 0.5f * (a.y - bz.y)), dd = make_float2(0.5f * (a.x - bz.x), 0.5f * (a.y + bz.y));
            const float2 o = make_float2(dd.y, -dd.x); // -i dd
            const float2 w = tw[k];
            return make_float2(ee.x + (o.x * w.x - o.y * w.y), ee.y + (o.x * w.y + o.y * w.x));
        };
        if constexpr (MODE == 0) {
            float2 *__restrict__ dst = reinterpret_cast<float2 *>(spectrum + (size_t)row * (N + 8));
            for (uint32_t k = t; k < M; k += T) dst[k] = bin(k);
            if (t == 0) dst[M] = make_float2(big_z[0].x - big_z[0].y, 0.0f); // the Nyquist bin X[M] = Re Z0 - Im Z0
        } else {
            float acc = 0.0f;
            for (uint32_t k = t; k < M; k += T) acc += bin_error(bin(k), target[k], inv_n * inv_wf);
            const float total = big_block_sum(acc, red);
            if (t == 0) fitness[row] = total;
        }
        __syncthreads(); // the next row overwrites big_z and red
    }
}

template <int LOG2N>
__global__ __launch_bounds__(kBigThreads) void k_fitness_big(const float *__restrict__ spectrum, const float *__restrict__ target,
                                                             float *__restrict__ fitness, uint32_t p_len, float inv_n, float inv_wf)
{
    constexpr uint32_t N = 1u << LOG2N, M = N / 2, T = kBigThreads;
    __shared__ float red[kBigThreads / kWave];
    for (uint32_t row = blockIdx.x; row < p_len; row += gridDim.x) {
        const float2 *__restrict__ src = reinterpret_cast<const float2 *>(spectrum + (size_t)row * (N + 8));
        float acc = 0.0f;
        for (uint32_t k = threadIdx.x; k < M; k += T) acc += bin_error(src[k], target[k], inv_n * inv_wf);
        const float total = big_block_sum(acc, red);
        if (threadIdx.x == 0) fitness[row] = total;
        __syncthreads();
    }
}

// fitnessPopulation on materialised rows with k_fft_x's bin -> (lane, register) map and summation order
template <int LOG2N>
__global__ __launch_bounds__(x_waves<LOG2N>() * kWave) void k_fitness_x(const float *__restrict__ spectrum, const float *__restrict__ target,
                                                                       float *__restrict__ fitness, uint32_t p_len, float inv_n, float inv_wf)
{
    constexpr int N = 1 << LOG2N, E = x_points<LOG2N>(), EB = (E == 2 ? 1 : E == 4 ? 2 : E == 8 ? 3 : E == 16 ? 4 : E == 32 ? 5 : 6), W = x_waves<LOG2N>();
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t pp = __brev(lane) >> 26;
    for (uint32_t row = blockIdx.x * W + wave; row < p_len; row += gridDim.x * W) {
        const float2 *__restrict__ src = reinterpret_cast<const float2 *>(spectrum + (size_t)row * (N + 8)) + E * pp;
        float acc = 0.0f;
        static_for<0, E>([&](auto r_tag) { // k_fft_x's order: register pairs (RR, R2 = bitrev(E - bitrev(RR)))
            constexpr int RR = decltype(r_tag)::value, Q = x_bitrev(RR, EB), R2 = Q == 0 ? 0 : x_bitrev(E - Q, EB);
            if constexpr (R2 >= RR) {
                acc += bin_error(src[Q], target[E * pp + Q], inv_n * inv_wf);
#if SOTS_X_PAIR_SPLIT
                // the pair's second bin is the PARTNER lane's: M - k (k_fft_x's pair2x)
                if constexpr (R2 != RR)
                    acc += bin_error(reinterpret_cast<const float2 *>(spectrum + (size_t)row * (N + 8))[N / 2 - (E * pp + Q)], target[N / 2 - (E * pp + Q)], inv_n * inv_wf);
#else
                if constexpr (R2 != RR) acc += bin_error(src[x_bitrev(R2, EB)], target[E * pp + x_bitrev(R2, EB)], inv_n * inv_wf);
#endif
            }
        });
        acc = wave_sum(acc);
        if (lane == 0) fitness[row] = acc;
    }
}
// the table image of the fused kernel (k_fft_x<LOG2N, 1, true>), entry by entry what its workgroups would make themselves
template <int LOG2N>
__global__ __launch_bounds__(256) void k_x_tables(float *__restrict__ image, const float2 *__restrict__ tw,
                                                  const float *__restrict__ window, const float *__restrict__ target)
{
    constexpr int E = x_points<LOG2N>(), EB = (E == 2 ? 1 : E == 4 ? 2 : E == 8 ? 3 : E == 16 ? 4 : E == 32 ? 5 : 6), S2 = E + 2, S1 = E + 4;
    float2 *const tw2_s = reinterpret_cast<float2 *>(image), *const tws_s = tw2_s + kWave * S2, *const win_s = tws_s + kWave * S2;
    float *const tgt_s = reinterpret_cast<float *>(win_s + kWave * S2);
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < (uint32_t)(kWave * E); e += gridDim.x * blockDim.x) {
        const uint32_t l = e / E, r = e % E;
        const uint32_t q = __brev(r) >> (32 - EB), pp = __brev(l) >> 26, k = q + E * pp;
        tw2_s[l * S2 + r] = tw[2u * l * q];
        tws_s[l * S2 + r] = tw[k];
        win_s[l * S2 + r] = reinterpret_cast<const float2 *>(window)[l + kWave * r];
        tgt_s[l * S1 + r] = target[x_target_bin<E, EB>(l, r)];
    }
}
#pragma clang fp contract(off)

// ------------------------------------------------------------------------------------
// island exchange: rows of [fitness, values, steps]
// ------------------------------------------------------------------------------------
__global__ void k_pack_rows(const float *__restrict__ values, const float *__restrict__ steps,
                            const float *__restrict__ fitness, float *__restrict__ rows,
                            uint32_t first_row, uint32_t n_rows, uint32_t d)
{
    const uint32_t w = 2 * d + 1, total = n_rows * w;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const uint32_t r = e / w, c = e - r * w, src = first_row + r;
        rows[e] = c == 0 ? fitness[src] : c <= d ? values[src * d + (c - 1)] : steps[src * d + (c - 1 - d)];
    }
}

// rows [skip_first, skip_first + skip_count) of the source are passed over (an island's own
// block inside an all-gathered buffer)
__global__ void k_unpack_rows(float *__restrict__ values, float *__restrict__ steps,
                              float *__restrict__ fitness, const float *__restrict__ rows,
                              uint32_t first_row, uint32_t n_rows, uint32_t d, uint32_t skip_first,
                              uint32_t skip_count)
{
    const uint32_t w = 2 * d + 1, total = n_rows * w;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const uint32_t r = e / w, c = e - r * w, dst = first_row + r;
        const uint32_t sr = r < skip_first ? r : r + skip_count;
        const float v = rows[sr * w + c];
        if (c == 0) fitness[dst] = v;
        else if (c <= d) values[dst * d + (c - 1)] = v;
        else steps[dst * d + (c - 1 - d)] = v;
    }
}

inline uint32_t grid_for(uint64_t work, uint32_t threads, uint32_t cap = 2048)
{
    uint64_t g = (work + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (uint32_t)g;
}

} // namespace

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
hipError_t launch_init_population(hipStream_t st, float *values, float *steps, float *fitness,
                                  const PopDims &pd, uint32_t chunk)
{
    k_init_population<<<grid_for((uint64_t)pd.p * pd.d, 256), 256, 0, st>>>(values, steps, fitness, pd, chunk);
    return hipGetLastError();
}

hipError_t launch_recombine(hipStream_t st, const float *vin, const float *sin, float *vout, float *sout,
                            const PopDims &pd)
{
    k_recombine<<<grid_for((uint64_t)pd.p * pd.d, 256), 256, 0, st>>>(vin, sin, vout, sout, pd);
    return hipGetLastError();
}

hipError_t launch_mutate(hipStream_t st, float *values, float *steps, const PopDims &pd,
                         const MutateConsts &mc, uint32_t generation)
{
    k_mutate<<<grid_for((uint64_t)pd.p * pd.d, 256), 256, 0, st>>>(values, steps, pd, mc, generation);
    return hipGetLastError();
}

hipError_t launch_recombine_mutate(hipStream_t st, const float *vin, const float *sin, float *vout,
                                   float *sout, const PopDims &pd, const MutateConsts &mc,
                                   uint32_t generation)
{
    k_recombine_mutate<<<grid_for((uint64_t)pd.p * pd.d, 256), 256, 0, st>>>(vin, sin, vout, sout, pd, mc,
                                                                               generation);
    return hipGetLastError();
}

// Where k_synth_tp runs: at most 16 individuals per CU for the 2-operator voice (populations up to 4096; 22-27 us of
// synthesis against 32-35 at N = 1024; beyond that k_synth's cut kernels win), 12 / 9 / 5 for 3 / 4 operators in series /
// three parallel chains (what fits beside the table)
#ifndef SOTS_TP_2OP_MAX
#define SOTS_TP_2OP_MAX 16
#endif
#ifndef SOTS_OL_MIN_SHARE
#define SOTS_OL_MIN_SHARE 48
#endif
// Where k_synth_ol runs (same-box A/Bs, profiles/r04_experiments.md): the 4-operator voice from 65 individuals per CU (65 ... 240
// in one tile: 128 per CU 146-157 against 189 us, 192 per CU 199 against 297; beyond that in equal tiles: 256 per CU 302-307 against 313-317 + the
// variation launch, 313 per CU 389 against 612), the 3-operator voice at 65 ... 240 (128 per CU 91 against 95, 192 per CU 106 against
// 128; at 256 and 512 per CU k_synth wins: 126 against 152, 254 against 310)
bool synth_operators_in_lanes(uint32_t kind, uint32_t p, uint32_t num_cus)
{
#ifdef SOTS_SYNTH_NO_OL
    return false;
#else
    const uint32_t cus = num_cus ? num_cus : 256u, share = (p + cus - 1) / cus;
    if (kind == SOTS_SYNTH_4OP_SERIES) return share > 64u;
    if (kind == SOTS_SYNTH_3OP_SERIES) return share > 64u && share <= (uint32_t)kOlTileRows;
    return false;
#endif
}
bool synth_time_parallel(uint32_t kind, uint32_t p, uint32_t num_cus)
{
#ifdef SOTS_SYNTH_NO_TP
    return false;
#else
    const uint32_t share = (p + (num_cus ? num_cus : 256u) - 1) / (num_cus ? num_cus : 256u);
    switch (kind) {
    case SOTS_SYNTH_2OP: return share <= (uint32_t)SOTS_TP_2OP_MAX;
    case SOTS_SYNTH_3OP_SERIES: return share <= (uint32_t)tp_max_individuals<SOTS_SYNTH_3OP_SERIES>();
    case SOTS_SYNTH_4OP_SERIES: return share <= (uint32_t)tp_max_individuals<SOTS_SYNTH_4OP_SERIES>();
    case SOTS_SYNTH_TRIPLE_PAR: return share <= (uint32_t)tp_max_individuals<SOTS_SYNTH_TRIPLE_PAR>();
    default: return false;
    }
#endif
}

// ------------------------------------------------------------------------------------
// The voices with the arithmetic of the reference's DEVICE kernels (sots_set_synth_arithmetic, SOTS_ARITH_DEVICE_KERNELS):
// ocl_program.cl:280-443 writes the sample-rate ratio as the double expression (WAVETABLE_SIZE / 44100.0), so an increment
// is a double product rounded to float once (:308, :356, :417) and a modulated phase advances by a double multiply-add
// rounded once (:319, :364, :368, :425); the float multiply-adds are fused by the OpenCL build; the 3-op voice's second
// offset is params[4] (:363); the first phase of the 2-op and triple voices has no lower wrap (:322-323); the index is the
// bare conversion, so a phase that lands on W reads entry W (here: 0 = sin 2 pi, the table's own continuation).
// tests/test_ocl_reference.py holds the outputs of those kernels - compiled as they stand and run on this GPU - and this
// kernel reproduces them bit for bit.  A compatibility path (a lane per individual, samples through a 64 x 16 tile so that
// four lanes store one row's 64 bytes): nothing of BASELINE's configurations runs here.
// ------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void k_synth_dev(const float *__restrict__ values, const float *__restrict__ wavetable,
                                                   float *__restrict__ audio, SynthParams sp, uint32_t p_len, uint32_t n, uint32_t pitch)
{
    constexpr int D = VoiceShape<KIND>::D, U = 16, STRIDE = U + 1;
    static_assert(KIND != SOTS_SYNTH_4OP_SERIES, "the reference has no device kernel for the build-defined 4-op voice");
    __shared__ float tab[kWavetableSize + 1];
    __shared__ float tile_all[4][kWave * STRIDE];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    for (uint32_t i = tid; i < kWavetableSize; i += 256) tab[i] = wavetable[i];
    if (tid == 0) tab[kWavetableSize] = 0.0f;
    const uint32_t row = blockIdx.x * 256u + tid;
    const bool live = row < p_len;
    float ps[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const int s = KIND == SOTS_SYNTH_TRIPLE_PAR ? (i & 3) : i; // (one set of bounds for the three voices, as everywhere in this library)
        ps[i] = __builtin_fmaf(live ? values[(size_t)row * D + i] : 0.0f, sp.pmax[s] - sp.pmin[s], sp.pmin[s]);
    }
    const double cd = (double)kWavetableSize / 44100.0;
    auto at = [&](float pos) -> float {
        uint32_t i = pos > 0.0f ? (uint32_t)pos : 0u;
        return tab[i > kWavetableSize ? kWavetableSize : i];
    };
    auto advance = [&](float pos, float cur) -> float { return (float)__builtin_fma(cd, (double)cur, (double)pos); };
    auto wrap_hi = [&](float &pos) { if (pos >= kWf) pos -= kWf; };
    auto wrap_lo = [&](float &pos) { if (pos < 0.0f) pos += kWf; };
    constexpr int J = KIND == SOTS_SYNTH_TRIPLE_PAR ? 3 : 1;
    float pa[J] = {}, pb[J] = {}, pc = 0.0f, inc[J], mod[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        mod[j] = ps[4 * j + 0] * ps[4 * j + 1];
        inc[j] = (float)(cd * (double)ps[KIND == SOTS_SYNTH_3OP_SERIES ? 1 : 4 * j]);
    }
    const float m2 = KIND == SOTS_SYNTH_3OP_SERIES ? ps[2] * ps[3] : 0.0f, m3 = KIND == SOTS_SYNTH_3OP_SERIES ? ps[D - 2] * ps[D - 1] : 0.0f;
    float *const tile = tile_all[wave];
    __syncthreads();
    for (uint32_t s0 = 0; s0 < n; s0 += U) {
#pragma unroll 4
        for (int u = 0; u < U; ++u) {
            float out;
            if constexpr (KIND == SOTS_SYNTH_3OP_SERIES) {
                const float cur1 = __builtin_fmaf(at(pa[0]), mod[0], ps[3]);
                pa[0] += inc[0];
                const float cur2 = __builtin_fmaf(at(pb[0]), m2, ps[4]);
                pb[0] = advance(pb[0], cur1);
                out = at(pc) * m3;
                pc = advance(pc, cur2);
                wrap_hi(pa[0]), wrap_lo(pa[0]), wrap_hi(pb[0]), wrap_lo(pb[0]), wrap_hi(pc), wrap_lo(pc);
            } else {
                float tot[J];
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const float cur = __builtin_fmaf(at(pa[j]), mod[j], ps[4 * j + 2]);
                    tot[j] = at(pb[j]) * ps[4 * j + 3];
                    pa[j] += inc[j];
                    pb[j] = advance(pb[j], cur);
                    wrap_hi(pa[j]), wrap_hi(pb[j]), wrap_lo(pb[j]);
                }
                if constexpr (J == 3) out = (float)((double)(tot[0] + tot[1] + tot[2]) / 3.0);
                else out = tot[0];
            }
            tile[lane * STRIDE + u] = out;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) { // four lanes per row: 64 contiguous bytes
            const uint32_t r = it * 16u + lane / 4u, c = (lane & 3u) * 4u;
            const uint32_t grow = blockIdx.x * 256u + wave * kWave + r;
            const float4 v = make_float4(tile[r * STRIDE + c], tile[r * STRIDE + c + 1], tile[r * STRIDE + c + 2], tile[r * STRIDE + c + 3]);
            if (grow < p_len) *reinterpret_cast<float4 *>(audio + (size_t)grow * pitch + s0 + c) = v;
        }
        __syncthreads();
    }
}

hipError_t launch_synth_device_arith(hipStream_t st, uint32_t kind, const float *values, const float *wavetable, float *audio,
                                     const SynthParams &sp, uint32_t p, uint32_t log2n, uint32_t pitch)
{
    const uint32_t n = 1u << log2n, grid = (p + 255u) / 256u;
    switch (kind) {
    case SOTS_SYNTH_2OP: k_synth_dev<SOTS_SYNTH_2OP><<<grid, 256, 0, st>>>(values, wavetable, audio, sp, p, n, pitch); break;
    case SOTS_SYNTH_3OP_SERIES: k_synth_dev<SOTS_SYNTH_3OP_SERIES><<<grid, 256, 0, st>>>(values, wavetable, audio, sp, p, n, pitch); break;
    case SOTS_SYNTH_TRIPLE_PAR: k_synth_dev<SOTS_SYNTH_TRIPLE_PAR><<<grid, 256, 0, st>>>(values, wavetable, audio, sp, p, n, pitch); break;
    default: return hipErrorInvalidValue; // (the 4-op voice is build-defined: the reference has no kernel to agree with)
    }
    return hipGetLastError();
}

hipError_t launch_synth(hipStream_t st, uint32_t kind, const float *values, const float *wavetable,
                        float *audio, const SynthParams &sp, uint32_t p, uint32_t log2n, uint32_t pitch,
                        uint32_t num_cus, const Variation *variation, bool allow_cut)
{
    Variation var = {};
    if (variation) var = *variation;
    const uint32_t n = 1u << log2n;
    const uint32_t cus = num_cus ? num_cus : 256;
    // The 128 KiB table allows one workgroup per CU, so the workgroup is sized to the CU's share
    // of the population, up to one wavefront per SIMD; larger populations loop.  Up to two
    // wavefronts' worth per CU, a series voice is cut into two wavefronts per 64 individuals.
    const uint32_t share = (p + cus - 1) / cus;
    // a few individuals per CU: the time axis in the lanes (k_synth_tp), two wavefronts per operator
    if (allow_cut && synth_time_parallel(kind, p, num_cus)) {
        switch (kind) {
        case SOTS_SYNTH_2OP: k_synth_tp<SOTS_SYNTH_2OP><<<(p + share - 1) / share, tp_waves<SOTS_SYNTH_2OP>() * kWave, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, share, var); break;
        case SOTS_SYNTH_3OP_SERIES: k_synth_tp<SOTS_SYNTH_3OP_SERIES><<<(p + share - 1) / share, tp_waves<SOTS_SYNTH_3OP_SERIES>() * kWave, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, share, var); break;
        case SOTS_SYNTH_TRIPLE_PAR: k_synth_tp<SOTS_SYNTH_TRIPLE_PAR><<<(p + share - 1) / share, tp_waves<SOTS_SYNTH_TRIPLE_PAR>() * kWave, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, share, var); break;
        default: k_synth_tp<SOTS_SYNTH_4OP_SERIES><<<(p + share - 1) / share, tp_waves<SOTS_SYNTH_4OP_SERIES>() * kWave, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, share, var); break;
        }
        return hipGetLastError();
    }
    // The 3- and 4-operator voices from 65 individuals per CU (BASELINE configs[3]'s shard: 128): the operators in the lanes
    // (k_synth_ol; where exactly: synth_operators_in_lanes above) - every wavefront carries 16 individuals, up to fifteen wavefronts per
    // CU without hand-overs or barriers.  The 2-operator voice stays on k_synth (224 per CU: 49.3 against 51.1 us, and 256 per CU - BASELINE
    // configs[2] - leaves no LDS for the spare table entry and the progress counters); `SOTS_OL_ALL` is an experiment switch.
#ifndef SOTS_SYNTH_NO_OL
#ifdef SOTS_OL_ALL
    const bool use_ol = allow_cut && kind != SOTS_SYNTH_TRIPLE_PAR && share >= (uint32_t)SOTS_OL_MIN_SHARE;
#else
    const bool use_ol = allow_cut && synth_operators_in_lanes(kind, p, num_cus);
#endif
    if (use_ol) {
        auto launch_ol = [&](auto kind_tag) {
            constexpr int K = decltype(kind_tag)::value, OPS = VoiceShape<K>::OPS, IPW = OlShape<OPS>::IPW;
            constexpr uint32_t max_rows = ol_max_waves<OPS>() * IPW;
            // a CU's share in equal tiles of at most max_rows rows (256 per CU: two tiles of 128, not 240 + 16)
            const uint32_t tiles_per_cu = (share + max_rows - 1) / max_rows;
            uint32_t rows = ((share + tiles_per_cu - 1) / tiles_per_cu + IPW - 1) / IPW * IPW;
            rows = rows > max_rows ? max_rows : rows;
            uint32_t grid = (p + rows - 1) / rows;
            grid = grid > cus ? cus : grid; // larger populations: a workgroup takes several tiles
            k_synth_ol<K><<<grid, rows / IPW * kWave, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var);
        };
        switch (kind) {
        case SOTS_SYNTH_2OP: launch_ol(ic<SOTS_SYNTH_2OP>{}); break;
        case SOTS_SYNTH_3OP_SERIES: launch_ol(ic<SOTS_SYNTH_3OP_SERIES>{}); break;
        default: launch_ol(ic<SOTS_SYNTH_4OP_SERIES>{}); break;
        }
        return hipGetLastError();
    }
#endif
    uint32_t waves = (share + kWave - 1) / kWave;
    waves = waves < 1 ? 1 : waves > (uint32_t)kSynthWaves ? (uint32_t)kSynthWaves : waves;
    const bool cut = allow_cut && waves <= 2 && kind != SOTS_SYNTH_TRIPLE_PAR;
    const uint32_t threads = (cut ? 2 : 1) * waves * kWave, grid = grid_for(p, waves * kWave, cus);
    // variation folded in, 2-operator voice, full workgroups with one tile each: sixteen wavefronts make the individuals
    if (var.vin && kind == SOTS_SYNTH_2OP && !cut && (uint64_t)grid * waves * kWave * 2 >= p) { // one or two tiles per workgroup
        k_synth<SOTS_SYNTH_2OP, 0, true><<<grid, 4 * waves * kWave, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var);
        return hipGetLastError();
    }
    // ... and the cut kernels of a small population likewise: four threads per individual make the genes (one, two or three
    // each), then a wavefront per stage and 64 individuals stays (one tile per workgroup there)
    if (var.vin && cut && (uint64_t)grid * waves * kWave >= p) {
        const uint32_t ht = 4 * waves * kWave;
        switch (kind) {
        case SOTS_SYNTH_2OP: k_synth<SOTS_SYNTH_2OP, 1, true><<<grid, ht, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var); return hipGetLastError();
        // up to 64 individuals per CU every operator of a 3- or 4-operator chain gets a wavefront of its own (three or four
        // SIMDs busy; a stage is one operator long: 3-op N = 2048 P = 1024 105 -> 81 us per generation, 4-op N = 4096 204 -> 170)
        case SOTS_SYNTH_3OP_SERIES:
            if (waves == 1) k_synth<SOTS_SYNTH_3OP_SERIES, 1, true, 2><<<grid, ht, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var);
            else k_synth<SOTS_SYNTH_3OP_SERIES, 2, true><<<grid, ht, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var);
            return hipGetLastError();
        case SOTS_SYNTH_4OP_SERIES:
            if (waves == 1) k_synth<SOTS_SYNTH_4OP_SERIES, 1, true, 2, 3><<<grid, ht, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var);
            else k_synth<SOTS_SYNTH_4OP_SERIES, 1, true, 2, 3><<<grid, ht, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var); // (all eight wavefronts stay)
            return hipGetLastError();
        default: break;
        }
    }
#define SOTS_SYNTH_CASE(K, S)                                                                            \
    case K:                                                                                              \
        if (cut) k_synth<K, S><<<grid, threads, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var); \
        else k_synth<K, 0><<<grid, threads, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var);     \
        break;
    switch (kind) {
        SOTS_SYNTH_CASE(SOTS_SYNTH_2OP, 1)
    case SOTS_SYNTH_3OP_SERIES:
        // one group of 64 per CU: a wavefront per operator (three SIMDs); two groups: the chain cut once (four wavefronts)
        if (cut && waves == 1) k_synth<SOTS_SYNTH_3OP_SERIES, 1, false, 2><<<grid, 3 * kWave, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var);
        else if (cut) k_synth<SOTS_SYNTH_3OP_SERIES, 2><<<grid, threads, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var);
        else k_synth<SOTS_SYNTH_3OP_SERIES, 0><<<grid, threads, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var);
        break;
    case SOTS_SYNTH_4OP_SERIES:
        // up to 128 individuals per CU a wavefront per operator: four wavefronts for one group of 64 (round 2: three stages
        // {0, 1} | {2} | {3} with 16-sample blocks: 165 against 131 us at P = 1024, N = 4096), eight for two (BASELINE configs[3]'s
        // shard; `-DSOTS_SYNTH_NO_CUT3`: the single cut of round 2 there)
#ifndef SOTS_SYNTH_NO_CUT3
        if (cut) k_synth<SOTS_SYNTH_4OP_SERIES, 1, false, 2, 3><<<grid, 4 * waves * kWave, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var);
#else
        if (cut && waves == 1) k_synth<SOTS_SYNTH_4OP_SERIES, 1, false, 2, 3><<<grid, 4 * kWave, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var);
        else if (cut) k_synth<SOTS_SYNTH_4OP_SERIES, 2><<<grid, threads, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var);
#endif
        else k_synth<SOTS_SYNTH_4OP_SERIES, 0><<<grid, threads, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var);
        break;
    case SOTS_SYNTH_TRIPLE_PAR:
        // up to 64 individuals per CU a wavefront per chain (three SIMDs): 134 -> ... us at P = 1024 (profiles/r03_experiments.md)
        if (allow_cut && waves == 1) k_synth<SOTS_SYNTH_TRIPLE_PAR, -1><<<grid, 3 * kWave, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var);
        else k_synth<SOTS_SYNTH_TRIPLE_PAR, 0><<<grid, threads, 0, st>>>(values, wavetable, audio, sp, p, n, pitch, var);
        break;
    default: return hipErrorInvalidValue;
    }
#undef SOTS_SYNTH_CASE
    return hipGetLastError();
}

hipError_t launch_window(hipStream_t st, float *audio, const float *window, uint32_t p, uint32_t log2n,
                         uint32_t pitch)
{
    const size_t total4 = ((size_t)p << log2n) / 4;
    k_window<<<grid_for(total4, 256, 8192), 256, 0, st>>>(audio, window, total4, log2n - 2, pitch / 4);
    return hipGetLastError();
}

// Grid of a grid-stride kernel whose items all cost the same: exactly as many workgroups as
// are resident at once (occupancy query, cached per kernel and per context: OccCache).  A larger grid runs in rounds
// and the last, partly filled round costs as much as a full one (measured: 4096 one-wave
// workgroups on 3072 slots took 1.5x the time of 3072).
template <typename K>
static uint32_t resident_grid(K kernel, int threads, uint32_t items, uint32_t num_cus, int *cache)
{
    if (*cache == 0) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        *cache = per_cu;
    }
    const uint64_t cap = (uint64_t)(num_cus ? num_cus : 256) * (uint64_t)*cache;
    // (a grid balanced to equal row counts - 2979 workgroups of 22 rows instead of 3072 of 21 or 22 at P = 65536 - measures
    // the same: profiles/r03_experiments.md)
    return (uint32_t)(items < cap ? items : cap);
}

// N <= 1024 runs on the wavefront-per-row kernels with LDS exchanges (k_fft, k_fitness), longer rows on the
// wavefront-per-row kernels with the sub-transforms across lanes (k_fft_x, k_fitness_x)
#define SOTS_DISPATCH_WAVE(log2n, CALL)   \
    switch (log2n) {                      \
    case 9: { CALL(9); break; }           \
    case 10: { CALL(10); break; }         \
    default: return hipErrorInvalidValue; \
    }

#ifndef SOTS_X_MIN
#define SOTS_X_MIN 11
#endif
// ... and N = 256 (round 4): two complex points per lane, the same kernel (k_fft's radix-4 / radix-8 passes need four or eight)
static bool x_from(uint32_t log2n) { return log2n == 8 || (log2n >= SOTS_X_MIN && log2n <= 13); }
#define SOTS_DISPATCH_X(log2n, CALL)      \
    switch (log2n) {                      \
    case 8: { CALL(8); break; }           \
    case 10: { CALL(10); break; }         \
    case 11: { CALL(11); break; }         \
    case 12: { CALL(12); break; }         \
    case 13: { CALL(13); break; }         \
    default: return hipErrorInvalidValue; \
    }
static bool big_from(uint32_t log2n) { return log2n == 14 || log2n == 15; }
// a workgroup per row: as many workgroups as rows, at most four per CU (they loop)
template <int L, int MODE, bool WIN>
static hipError_t launch_fft_big(hipStream_t st, uint32_t p, uint32_t num_cus, const float *audio, float *spectrum, const float *target,
                                 float *fitness, const float2 *tw, const float *window, float inv_n, float inv_wf, uint32_t pitch)
{
    const uint32_t cus = num_cus ? num_cus : 256u, grid = p < 4u * cus ? p : 4u * cus;
    const size_t lds = ((size_t)(1u << L) / 2u) * sizeof(float2) + 64u;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fft_big<L, MODE, WIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    k_fft_big<L, MODE, WIN><<<grid, kBigThreads, lds, st>>>(audio, spectrum, target, fitness, tw, window, p, inv_n, inv_wf, pitch);
    return hipGetLastError();
}
template <int L>
static hipError_t launch_fitness_big(hipStream_t st, uint32_t p, uint32_t num_cus, const float *spectrum, const float *target, float *fitness,
                                     float inv_n, float inv_wf)
{
    const uint32_t cus = num_cus ? num_cus : 256u, grid = p < 4u * cus ? p : 4u * cus;
    k_fitness_big<L><<<grid, kBigThreads, 0, st>>>(spectrum, target, fitness, p, inv_n, inv_wf);
    return hipGetLastError();
}
#define SOTS_X_GRID(K, L, MODE) resident_grid(K, x_waves<L, MODE>() * kWave, (p + x_waves<L, MODE>() - 1) / x_waves<L, MODE>(), num_cus, &occ_x[L])

// N = 1024 from one row per resident wavefront (P >= 12 x CUs): one workgroup of twelve wavefronts per CU, rows dealt as
// the wavefronts ask (k_fft); 15.7 -> 14.7 us at P = 8192 already, level at 4096
static bool fft_wide(uint32_t p, uint32_t log2n, uint32_t num_cus)
{
#ifdef SOTS_FFT_NO_WIDE
    return false; // (experiments: the one-wavefront workgroups at every size)
#else
#ifndef SOTS_FFT_WIDE_ROWS
#define SOTS_FFT_WIDE_ROWS 1u
#endif
    return log2n == 10 && p >= SOTS_FFT_WIDE_ROWS * fft_wide_waves<10>() * (num_cus ? num_cus : 256u);
#endif
}
#define SOTS_WIDE_GRID(K, slot) resident_grid(K, fft_wide_waves<10>() * kWave, (p + fft_wide_waves<10>() - 1) / fft_wide_waves<10>(), num_cus, &oc->wide[slot])

hipError_t launch_fft(hipStream_t st, const float *audio, float *spectrum, const float2 *twiddle,
                      uint32_t p, uint32_t log2n, uint32_t pitch, uint32_t num_cus, OccCache *oc)
{
    int *occ = oc->fft;
    if (big_from(log2n)) {
        if (log2n == 14) return launch_fft_big<14, 0, false>(st, p, num_cus, audio, spectrum, nullptr, nullptr, twiddle, nullptr, 0.f, 0.f, pitch);
        return launch_fft_big<15, 0, false>(st, p, num_cus, audio, spectrum, nullptr, nullptr, twiddle, nullptr, 0.f, 0.f, pitch);
    }
    if (fft_wide(p, log2n, num_cus) && !x_from(log2n)) {
        constexpr int W = fft_wide_waves<10>();
        k_fft<10, 0, false, W><<<SOTS_WIDE_GRID((k_fft<10, 0, false, W>), 0), W * kWave, 0, st>>>(audio, spectrum, nullptr, nullptr, twiddle, nullptr, p, 0.f, 0.f, pitch);
        return hipGetLastError();
    }
    if (x_from(log2n)) {
        int *occ_x = oc->x_fft;
#define CALL(L) k_fft_x<L, 0, false><<<SOTS_X_GRID((k_fft_x<L, 0, false>), L, 0), x_waves<L, 0>() * kWave, 0, st>>>(audio, spectrum, nullptr, nullptr, twiddle, nullptr, p, 0.f, 0.f, pitch, nullptr)
        SOTS_DISPATCH_X(log2n, CALL)
#undef CALL
        return hipGetLastError();
    }
#define CALL(L) k_fft<L, 0, false><<<resident_grid(k_fft<L, 0, false>, kWave, p, num_cus, &occ[L]), kWave, 0, st>>>(audio, spectrum, nullptr, nullptr, twiddle, nullptr, p, 0.f, 0.f, pitch)
    SOTS_DISPATCH_WAVE(log2n, CALL)
#undef CALL
    return hipGetLastError();
}

hipError_t launch_fitness(hipStream_t st, const float *spectrum, const float *target, float *fitness,
                          uint32_t p, uint32_t log2n, float inv_n, float inv_wf, uint32_t num_cus, OccCache *oc)
{
    int *occ = oc->fitness;
    if (big_from(log2n)) {
        if (log2n == 14) return launch_fitness_big<14>(st, p, num_cus, spectrum, target, fitness, inv_n, inv_wf);
        return launch_fitness_big<15>(st, p, num_cus, spectrum, target, fitness, inv_n, inv_wf);
    }
    if (x_from(log2n)) {
        int *occ_x = oc->x_fitness;
#define CALL(L) k_fitness_x<L><<<SOTS_X_GRID(k_fitness_x<L>, L, 1), x_waves<L>() * kWave, 0, st>>>(spectrum, target, fitness, p, inv_n, inv_wf)
        SOTS_DISPATCH_X(log2n, CALL)
#undef CALL
        return hipGetLastError();
    }
#define CALL(L) k_fitness<L><<<resident_grid(k_fitness<L>, kWave, p, num_cus, &occ[L]), kWave, 0, st>>>(spectrum, target, fitness, p, inv_n, inv_wf)
    SOTS_DISPATCH_WAVE(log2n, CALL)
#undef CALL
    return hipGetLastError();
}

// the table image the fused long-row kernel copies in (bytes: x_table_bytes; rebuilt whenever the target changes)
size_t x_table_bytes(uint32_t log2n)
{
    switch (log2n) {
    case 10: return x_table_floats<10>() * sizeof(float);
    case 11: return x_table_floats<11>() * sizeof(float);
    case 12: return x_table_floats<12>() * sizeof(float);
    case 13: return x_table_floats<13>() * sizeof(float);
    default: return 0;
    }
}
hipError_t launch_x_tables(hipStream_t st, float *image, const float2 *twiddle, const float *window, const float *target, uint32_t log2n)
{
    if (!x_from(log2n)) return hipSuccess;
    hipError_t e = hipMemsetAsync(image, 0, x_table_bytes(log2n), st); // (the padding between the lanes' rows)
    if (e != hipSuccess) return e;
#define CALL(L) k_x_tables<L><<<8, 256, 0, st>>>(image, twiddle, window, target)
    SOTS_DISPATCH_X(log2n, CALL)
#undef CALL
    return hipGetLastError();
}

hipError_t launch_fft_fitness(hipStream_t st, const float *audio, const float *window, const float *target,
                              float *fitness, const float2 *twiddle, uint32_t p, uint32_t log2n, uint32_t pitch,
                              float inv_n, float inv_wf, uint32_t num_cus, OccCache *oc)
{
    int *occ_w = oc->fused_win, *occ_n = oc->fused_raw;
    if (big_from(log2n)) {
        if (log2n == 14) {
            if (window) return launch_fft_big<14, 1, true>(st, p, num_cus, audio, nullptr, target, fitness, twiddle, window, inv_n, inv_wf, pitch);
            return launch_fft_big<14, 1, false>(st, p, num_cus, audio, nullptr, target, fitness, twiddle, nullptr, inv_n, inv_wf, pitch);
        }
        if (window) return launch_fft_big<15, 1, true>(st, p, num_cus, audio, nullptr, target, fitness, twiddle, window, inv_n, inv_wf, pitch);
        return launch_fft_big<15, 1, false>(st, p, num_cus, audio, nullptr, target, fitness, twiddle, nullptr, inv_n, inv_wf, pitch);
    }
    if (x_from(log2n)) {
        if (window) {
            int *occ_x = oc->x_fused_win;
            // a small population: a wavefront per SIMD.  The threshold (fewer 16-row workgroups than CUs) was measured at
            // N = 4096 and 2048, where x_waves is 16; N = 8192 (x_waves 8) follows it unmeasured
            if ((p + x_waves<12>() - 1) / x_waves<12>() < (num_cus ? num_cus : 256u)) {
                int *occ_s = oc->x_small;
#define CALL(L) k_fft_x<L, 1, true, 4><<<resident_grid((k_fft_x<L, 1, true, 4>), 4 * kWave, (p + 3) / 4, num_cus, &occ_s[L]), 4 * kWave, 0, st>>>(audio, nullptr, target, fitness, twiddle, window, p, inv_n, inv_wf, pitch, oc->x_image)
                SOTS_DISPATCH_X(log2n, CALL)
#undef CALL
                return hipGetLastError();
            }
#define CALL(L) k_fft_x<L, 1, true><<<SOTS_X_GRID((k_fft_x<L, 1, true>), L, 1), x_waves<L>() * kWave, 0, st>>>(audio, nullptr, target, fitness, twiddle, window, p, inv_n, inv_wf, pitch, oc->x_image)
            SOTS_DISPATCH_X(log2n, CALL)
#undef CALL
        } else {
            int *occ_x = oc->x_fused_raw;
#define CALL(L) k_fft_x<L, 1, false><<<SOTS_X_GRID((k_fft_x<L, 1, false>), L, 1), x_waves<L>() * kWave, 0, st>>>(audio, nullptr, target, fitness, twiddle, nullptr, p, inv_n, inv_wf, pitch, nullptr)
            SOTS_DISPATCH_X(log2n, CALL)
#undef CALL
        }
        return hipGetLastError();
    }
    if (fft_wide(p, log2n, num_cus)) {
        constexpr int W = fft_wide_waves<10>();
        if (window) k_fft<10, 1, true, W><<<SOTS_WIDE_GRID((k_fft<10, 1, true, W>), 1), W * kWave, 0, st>>>(audio, nullptr, target, fitness, twiddle, window, p, inv_n, inv_wf, pitch);
        else k_fft<10, 1, false, W><<<SOTS_WIDE_GRID((k_fft<10, 1, false, W>), 2), W * kWave, 0, st>>>(audio, nullptr, target, fitness, twiddle, nullptr, p, inv_n, inv_wf, pitch);
        return hipGetLastError();
    }
    if (window) {
#define CALL(L) k_fft<L, 1, true><<<resident_grid(k_fft<L, 1, true>, kWave, p, num_cus, &occ_w[L]), kWave, 0, st>>>(audio, nullptr, target, fitness, twiddle, window, p, inv_n, inv_wf, pitch)
        SOTS_DISPATCH_WAVE(log2n, CALL)
#undef CALL
        return hipGetLastError();
    }
#define CALL(L) k_fft<L, 1, false><<<resident_grid(k_fft<L, 1, false>, kWave, p, num_cus, &occ_n[L]), kWave, 0, st>>>(audio, nullptr, target, fitness, twiddle, nullptr, p, inv_n, inv_wf, pitch)
    SOTS_DISPATCH_WAVE(log2n, CALL)
#undef CALL
    return hipGetLastError();
}

constexpr uint32_t kSortMaxTiles = 256;
constexpr uint32_t kSortTwoLevelFrom = 131072; // padded population from which runs of 8 tiles are merged first

// Plans: one LDS tile (n_pad <= 4096); tiles of 1024 + one rank level (up to 65536: 64 tiles);
// tiles of 1024 merged into runs of 8192 + a rank level over the runs (two levels: the rank work is
// quadratic in the number of sorted pieces, 128 tiles -> 16 runs at 131072); beyond 128 runs
// (n_pad > 1M) global bitonic steps over tiles of 4096.
constexpr uint32_t kSortMaxRuns = 128;         // 1 Mi keys; the counts of the run level take runs * n_pad * 2 bytes
static bool sort_two_level(uint32_t n_pad) { return n_pad >= kSortTwoLevelFrom && n_pad / kRankLdsKeys <= kSortMaxRuns; }

static void sort_plan(uint32_t n_pad, uint32_t &tile, uint32_t &tiles)
{
    if (n_pad <= kSortSmall) tile = n_pad; // (k_sort_small: no tiles at all)
    else if (n_pad <= 65536u || sort_two_level(n_pad)) tile = 1024u;
    else tile = kSortTile;
    tiles = n_pad / tile;
}

// tiles staged per rank workgroup: as many as fit the LDS buffer and a 16-bit count
static uint32_t rank_group(uint32_t tile, uint32_t tiles)
{
    uint32_t group = kRankLdsKeys / tile;
    if (group > kRankGroup) group = kRankGroup;
    if (group < 1) group = 1;
    while (group > tiles) group >>= 1;
    return group;
}

// scratch layout: [partial counts (u16)] then, for two levels, [merged runs (u64, n_pad keys)]
static size_t sort_partial_bytes(uint32_t n_pad)
{
    uint32_t tile, tiles;
    sort_plan(n_pad, tile, tiles);
    if (sort_two_level(n_pad)) return (size_t)(n_pad / kRankLdsKeys) * n_pad * sizeof(uint16_t);
    if (tiles <= 1 || tiles > kSortMaxTiles) return 16;
    const uint32_t group = rank_group(tile, tiles);
    return (size_t)(tiles / group) * n_pad * sizeof(uint16_t);
}

size_t sort_scratch_bytes(uint32_t p)
{
    const uint32_t n_pad = next_pow2(p < 2 ? 2 : p);
    size_t bytes = (sort_partial_bytes(n_pad) + 255) & ~(size_t)255;
    if (sort_two_level(n_pad)) bytes += (size_t)n_pad * sizeof(uint64_t);
    return bytes;
}

hipError_t launch_sort(hipStream_t st, const float *vin, const float *sin, const float *fin,
                       float *vout, float *sout, float *fout, uint64_t *keys, void *scratch, uint32_t p,
                       uint32_t d, uint32_t first_row, const SortExchange *exchange)
{
    const SortExchange ex = exchange ? *exchange : SortExchange{};
    const uint32_t n_pad = next_pow2(p < 2 ? 2 : p);
    if (n_pad <= kSortSmall) { // one launch
        switch ((n_pad + kWave - 1) / kWave) {
        case 1: k_sort_small<1><<<1, 1 * kWave, 0, st>>>(vin, sin, fin, vout, sout, fout, p, d, first_row, ex); break;
        case 2: k_sort_small<2><<<1, 2 * kWave, 0, st>>>(vin, sin, fin, vout, sout, fout, p, d, first_row, ex); break;
        case 4: k_sort_small<4><<<1, 4 * kWave, 0, st>>>(vin, sin, fin, vout, sout, fout, p, d, first_row, ex); break;
        case 8: k_sort_small<8><<<1, 8 * kWave, 0, st>>>(vin, sin, fin, vout, sout, fout, p, d, first_row, ex); break;
        default: k_sort_small<16><<<1, 16 * kWave, 0, st>>>(vin, sin, fin, vout, sout, fout, p, d, first_row, ex); break;
        }
        return hipGetLastError();
    }
    uint32_t tile, tiles;
    sort_plan(n_pad, tile, tiles);
    uint32_t threads = tile / 2;
    threads = threads < 64 ? 64 : threads > (uint32_t)kSortThreads ? (uint32_t)kSortThreads : threads;
    const bool two_level = sort_two_level(n_pad);
    const bool rank_merge = two_level || (tiles > 1 && tiles <= kSortMaxTiles);
    if (rank_merge && tile == 4 * kRunKeys)
        k_sort_tiles_1k<<<tiles, 256, 0, st>>>(fin, keys, p);
    else
        k_sort_tiles<<<tiles, threads, 0, st>>>(fin, keys, p, tile, rank_merge ? 0u : 1u);
    if (two_level) {
        uint16_t *partial = static_cast<uint16_t *>(scratch);
        uint64_t *merged = reinterpret_cast<uint64_t *>(static_cast<char *>(scratch) +
                                                        ((sort_partial_bytes(n_pad) + 255) & ~(size_t)255));
        // level 1: groups of 8 tiles -> sorted runs of 8192 keys; level 2: every key against every run
        k_sort_rank_pairs<kRankGroup, true><<<tiles, kRankThreads, 0, st>>>(keys, nullptr, n_pad, tile, merged);
        const uint32_t runs = n_pad / kRankLdsKeys;
        k_sort_rank_pairs<1><<<dim3(runs, runs), kRankThreads, 0, st>>>(merged, partial, n_pad, kRankLdsKeys);
        k_sort_rank_scatter<<<n_pad / 64, kRankThreads, 0, st>>>(merged, partial, vin, sin, fin, vout, sout, fout,
                                                               n_pad, runs, p, d, first_row, ex);
        return hipGetLastError();
    }
    if (rank_merge) {
        uint16_t *partial = static_cast<uint16_t *>(scratch);
        const uint32_t group = rank_group(tile, tiles), groups = tiles / group;
        switch (group) {
        case 8: k_sort_rank_pairs<8><<<dim3(tiles, groups), kRankThreads, 0, st>>>(keys, partial, n_pad, tile); break;
        case 4: k_sort_rank_pairs<4><<<dim3(tiles, groups), kRankThreads, 0, st>>>(keys, partial, n_pad, tile); break;
        case 2: k_sort_rank_pairs<2><<<dim3(tiles, groups), kRankThreads, 0, st>>>(keys, partial, n_pad, tile); break;
        default: k_sort_rank_pairs<1><<<dim3(tiles, groups), kRankThreads, 0, st>>>(keys, partial, n_pad, tile); break;
        }
        k_sort_rank_scatter<<<n_pad / 64, kRankThreads, 0, st>>>(keys, partial, vin, sin, fin, vout, sout, fout,
                                                               n_pad, groups, p, d, first_row, ex);
        return hipGetLastError();
    }
    for (uint32_t k = tile << 1; k <= n_pad && k != 0; k <<= 1) {
        for (uint32_t j = k >> 1; j >= tile; j >>= 1)
            k_sort_global_step<<<grid_for(n_pad / 2, 256), 256, 0, st>>>(keys, n_pad, k, j);
        k_sort_tile_merge<<<tiles, threads, 0, st>>>(keys, k, tile);
    }
    k_sort_gather<<<grid_for((uint64_t)p * (2 * d + 1), 256), 256, 0, st>>>(keys, vin, sin, fin, vout, sout,
                                                                            fout, p, d, first_row, ex);
    return hipGetLastError();
}

// keys buffer: n_pad 64-bit keys (full sort) or n_pad fitness-bit words + n_pad indices (selection),
// followed by the selection's per-tile samples
// The selection's key arrays: the population padded to a power of two and to at least kSelMinTiles tiles (the rank kernel
// holds one sample per lane and more: 16 tiles x 4 samples); a population of 2048 ... 8192 simply has tiles of nothing but
// padding keys behind its own (they sort last, stage one quantum each and move nothing)
static uint32_t sel_pad(uint32_t p)
{
    const uint32_t n_pad = next_pow2(p < 2 ? 2 : p);
    return n_pad < kSelMinTiles * kSelTile ? kSelMinTiles * kSelTile : n_pad;
}

size_t sort_keys_bytes(uint32_t p)
{
    const uint32_t n_pad = p > kSortSmall ? sel_pad(p) : next_pow2(p < 2 ? 2 : p);
    return (size_t)n_pad * sizeof(uint64_t) + (size_t)2 * kSelMaxTiles * kSelSamples * sizeof(uint32_t);
}

// The selection applies to every population that does not fit k_sort_small's one launch (P > 1024; round 3: it used to start at
// 8192, the three launches of the tile sort below that took 20-23 us where the selection's two take 15) while at most half of
// the rows are wanted (beyond that nearly everything would be staged and the full sort is the better plan).  Up to 64 tiles
// (P <= 65536) the rank kernel works on the 1024-key tiles; at 128 tiles (P <= 131072) they are first merged four by four
// (k_sel_merge4), which quarters its work; larger populations take the two-level full sort.
constexpr uint32_t kSelDirectTiles = 64, kSelMergedTiles = 128;
bool select_applies(uint32_t p, uint32_t need)
{
    const uint32_t tiles = sel_pad(p) / kSelTile;
    return p > kSortSmall && tiles <= kSelMergedTiles && need >= 1 && (uint64_t)need * 2 <= p;
}

// scratch of the merged plan: the 4096-key tiles (bits, indices) and their samples
size_t select_scratch_bytes(uint32_t p)
{
    const uint32_t n_pad = sel_pad(p);
    return (size_t)n_pad * 2 * sizeof(uint32_t) + (size_t)2 * kSelMaxTiles * kSelSamples * sizeof(uint32_t);
}

hipError_t launch_select(hipStream_t st, const float *vin, const float *sin, const float *fin, float *vout,
                         float *sout, float *fout, uint64_t *keys, void *scratch, uint32_t p, uint32_t d, uint32_t need,
                         uint32_t num_cus, const SortExchange *exchange)
{
    if (!select_applies(p, need)) return hipErrorInvalidValue;
    const SortExchange ex = exchange ? *exchange : SortExchange{};
    const uint32_t n_pad = sel_pad(p);
    const uint32_t tiles = n_pad / kSelTile;
    uint32_t *kbits = reinterpret_cast<uint32_t *>(keys), *kidx = kbits + n_pad, *samples = kidx + n_pad;
    // (split while the parts still find CUs of their own; the 128 tiles of P = 131072 stay whole)
    if (SOTS_SEL_SPLIT > 1 && tiles * SOTS_SEL_SPLIT <= (num_cus ? num_cus : 256))
        k_sel_tiles<SOTS_SEL_SPLIT><<<tiles * SOTS_SEL_SPLIT, kSelTile, 0, st>>>(fin, kbits, kidx, samples, p, tiles);
    else k_sel_tiles<1><<<tiles, kSelTile, 0, st>>>(fin, kbits, kidx, samples, p, tiles);
    // one workgroup per CU (the staging buffer takes the LDS); more only when a share of the staged
    // positions would exceed the per-workgroup key store
    uint32_t grid = num_cus ? num_cus : 256;
    const uint32_t min_grid = (tiles * kSelTile + kSelMaxOwn - 1) / kSelMaxOwn;
    if (grid < min_grid) grid = min_grid;
    // lanes per key x 8 search chains = the tiles one pass stages: 64 / 32 / 16 tiles of 1024 keys, <= 20 big tiles
    // at most 8 real tiles (P <= 8192): they are staged whole (32 KiB), the padding tiles not at all
    const uint32_t real_tiles = (p + kSelTile - 1) / kSelTile, whole = real_tiles <= 8 ? real_tiles : 0u;
#define SOTS_SEL(R, T, Q, L, KB, KI, SM, NT) \
    k_sel_rank_scatter<R, T, Q, L><<<grid, kSelThreads, 0, st>>>(KB, KI, SM, vin, sin, fin, vout, sout, fout, NT, need, p, d, ex, whole)
    if (tiles <= kSelDirectTiles) {
        switch (tiles * kSelSamples / kWave) {
        case 1: SOTS_SEL(1, kSelTile, kSelQuantum, 2, kbits, kidx, samples, tiles); break;
        case 2: SOTS_SEL(2, kSelTile, kSelQuantum, 4, kbits, kidx, samples, tiles); break;
        case 4: SOTS_SEL(4, kSelTile, kSelQuantum, 8, kbits, kidx, samples, tiles); break;
        default: return hipErrorInvalidValue;
        }
    } else {
        uint32_t *bbits = static_cast<uint32_t *>(scratch), *bidx = bbits + n_pad, *bsamples = bidx + n_pad;
        const uint32_t big = tiles / 4;
        k_sel_merge4<<<big, kSelTile, 0, st>>>(kbits, kidx, bbits, bidx, bsamples);
        switch (big * kSelBigSamples / kWave) {
        case 4: SOTS_SEL(4, kSelBigTile, kSelBigQuantum, 4, bbits, bidx, bsamples, big); break;
        default: return hipErrorInvalidValue;
        }
    }
#undef SOTS_SEL
    return hipGetLastError();
}

hipError_t launch_pack_rows(hipStream_t st, const float *values, const float *steps, const float *fitness,
                            float *rows, uint32_t first_row, uint32_t n_rows, uint32_t d)
{
    if (n_rows == 0) return hipSuccess;
    k_pack_rows<<<grid_for((uint64_t)n_rows * (2 * d + 1), 256), 256, 0, st>>>(values, steps, fitness, rows,
                                                                               first_row, n_rows, d);
    return hipGetLastError();
}

hipError_t launch_unpack_rows(hipStream_t st, float *values, float *steps, float *fitness,
                              const float *rows, uint32_t first_row, uint32_t n_rows, uint32_t d,
                              uint32_t skip_first, uint32_t skip_count)
{
    if (n_rows == 0) return hipSuccess;
    k_unpack_rows<<<grid_for((uint64_t)n_rows * (2 * d + 1), 256), 256, 0, st>>>(values, steps, fitness, rows,
                                                                                 first_row, n_rows, d, skip_first,
                                                                                 skip_count);
    return hipGetLastError();
}

} // namespace sots

#if defined(SOTS_STAMP) || defined(SOTS_STAMP_ENDS)
extern "C" int sots_debug_stamps(unsigned long long *host, size_t n)
{
    if (n > 2 * 16384) n = 2 * 16384;
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sots::g_stamps), n * sizeof(unsigned long long));
}
extern "C" int sots_debug_clear_stamps()
{
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(sots::g_stamps)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(unsigned long long) * 2 * 16384);
}
#endif
