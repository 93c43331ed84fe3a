// sots_capi.hip -- the C-ABI of include/sots_hip.h: context, buffers, stage sequencing,
// device timing.  No CPU fallback exists: every compute entry point launches gfx950 kernels
// or fails with an error code.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/sots_hip.h"
#include "sots_host_math.h"
#include "sots_kernels.h"

using namespace sots;

namespace {

struct StageClock {
    // recorded, not yet read.  Inside the fused generation loop consecutive stages SHARE the event between them (the
    // end of one is the start of the next: one event record per kernel boundary instead of two), so an event may
    // sit in two pairs; events are owned by the context's pool and go back to it once every clock has been read.
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0.0;
    uint64_t count = 0;
    std::vector<float> launches_ms; // per-launch durations since the last reset (bounded: kMaxLaunchSamples)
};
constexpr size_t kMaxLaunchSamples = 1u << 16;

thread_local std::string g_create_error;

} // namespace

struct sots_ctx {
    sots_config cfg{};
    PopDims pd{};
    MutateConsts mc{};
    SynthParams sp{};
    int device = 0;
    uint32_t num_cus = 256;
    hipStream_t own_stream = nullptr, stream = nullptr;
    uint32_t P = 0, D = 0, N = 0, log2n = 0, n_pad = 0;
    uint32_t pitch = 0; // floats between audio rows on the device (>= N)
    uint32_t rot = 0, generation = 0;
    bool target_set = false;
    // device buffers
    float *values = nullptr, *steps = nullptr, *fitness = nullptr; // [2][P][D], [2][P][D], [2][P]
    float *audio = nullptr, *spectrum = nullptr, *target = nullptr;
    float *wavetable = nullptr, *window = nullptr, *rows = nullptr;
    float2 *twiddle = nullptr;
    uint64_t *keys = nullptr;
    void *sort_scratch = nullptr;
    uint32_t rows_capacity = 0;
    OccCache occ{};
    float *x_image = nullptr; // k_fft_x's tables (N >= 2048), rebuilt with every target
    // selection state: after the fused loop's partial sort only rows [0, tail_first) of the current half
    // are in place; the unsorted half it came from is intact until the next generation starts
    uint32_t sort_mode = SOTS_SORT_LAZY_TAIL;
    uint32_t synth_arith = SOTS_ARITH_CPU_PATH;
    bool tail_pending = false;
    uint32_t tail_first = 0;
    // island exchange folded into the sort of the last generation of the next sots_execute_generations call
    SortExchange next_exchange{};
    bool next_exchange_set = false;
    hipEvent_t next_exchange_gate = nullptr; // the HOST waits for it right before that sort is enqueued
    // experiment switches; fixed in the shipped library, settable from the environment only in a
    // -DSOTS_EXPERIMENT build (tools/exp_*.sh)
    bool allow_cut = true;
    int skip_stage = 0;
    int fuse_variation = -1; // -1: by population shape, 0 / 1: forced
    // host tables
    std::vector<double> window64;
    float window_factor = 1.0f, inv_n = 0.0f, inv_wf = 1.0f;
    // timing
    bool timing = false;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t chain_tail = nullptr; // fused loop: the event the last stage ended with
    StageClock clocks[SOTS_STAGE_COUNT];
    mutable std::string err;
    char arch[32] = {0};
    char device_name[128] = {0};

    float *val(uint32_t half) const { return values + (size_t)half * P * D; }
    float *stp(uint32_t half) const { return steps + (size_t)half * P * D; }
    float *fit(uint32_t half) const { return fitness + (size_t)half * P; }
};

namespace {

int fail(const sots_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_error = buf;
    return code;
}

#define SOTS_HIP(ctx, call)                                                                       \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (void)hipGetLastError(); /* reported here: do not leave it for a later launch check */ \
            return fail(ctx, SOTS_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                      \
        }                                                                                         \
    } while (0)

#define SOTS_REQUIRE_CTX(ctx) \
    do {                      \
        if (!(ctx)) return fail(nullptr, SOTS_ERR_INVALID, "null context"); \
    } while (0)

uint32_t dims_of(uint32_t kind)
{
    switch (kind) {
    case SOTS_SYNTH_2OP: return 4;
    case SOTS_SYNTH_3OP_SERIES: return 6;
    case SOTS_SYNTH_TRIPLE_PAR: return 12;
    case SOTS_SYNTH_4OP_SERIES: return 8;
    default: return 0;
    }
}

int bind_device(const sots_ctx *ctx)
{
    SOTS_HIP(ctx, hipSetDevice(ctx->device));
    return SOTS_OK;
}

// ---- stage timing -------------------------------------------------------------------
hipEvent_t take_event(sots_ctx *c)
{
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// CHAIN (the fused generation loop, where the stages follow each other on the stream with nothing in between): the
// stage starts at the event the stage before it ended with.
struct StageScope {
    sots_ctx *ctx;
    StageClock *clock = nullptr;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    bool chain;
    StageScope(sots_ctx *c, int stage, bool chained = false) : ctx(c), chain(chained)
    {
        if (!c->timing) return;
        if (chain && c->chain_tail) {
            ev.first = c->chain_tail;
        } else {
            ev.first = take_event(c);
            if (ev.first) (void)hipEventRecord(ev.first, c->stream);
        }
        ev.second = take_event(c);
        if (!ev.first || !ev.second) { // (a lost pair costs a sample, nothing else)
            if (ev.first && ev.first != c->chain_tail) c->event_pool.push_back(ev.first);
            if (ev.second) c->event_pool.push_back(ev.second);
            return;
        }
        clock = &c->clocks[stage];
    }
    ~StageScope()
    {
        if (!clock) return;
        (void)hipEventRecord(ev.second, ctx->stream);
        clock->pending.push_back(ev);
        if (chain) ctx->chain_tail = ev.second;
    }
};

// reads every recorded pair of every stage and returns the events to the pool
int drain_clocks(sots_ctx *ctx)
{
    std::vector<hipEvent_t> used;
    int rc = SOTS_OK;
    for (auto &ck : ctx->clocks) {
        for (auto &ev : ck.pending) {
            used.push_back(ev.first);
            used.push_back(ev.second);
            if (rc) continue;
            float ms = 0.0f;
            if (hipEventSynchronize(ev.second) != hipSuccess || hipEventElapsedTime(&ms, ev.first, ev.second) != hipSuccess) {
                rc = fail(ctx, SOTS_ERR_HIP, "stage timing: %s", hipGetErrorString(hipGetLastError()));
                continue;
            }
            ck.total_ms += ms;
            ck.count += 1;
            if (ck.launches_ms.size() < kMaxLaunchSamples) ck.launches_ms.push_back(ms);
        }
        ck.pending.clear();
    }
    std::sort(used.begin(), used.end());
    used.erase(std::unique(used.begin(), used.end()), used.end());
    ctx->event_pool.insert(ctx->event_pool.end(), used.begin(), used.end());
    ctx->chain_tail = nullptr;
    return rc;
}

// keep the number of live events bounded on long runs
int maybe_drain(sots_ctx *ctx)
{
    if (!ctx->timing) return SOTS_OK;
    for (auto &ck : ctx->clocks)
        if (ck.pending.size() >= 4096) return drain_clocks(ctx);
    return SOTS_OK;
}

void free_ctx(sots_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    {
        std::vector<hipEvent_t> all = ctx->event_pool;
        for (auto &ck : ctx->clocks)
            for (auto &ev : ck.pending) all.push_back(ev.first), all.push_back(ev.second);
        std::sort(all.begin(), all.end());
        all.erase(std::unique(all.begin(), all.end()), all.end());
        for (hipEvent_t e : all) (void)hipEventDestroy(e);
    }
    void *bufs[] = {ctx->values, ctx->steps, ctx->fitness, ctx->audio, ctx->spectrum, ctx->target,
                    ctx->wavetable, ctx->window, ctx->rows, ctx->twiddle, ctx->keys, ctx->sort_scratch, ctx->x_image};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    (void)hipGetLastError(); // nothing sticky survives a context (the launchers read hipGetLastError after each launch)
    delete ctx;
}

int ensure_rows(sots_ctx *ctx, uint32_t n_rows)
{
    if (n_rows <= ctx->rows_capacity) return SOTS_OK;
    if (ctx->rows) SOTS_HIP(ctx, hipFree(ctx->rows));
    ctx->rows = nullptr;
    ctx->rows_capacity = 0;
    SOTS_HIP(ctx, hipMalloc((void **)&ctx->rows, (size_t)n_rows * (2 * ctx->D + 1) * sizeof(float)));
    ctx->rows_capacity = n_rows;
    return SOTS_OK;
}

// Rows of the sorted half that the next recombination reads (recombine_source, sots_kernels.hip):
// whole blocks of parents, floor(numParents / block) of them, at least one.  Immigrants go to the
// tail of THESE rows: with numParents not a multiple of the block, rows between the last whole
// block and numParents are never read and immigrants written there would be dead.
uint32_t breeding_rows(const sots_ctx *ctx)
{
    const uint32_t block = ctx->pd.block;
    uint32_t npb = ctx->cfg.num_parents / block;
    if (npb == 0) npb = 1;
    return npb * block;
}

// rows the selection must deliver in order: what recombination reads, and never fewer than the parents
uint32_t selected_rows(const sots_ctx *ctx)
{
    const uint32_t b = breeding_rows(ctx);
    return b > ctx->cfg.num_parents ? b : ctx->cfg.num_parents;
}

// Produces the rows the selection left out: the full sort of the (still intact) unsorted half, writing
// only rows >= tail_first of the current half (the rows in front are in place, immigrants included).
int complete_tail(sots_ctx *ctx)
{
    if (!ctx->tail_pending) return SOTS_OK;
    if (ctx->sort_mode == SOTS_SORT_TOP_ONLY) return SOTS_OK; // the caller asked for the selected rows only
    SOTS_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t dst = ctx->rot, src = ctx->rot ^ 1u, first = ctx->tail_first;
    if (first == 0) { // sots_stage_select without its sots_stage_rotate yet: the selected rows are in the other half
        dst = ctx->rot ^ 1u;
        src = ctx->rot;
        first = selected_rows(ctx);
    }
    {
        StageScope t(ctx, SOTS_STAGE_SORT_TAIL);
        SOTS_HIP(ctx, launch_sort(ctx->stream, ctx->val(src), ctx->stp(src), ctx->fit(src), ctx->val(dst), ctx->stp(dst),
                                  ctx->fit(dst), ctx->keys, ctx->sort_scratch, ctx->P, ctx->D, first));
    }
    ctx->tail_pending = false;
    return SOTS_OK;
}

// For every call that is about to WRITE population rows (a stage, write_population): the rows the selection left
// out are produced first (SOTS_SORT_LAZY_TAIL), and whatever the mode the pending state ends here - the half it would
// be completed from, or the rows it would fill, are about to change.  In SOTS_SORT_TOP_ONLY nothing is produced
// (rows >= S stay unspecified); pure reads keep the state in that mode, so switching to SOTS_SORT_LAZY_TAIL while the
// unsorted half is still intact completes it.
int settle_tail(sots_ctx *ctx)
{
    if (int rc = complete_tail(ctx)) return rc;
    ctx->tail_pending = false;
    ctx->tail_first = 0;
    return SOTS_OK;
}

int require_target(sots_ctx *ctx)
{
    if (!ctx->target_set)
        return fail(ctx, SOTS_ERR_STATE, "no target: call sots_set_target_audio or sots_set_target_spectrum first");
    return SOTS_OK;
}

} // namespace

// =======================================================================================
extern "C" {

int sots_create(const sots_config *cfg, sots_ctx **out)
{
    if (!cfg || !out) return fail(nullptr, SOTS_ERR_INVALID, "sots_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(sots_config))
        return fail(nullptr, SOTS_ERR_INVALID, "sots_config.struct_size %u != %zu", cfg->struct_size,
                    sizeof(sots_config));
    const uint32_t d = dims_of(cfg->synth_kind);
    if (d == 0) return fail(nullptr, SOTS_ERR_INVALID, "unknown synth_kind %u", cfg->synth_kind);
    if (cfg->num_dimensions != d)
        return fail(nullptr, SOTS_ERR_INVALID, "synth_kind %u needs numDimensions %u, got %u", cfg->synth_kind,
                    d, cfg->num_dimensions);
    if (cfg->audio_length_log2 < 8 || cfg->audio_length_log2 > 15)
        return fail(nullptr, SOTS_ERR_INVALID, "audioLengthLog2 %u outside 8..15", cfg->audio_length_log2);
    const uint64_t p64 = (uint64_t)cfg->num_parents + cfg->num_offspring;
    if (cfg->num_parents == 0 || p64 < 2 || p64 > (1ull << 26))
        return fail(nullptr, SOTS_ERR_INVALID, "population %llu (parents %u) not supported",
                    (unsigned long long)p64, cfg->num_parents);
    if (cfg->workgroup_size == 0 || p64 % cfg->workgroup_size != 0)
        return fail(nullptr, SOTS_ERR_INVALID,
                    "populationLength %llu must be a multiple of workgroupSize %u (the recombination block)",
                    (unsigned long long)p64, cfg->workgroup_size);

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, SOTS_ERR_NO_DEVICE, "no HIP device (%s)", hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, SOTS_ERR_NO_DEVICE, "device %d not in 0..%d", cfg->device, ndev - 1);

    sots_ctx *ctx = new sots_ctx();
    ctx->cfg = *cfg;
    ctx->device = cfg->device;
#define CREATE_HIP(call)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            int rc_ = fail(nullptr, SOTS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
            free_ctx(ctx);                                                                        \
            return rc_;                                                                           \
        }                                                                                         \
    } while (0)
    CREATE_HIP(hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    CREATE_HIP(hipGetDeviceProperties(&prop, ctx->device));
    snprintf(ctx->arch, sizeof ctx->arch, "%s", prop.gcnArchName);
    snprintf(ctx->device_name, sizeof ctx->device_name, "%s", prop.name);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        int rc = fail(nullptr, SOTS_ERR_NO_DEVICE, "device %d is %s; libsots_hip carries gfx950 code only",
                      ctx->device, prop.gcnArchName);
        free_ctx(ctx);
        return rc;
    }
    ctx->num_cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256;
    CREATE_HIP(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;

    ctx->P = (uint32_t)p64;
    ctx->D = d;
    ctx->log2n = cfg->audio_length_log2;
    ctx->N = 1u << ctx->log2n;
    ctx->n_pad = next_pow2(ctx->P < 2 ? 2 : ctx->P);
    {
        // audio rows are padded off the power-of-two stride (see sots_kernels.h)
        uint32_t pad = 32;
#ifdef SOTS_EXPERIMENT
        if (const char *e = getenv("SOTS_AUDIO_PAD")) pad = (uint32_t)strtoul(e, nullptr, 10) & ~3u; // floats
        if (const char *e = getenv("SOTS_SYNTH_CUT")) ctx->allow_cut = atoi(e) != 0;                 // 0: never cut the chain
        if (const char *e = getenv("SOTS_FUSE_VARIATION")) ctx->fuse_variation = atoi(e) != 0 ? 1 : 0;
        if (const char *e = getenv("SOTS_SKIP_STAGE")) ctx->skip_stage = atoi(e); // 1: no synthesis, 2: no spectral kernel (timing ablations)
#endif
        ctx->pitch = ctx->N + pad;
    }
    ctx->pd = make_pop_dims(ctx->P, ctx->D, cfg->num_parents, cfg->workgroup_size, cfg->gid_base, (uint32_t)cfg->seed,
                            (uint32_t)(cfg->seed >> 32));
    // Evolutionary_Strategy.hpp:611-627
    const float mpi = (float)3.14159265358979323846;
    ctx->mc.alpha = 1.4f;
    ctx->mc.one_over_alpha = 1.f / ctx->mc.alpha;
    ctx->mc.root_two_over_pi = sqrtf(2.f / (float)mpi);
    ctx->mc.beta_scale = 1.f / (float)ctx->D;
    const float beta = sqrtf(ctx->mc.beta_scale);
    ctx->mc.pow_alpha_beta = powf(ctx->mc.alpha, beta);
    ctx->mc.pow_inv_alpha_beta = powf(ctx->mc.one_over_alpha, beta);
    memcpy(ctx->sp.pmin, cfg->param_min, sizeof ctx->sp.pmin);
    memcpy(ctx->sp.pmax, cfg->param_max, sizeof ctx->sp.pmax);

    const size_t pd_bytes = (size_t)2 * ctx->P * ctx->D * sizeof(float);
    const size_t audio_bytes = (size_t)ctx->P * ctx->pitch * sizeof(float);
    const size_t spec_bytes = (size_t)ctx->P * (ctx->N + 8) * sizeof(float);
    CREATE_HIP(hipMalloc((void **)&ctx->values, pd_bytes));
    CREATE_HIP(hipMalloc((void **)&ctx->steps, pd_bytes));
    CREATE_HIP(hipMalloc((void **)&ctx->fitness, (size_t)2 * ctx->P * sizeof(float)));
    CREATE_HIP(hipMalloc((void **)&ctx->audio, audio_bytes));
    CREATE_HIP(hipMalloc((void **)&ctx->spectrum, spec_bytes));
    CREATE_HIP(hipMalloc((void **)&ctx->target, (size_t)(ctx->N / 2) * sizeof(float)));
    CREATE_HIP(hipMalloc((void **)&ctx->wavetable, (size_t)SOTS_WAVETABLE_SIZE * sizeof(float)));
    CREATE_HIP(hipMalloc((void **)&ctx->window, (size_t)ctx->N * sizeof(float)));
    CREATE_HIP(hipMalloc((void **)&ctx->twiddle, (size_t)ctx->N * sizeof(float2)));
    CREATE_HIP(hipMalloc((void **)&ctx->keys, sort_keys_bytes(ctx->P)));
    if (ctx->log2n >= 11 && x_table_bytes(ctx->log2n)) CREATE_HIP(hipMalloc((void **)&ctx->x_image, x_table_bytes(ctx->log2n)));
    {
        const size_t a = sort_scratch_bytes(ctx->P), b = select_scratch_bytes(ctx->P);
        CREATE_HIP(hipMalloc(&ctx->sort_scratch, a > b ? a : b));
    }
    CREATE_HIP(hipMemsetAsync(ctx->values, 0, pd_bytes, ctx->stream));
    CREATE_HIP(hipMemsetAsync(ctx->steps, 0, pd_bytes, ctx->stream));
    CREATE_HIP(hipMemsetAsync(ctx->fitness, 0, (size_t)2 * ctx->P * sizeof(float), ctx->stream));
    CREATE_HIP(hipMemsetAsync(ctx->audio, 0, audio_bytes, ctx->stream));
    CREATE_HIP(hipMemsetAsync(ctx->spectrum, 0, spec_bytes, ctx->stream));
    CREATE_HIP(hipMemsetAsync(ctx->target, 0, (size_t)(ctx->N / 2) * sizeof(float), ctx->stream));

    // host tables the reference also builds on the CPU and uploads (...OpenCL.hpp:315-317)
    const std::vector<float> table = make_wavetable();
    ctx->window64 = make_window(ctx->N, &ctx->window_factor);
    std::vector<float> window32(ctx->N);
    for (uint32_t i = 0; i < ctx->N; ++i) window32[i] = (float)ctx->window64[i];
    const std::vector<float> tw = make_twiddles(ctx->N);
    ctx->inv_n = 1.0f / (float)ctx->N;          // fftOneOverSize
    ctx->inv_wf = 1.f / ctx->window_factor;     // fftOneOverWindowFactor
    CREATE_HIP(hipMemcpyAsync(ctx->wavetable, table.data(), table.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    CREATE_HIP(hipMemcpyAsync(ctx->window, window32.data(), window32.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    CREATE_HIP(hipMemcpyAsync(ctx->twiddle, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    CREATE_HIP(hipStreamSynchronize(ctx->stream));
#undef CREATE_HIP
    *out = ctx;
    return SOTS_OK;
}

void sots_destroy(sots_ctx *ctx) { free_ctx(ctx); }

const char *sots_last_error(const sots_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int sots_set_stream(sots_ctx *ctx, void *hip_stream)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = bind_device(ctx)) return rc;
    SOTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return SOTS_OK;
}

int sots_synchronize(sots_ctx *ctx)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = bind_device(ctx)) return rc;
    SOTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SOTS_OK;
}

// ---- target ---------------------------------------------------------------------------
int sots_set_target_spectrum(sots_ctx *ctx, const float *magnitudes, uint32_t num_bins)
{
    SOTS_REQUIRE_CTX(ctx);
    if (!magnitudes || num_bins != ctx->N / 2)
        return fail(ctx, SOTS_ERR_SIZE, "target spectrum needs %u bins, got %u", ctx->N / 2, num_bins);
    if (int rc = bind_device(ctx)) return rc;
    SOTS_HIP(ctx, hipMemcpyAsync(ctx->target, magnitudes, (size_t)num_bins * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (ctx->x_image) { // long rows: the fused spectral kernel's tables, in the layout it reads them in
        SOTS_HIP(ctx, launch_x_tables(ctx->stream, ctx->x_image, ctx->twiddle, ctx->window, ctx->target, ctx->log2n));
        ctx->occ.x_image = ctx->x_image;
    }
    SOTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->target_set = true;
    return SOTS_OK;
}

int sots_set_target_audio(sots_ctx *ctx, const float *audio, uint32_t num_samples)
{
    SOTS_REQUIRE_CTX(ctx);
    if (!audio || num_samples < ctx->N)
        return fail(ctx, SOTS_ERR_SIZE, "target audio needs %u samples, got %u", ctx->N, num_samples);
    const std::vector<float> mag = target_spectrum(audio, ctx->N, ctx->window64, ctx->window_factor);
    return sots_set_target_spectrum(ctx, mag.data(), ctx->N / 2);
}

// ---- population -------------------------------------------------------------------------
int sots_init_population(sots_ctx *ctx, uint32_t chunk_index)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = bind_device(ctx)) return rc;
    ctx->rot = 0; // initPopulationCL, ...OpenCL.hpp:371
    ctx->generation = 0;
    // a tail the last run's selection left pending belongs to the OLD population: dropped, never completed into the new one
    ctx->tail_pending = false;
    ctx->tail_first = 0;
    ctx->next_exchange_set = false;
    {
        StageScope t(ctx, SOTS_STAGE_INIT);
        SOTS_HIP(ctx, launch_init_population(ctx->stream, ctx->val(0), ctx->stp(0), ctx->fit(0), ctx->pd, chunk_index));
    }
    return maybe_drain(ctx);
}

static int copy_population(sots_ctx *ctx, uint32_t half, bool to_device, void *values, size_t values_bytes,
                           void *steps, size_t steps_bytes, void *fitness, size_t fitness_bytes)
{
    const size_t pd_bytes = (size_t)ctx->P * ctx->D * sizeof(float), f_bytes = (size_t)ctx->P * sizeof(float);
    if ((values && values_bytes != pd_bytes) || (steps && steps_bytes != pd_bytes) ||
        (fitness && fitness_bytes != f_bytes))
        return fail(ctx, SOTS_ERR_SIZE, "population byte counts must be %zu (values, steps) and %zu (fitness)",
                    pd_bytes, f_bytes);
    if (int rc = bind_device(ctx)) return rc;
    const hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
    auto cp = [&](void *host, void *dev, size_t bytes) {
        return to_device ? hipMemcpyAsync(dev, host, bytes, kind, ctx->stream)
                         : hipMemcpyAsync(host, dev, bytes, kind, ctx->stream);
    };
    if (values) SOTS_HIP(ctx, cp(values, ctx->val(half), pd_bytes));
    if (steps) SOTS_HIP(ctx, cp(steps, ctx->stp(half), pd_bytes));
    if (fitness) SOTS_HIP(ctx, cp(fitness, ctx->fit(half), f_bytes));
    SOTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SOTS_OK;
}

int sots_write_population(sots_ctx *ctx, const float *values, size_t values_bytes, const float *steps,
                          size_t steps_bytes, const float *fitness, size_t fitness_bytes)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = settle_tail(ctx)) return rc;
    return copy_population(ctx, ctx->rot, true, (void *)values, values_bytes, (void *)steps, steps_bytes,
                           (void *)fitness, fitness_bytes);
}

int sots_read_population(sots_ctx *ctx, float *values, size_t values_bytes, float *steps, size_t steps_bytes,
                         float *fitness, size_t fitness_bytes)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = complete_tail(ctx)) return rc;
    return copy_population(ctx, ctx->rot, false, values, values_bytes, steps, steps_bytes, fitness, fitness_bytes);
}

int sots_read_population_other(sots_ctx *ctx, float *values, size_t values_bytes, float *steps,
                               size_t steps_bytes, float *fitness, size_t fitness_bytes)
{
    SOTS_REQUIRE_CTX(ctx);
    return copy_population(ctx, ctx->rot ^ 1u, false, values, values_bytes, steps, steps_bytes, fitness, fitness_bytes);
}

// ---- synthesiser buffers ---------------------------------------------------------------
int sots_write_synth(sots_ctx *ctx, const float *audio, size_t audio_bytes, const float *spectrum,
                     size_t spectrum_bytes)
{
    SOTS_REQUIRE_CTX(ctx);
    const size_t a_bytes = (size_t)ctx->P * ctx->N * sizeof(float), s_bytes = (size_t)ctx->P * (ctx->N + 8) * sizeof(float);
    if ((audio && audio_bytes != a_bytes) || (spectrum && spectrum_bytes != s_bytes))
        return fail(ctx, SOTS_ERR_SIZE, "synth byte counts must be %zu (audio) and %zu (spectrum)", a_bytes, s_bytes);
    if (int rc = bind_device(ctx)) return rc;
    if (audio)
        SOTS_HIP(ctx, hipMemcpy2DAsync(ctx->audio, (size_t)ctx->pitch * sizeof(float), audio, (size_t)ctx->N * sizeof(float),
                                       (size_t)ctx->N * sizeof(float), ctx->P, hipMemcpyHostToDevice, ctx->stream));
    if (spectrum) SOTS_HIP(ctx, hipMemcpyAsync(ctx->spectrum, spectrum, s_bytes, hipMemcpyHostToDevice, ctx->stream));
    SOTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SOTS_OK;
}

int sots_read_synth(sots_ctx *ctx, float *audio, size_t audio_bytes, float *spectrum, size_t spectrum_bytes,
                    float *target, size_t target_bytes)
{
    SOTS_REQUIRE_CTX(ctx);
    const size_t a_bytes = (size_t)ctx->P * ctx->N * sizeof(float), s_bytes = (size_t)ctx->P * (ctx->N + 8) * sizeof(float);
    const size_t t_bytes = (size_t)(ctx->N / 2) * sizeof(float);
    if ((audio && audio_bytes != a_bytes) || (spectrum && spectrum_bytes != s_bytes) || (target && target_bytes != t_bytes))
        return fail(ctx, SOTS_ERR_SIZE, "synth byte counts must be %zu (audio), %zu (spectrum), %zu (target)",
                    a_bytes, s_bytes, t_bytes);
    if (int rc = bind_device(ctx)) return rc;
    if (audio)
        SOTS_HIP(ctx, hipMemcpy2DAsync(audio, (size_t)ctx->N * sizeof(float), ctx->audio, (size_t)ctx->pitch * sizeof(float),
                                       (size_t)ctx->N * sizeof(float), ctx->P, hipMemcpyDeviceToHost, ctx->stream));
    if (spectrum) SOTS_HIP(ctx, hipMemcpyAsync(spectrum, ctx->spectrum, s_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (target) SOTS_HIP(ctx, hipMemcpyAsync(target, ctx->target, t_bytes, hipMemcpyDeviceToHost, ctx->stream));
    SOTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SOTS_OK;
}

// ---- stages -------------------------------------------------------------------------------
int sots_stage_recombine(sots_ctx *ctx)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = settle_tail(ctx)) return rc;
    if (int rc = bind_device(ctx)) return rc;
    const uint32_t src = ctx->rot, dst = ctx->rot ^ 1u;
    {
        StageScope t(ctx, SOTS_STAGE_RECOMBINE);
        SOTS_HIP(ctx, launch_recombine(ctx->stream, ctx->val(src), ctx->stp(src), ctx->val(dst), ctx->stp(dst), ctx->pd));
    }
    ctx->rot = dst; // out-of-place recombination: the recombined population is the current one
    return maybe_drain(ctx);
}

int sots_stage_mutate(sots_ctx *ctx)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = settle_tail(ctx)) return rc;
    if (int rc = bind_device(ctx)) return rc;
    {
        StageScope t(ctx, SOTS_STAGE_MUTATE);
        SOTS_HIP(ctx, launch_mutate(ctx->stream, ctx->val(ctx->rot), ctx->stp(ctx->rot), ctx->pd, ctx->mc, ctx->generation));
    }
    return maybe_drain(ctx);
}

int sots_stage_synthesise(sots_ctx *ctx)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = settle_tail(ctx)) return rc;
    if (int rc = bind_device(ctx)) return rc;
    {
        StageScope t(ctx, SOTS_STAGE_SYNTHESISE);
        if (ctx->synth_arith == SOTS_ARITH_DEVICE_KERNELS)
            SOTS_HIP(ctx, launch_synth_device_arith(ctx->stream, ctx->cfg.synth_kind, ctx->val(ctx->rot), ctx->wavetable, ctx->audio,
                                                    ctx->sp, ctx->P, ctx->log2n, ctx->pitch));
        else
        SOTS_HIP(ctx, launch_synth(ctx->stream, ctx->cfg.synth_kind, ctx->val(ctx->rot), ctx->wavetable,
                                   ctx->audio, ctx->sp, ctx->P, ctx->log2n, ctx->pitch, ctx->num_cus, nullptr, ctx->allow_cut));
    }
    return maybe_drain(ctx);
}

int sots_stage_window(sots_ctx *ctx)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = settle_tail(ctx)) return rc;
    if (int rc = bind_device(ctx)) return rc;
    {
        StageScope t(ctx, SOTS_STAGE_WINDOW);
        SOTS_HIP(ctx, launch_window(ctx->stream, ctx->audio, ctx->window, ctx->P, ctx->log2n, ctx->pitch));
    }
    return maybe_drain(ctx);
}

int sots_stage_fft(sots_ctx *ctx)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = settle_tail(ctx)) return rc;
    if (int rc = bind_device(ctx)) return rc;
    {
        StageScope t(ctx, SOTS_STAGE_FFT);
        SOTS_HIP(ctx, launch_fft(ctx->stream, ctx->audio, ctx->spectrum, ctx->twiddle, ctx->P, ctx->log2n, ctx->pitch, ctx->num_cus, &ctx->occ));
    }
    return maybe_drain(ctx);
}

int sots_stage_fitness(sots_ctx *ctx)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = settle_tail(ctx)) return rc;
    if (int rc = require_target(ctx)) return rc;
    if (int rc = bind_device(ctx)) return rc;
    {
        StageScope t(ctx, SOTS_STAGE_FITNESS);
        SOTS_HIP(ctx, launch_fitness(ctx->stream, ctx->spectrum, ctx->target, ctx->fit(ctx->rot), ctx->P, ctx->log2n,
                                     ctx->inv_n, ctx->inv_wf, ctx->num_cus, &ctx->occ));
    }
    return maybe_drain(ctx);
}

int sots_stage_sort(sots_ctx *ctx)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = settle_tail(ctx)) return rc;
    if (int rc = bind_device(ctx)) return rc;
    const uint32_t src = ctx->rot, dst = ctx->rot ^ 1u;
    {
        StageScope t(ctx, SOTS_STAGE_SORT);
        SOTS_HIP(ctx, launch_sort(ctx->stream, ctx->val(src), ctx->stp(src), ctx->fit(src), ctx->val(dst), ctx->stp(dst),
                                  ctx->fit(dst), ctx->keys, ctx->sort_scratch, ctx->P, ctx->D));
    }
    return maybe_drain(ctx);
}

int sots_stage_select(sots_ctx *ctx)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = settle_tail(ctx)) return rc;
    if (int rc = bind_device(ctx)) return rc;
    const uint32_t src = ctx->rot, dst = ctx->rot ^ 1u, need = selected_rows(ctx);
    if (ctx->sort_mode == SOTS_SORT_FULL || !select_applies(ctx->P, need)) return sots_stage_sort(ctx);
    {
        StageScope t(ctx, SOTS_STAGE_SORT);
        SOTS_HIP(ctx, launch_select(ctx->stream, ctx->val(src), ctx->stp(src), ctx->fit(src), ctx->val(dst), ctx->stp(dst),
                                    ctx->fit(dst), ctx->keys, ctx->sort_scratch, ctx->P, ctx->D, need, ctx->num_cus));
    }
    ctx->tail_pending = true; // completed from the current (unsorted) half once sots_stage_rotate has flipped
    ctx->tail_first = 0;      // marks "selected into the other half, not rotated yet"
    return maybe_drain(ctx);
}

int sots_stage_rotate(sots_ctx *ctx)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = bind_device(ctx)) return rc;
    // rotationIndex_ flip, ...OpenCL.hpp:486; a kernel argument here, so no transfer
    StageScope t(ctx, SOTS_STAGE_ROTATE);
    if (ctx->tail_pending && ctx->tail_first == 0) { // sots_stage_select just ran: its rows are in the OTHER half
        ctx->tail_first = selected_rows(ctx);
    } else if (ctx->tail_pending) {
        if (int rc = settle_tail(ctx)) return rc; // (the flip below would swap the halves under a pending state)
    }
    ctx->rot ^= 1u;
    ctx->generation += 1;
    return SOTS_OK;
}

int sots_execute_generation(sots_ctx *ctx)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = require_target(ctx)) return rc;
    int rc;
    if ((rc = sots_stage_recombine(ctx))) return rc;
    if ((rc = sots_stage_mutate(ctx))) return rc;
    if ((rc = sots_stage_synthesise(ctx))) return rc;
    if ((rc = sots_stage_window(ctx))) return rc;
    if ((rc = sots_stage_fft(ctx))) return rc;
    if ((rc = sots_stage_fitness(ctx))) return rc;
    if ((rc = sots_stage_sort(ctx))) return rc;
    return sots_stage_rotate(ctx);
}

int sots_execute_generations(sots_ctx *ctx, uint32_t n)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = require_target(ctx)) return rc;
    if (int rc = bind_device(ctx)) return rc;
    if (ctx->tail_pending && ctx->tail_first == 0)
        return fail(ctx, SOTS_ERR_STATE, "sots_stage_select must be followed by sots_stage_rotate");
    const uint32_t need = selected_rows(ctx);
    const bool select = ctx->sort_mode != SOTS_SORT_FULL && select_applies(ctx->P, need);
    // an exchange folded into the last generation's sort (sots_fuse_exchange_next_sort) is used once
    const SortExchange exchange = ctx->next_exchange;
    const bool with_exchange = ctx->next_exchange_set && n > 0;
    hipEvent_t gate = with_exchange ? ctx->next_exchange_gate : nullptr;
    if (n > 0) ctx->next_exchange_set = false, ctx->next_exchange_gate = nullptr;
    if (with_exchange && select && exchange.sink && exchange.sink_rows > need)
        return fail(ctx, SOTS_ERR_INVALID, "fused exchange: %u elite rows asked for, sortPopulation places %u per generation here", exchange.sink_rows, need);
    for (uint32_t g = 0; g < n; ++g) {
        // the variation below overwrites the unsorted half a pending tail would be completed from; nobody has
        // asked for those rows, so they are dropped
        ctx->tail_pending = false;
        ctx->chain_tail = nullptr; // (the caller may have put other work on the stream since the last generation)
        uint32_t src = ctx->rot, dst = ctx->rot ^ 1u;
        // Large populations of 4-gene individuals make their individuals inside the synthesis kernel
        // (one launch less, 151 vs 157 us per generation at P = 65536); with few wavefronts per CU or
        // more genes the serial per-lane variation costs more than the launch it saves (measured).
        const bool device_arith = ctx->synth_arith == SOTS_ARITH_DEVICE_KERNELS; // (compatibility kernel: makes no individuals)
        const bool fuse_variation = device_arith ? false : ctx->fuse_variation >= 0 ? ctx->fuse_variation == 1
                                                             : (ctx->pd.d <= 4 && ctx->P >= 192u * (ctx->num_cus ? ctx->num_cus : 256u)) ||
                                                                   // a few individuals per CU: k_synth_tp makes them, a thread per gene
                                                                   (ctx->allow_cut && synth_time_parallel(ctx->cfg.synth_kind, ctx->P, ctx->num_cus)) ||
                                                                   // small populations of the 2- and 3-operator voices: the cut kernel's helper wavefronts (launch_synth);
                                                                   // with 8 genes it is a draw (204 = 204 us at P = 1024, 373 against 368 at 32768)
                                                                   (ctx->pd.d <= 6 && ctx->cfg.synth_kind != SOTS_SYNTH_TRIPLE_PAR && ctx->allow_cut && ctx->P <= 128u * (ctx->num_cus ? ctx->num_cus : 256u)) ||
                                                                   // 3- and 4-operator voices from 65 individuals per CU: k_synth_ol (round 4) makes its workgroup's
                                                                   // individuals first, a thread per gene, whatever the number of genes
                                                                   (ctx->allow_cut && synth_operators_in_lanes(ctx->cfg.synth_kind, ctx->P, ctx->num_cus));
        if (!fuse_variation) {
            StageScope t(ctx, SOTS_STAGE_FUSED_VARIATION, true);
            SOTS_HIP(ctx, launch_recombine_mutate(ctx->stream, ctx->val(src), ctx->stp(src), ctx->val(dst), ctx->stp(dst),
                                                  ctx->pd, ctx->mc, ctx->generation));
        }
        ctx->rot = dst;
        if (ctx->skip_stage != 1) {
            StageScope t(ctx, SOTS_STAGE_FUSED_SYNTH, true);
            // raw synthesis (the window is applied by the FFT kernel as it loads the row); by default the
            // kernel also makes its individuals: recombination + mutation from the sorted half
            sots::Variation var = {ctx->val(src), ctx->stp(src), ctx->val(dst), ctx->stp(dst), ctx->pd, ctx->mc, ctx->generation};
            if (device_arith)
                SOTS_HIP(ctx, launch_synth_device_arith(ctx->stream, ctx->cfg.synth_kind, ctx->val(ctx->rot), ctx->wavetable, ctx->audio,
                                                        ctx->sp, ctx->P, ctx->log2n, ctx->pitch));
            else
            SOTS_HIP(ctx, launch_synth(ctx->stream, ctx->cfg.synth_kind, ctx->val(ctx->rot), ctx->wavetable,
                                       ctx->audio, ctx->sp, ctx->P, ctx->log2n, ctx->pitch, ctx->num_cus,
                                       fuse_variation ? &var : nullptr, ctx->allow_cut));
        }
        if (ctx->skip_stage != 2) {
            StageScope t(ctx, SOTS_STAGE_FUSED_SPECTRAL, true);
            SOTS_HIP(ctx, launch_fft_fitness(ctx->stream, ctx->audio, ctx->window, ctx->target, ctx->fit(ctx->rot), ctx->twiddle, ctx->P,
                                             ctx->log2n, ctx->pitch, ctx->inv_n, ctx->inv_wf, ctx->num_cus, &ctx->occ));
        }
        src = ctx->rot;
        dst = ctx->rot ^ 1u;
        const SortExchange *ex = with_exchange && g + 1 == n ? &exchange : nullptr;
        // The rows this sort takes were gathered on ANOTHER stream.  The host waits for that collective here, with this
        // generation's variation, synthesis and spectral kernels already on the stream (the GPU stays busy, and the
        // collective - started a generation ago - is normally long done), instead of making the stream wait: on this
        // runtime a cross-stream event wait costs the waiting stream ~18 us even when the event is complete
        // (tools/ubench/cross_stream.hip).  A kernel launched after the host has seen the event sees the rows.
        // In front of the sort's StageScope: a host wait is not sort-kernel time.
        if (ex && gate) SOTS_HIP(ctx, hipEventSynchronize(gate));
        {
            StageScope t(ctx, SOTS_STAGE_SORT, true);
            if (select) {
                // the rows recombination reads, in order; the rest of the order is produced on demand
                SOTS_HIP(ctx, launch_select(ctx->stream, ctx->val(src), ctx->stp(src), ctx->fit(src), ctx->val(dst),
                                            ctx->stp(dst), ctx->fit(dst), ctx->keys, ctx->sort_scratch, ctx->P, ctx->D, need, ctx->num_cus, ex));
            } else {
                SOTS_HIP(ctx, launch_sort(ctx->stream, ctx->val(src), ctx->stp(src), ctx->fit(src), ctx->val(dst), ctx->stp(dst),
                                          ctx->fit(dst), ctx->keys, ctx->sort_scratch, ctx->P, ctx->D, 0, ex));
            }
        }
        ctx->rot = dst;
        ctx->tail_pending = select;
        ctx->tail_first = need;
        ctx->generation += 1;
        if (int rc = maybe_drain(ctx)) return rc;
    }
    return SOTS_OK;
}

int sots_set_sort_mode(sots_ctx *ctx, uint32_t mode)
{
    SOTS_REQUIRE_CTX(ctx);
    if (mode > SOTS_SORT_TOP_ONLY) return fail(ctx, SOTS_ERR_INVALID, "unknown sort mode %u", mode);
    // the new mode decides what happens to a pending tail: leaving SOTS_SORT_TOP_ONLY completes it (the unsorted half
    // is intact as long as the state is pending, settle_tail), entering it keeps the rows unspecified
    ctx->sort_mode = mode;
    return complete_tail(ctx);
}

int sots_set_synth_arithmetic(sots_ctx *ctx, uint32_t arith)
{
    SOTS_REQUIRE_CTX(ctx);
    if (arith > SOTS_ARITH_DEVICE_KERNELS) return fail(ctx, SOTS_ERR_INVALID, "unknown synthesis arithmetic %u", arith);
    if (arith == SOTS_ARITH_DEVICE_KERNELS && ctx->cfg.synth_kind == SOTS_SYNTH_4OP_SERIES)
        return fail(ctx, SOTS_ERR_INVALID, "the reference has no device kernel for the build-defined 4-op voice");
    ctx->synth_arith = arith;
    return SOTS_OK;
}

int sots_get_generation(const sots_ctx *ctx, uint32_t *generation)
{
    SOTS_REQUIRE_CTX(ctx);
    if (!generation) return fail(ctx, SOTS_ERR_INVALID, "null generation pointer");
    *generation = ctx->generation;
    return SOTS_OK;
}

int sots_set_generation(sots_ctx *ctx, uint32_t generation)
{
    SOTS_REQUIRE_CTX(ctx);
    ctx->generation = generation;
    return SOTS_OK;
}

// ---- timing ----------------------------------------------------------------------------------
int sots_timing_enable(sots_ctx *ctx, int enabled)
{
    SOTS_REQUIRE_CTX(ctx);
    ctx->timing = enabled != 0;
    return SOTS_OK;
}

int sots_timing_reset(sots_ctx *ctx)
{
    SOTS_REQUIRE_CTX(ctx);
    if (int rc = bind_device(ctx)) return rc;
    if (int rc = drain_clocks(ctx)) return rc;
    for (auto &ck : ctx->clocks) {
        ck.total_ms = 0.0;
        ck.count = 0;
        ck.launches_ms.clear();
    }
    return SOTS_OK;
}

int sots_stage_time_ms(sots_ctx *ctx, int stage, double *total_ms, uint64_t *count)
{
    SOTS_REQUIRE_CTX(ctx);
    if (stage < 0 || stage >= SOTS_STAGE_COUNT) return fail(ctx, SOTS_ERR_INVALID, "stage %d out of range", stage);
    if (int rc = bind_device(ctx)) return rc;
    if (int rc = drain_clocks(ctx)) return rc;
    if (total_ms) *total_ms = ctx->clocks[stage].total_ms;
    if (count) *count = ctx->clocks[stage].count;
    return SOTS_OK;
}

int sots_stage_launch_times_ms(sots_ctx *ctx, int stage, float *out_ms, uint64_t capacity, uint64_t *written)
{
    SOTS_REQUIRE_CTX(ctx);
    if (stage < 0 || stage >= SOTS_STAGE_COUNT) return fail(ctx, SOTS_ERR_INVALID, "stage %d out of range", stage);
    if (!written || (capacity && !out_ms)) return fail(ctx, SOTS_ERR_INVALID, "stage_launch_times: null argument");
    if (int rc = bind_device(ctx)) return rc;
    if (int rc = drain_clocks(ctx)) return rc;
    const std::vector<float> &v = ctx->clocks[stage].launches_ms;
    const uint64_t n = v.size() < capacity ? v.size() : capacity;
    for (uint64_t i = 0; i < n; ++i) out_ms[i] = v[i];
    *written = n;
    return SOTS_OK;
}

// ---- island exchange ------------------------------------------------------------------------
int sots_pack_elites_device(sots_ctx *ctx, void *device_rows, uint32_t n_rows)
{
    SOTS_REQUIRE_CTX(ctx);
    if (!device_rows || n_rows > ctx->P) return fail(ctx, SOTS_ERR_INVALID, "pack_elites: bad rows/n_rows %u", n_rows);
    if (ctx->tail_pending && n_rows > ctx->tail_first) {
        if (ctx->sort_mode == SOTS_SORT_TOP_ONLY)
            return fail(ctx, SOTS_ERR_STATE, "pack_elites: %u rows asked for, SOTS_SORT_TOP_ONLY placed %u", n_rows, ctx->tail_first);
        if (int rc = complete_tail(ctx)) return rc;
    }
    if (int rc = bind_device(ctx)) return rc;
    SOTS_HIP(ctx, launch_pack_rows(ctx->stream, ctx->val(ctx->rot), ctx->stp(ctx->rot), ctx->fit(ctx->rot),
                                   (float *)device_rows, 0, n_rows, ctx->D));
    return SOTS_OK;
}

int sots_inject_immigrants_device(sots_ctx *ctx, const void *device_rows, uint32_t n_rows)
{
    SOTS_REQUIRE_CTX(ctx);
    if (!device_rows || n_rows > breeding_rows(ctx))
        return fail(ctx, SOTS_ERR_INVALID, "inject_immigrants: %u rows do not fit the %u parent rows recombination reads", n_rows, breeding_rows(ctx));
    if (int rc = bind_device(ctx)) return rc;
    SOTS_HIP(ctx, launch_unpack_rows(ctx->stream, ctx->val(ctx->rot), ctx->stp(ctx->rot), ctx->fit(ctx->rot),
                                     (const float *)device_rows, breeding_rows(ctx) - n_rows, n_rows, ctx->D, 0, 0));
    return SOTS_OK;
}

int sots_inject_gathered_device(sots_ctx *ctx, const void *gathered_rows, uint32_t world, uint32_t rank, uint32_t elites)
{
    SOTS_REQUIRE_CTX(ctx);
    if (!gathered_rows || world == 0 || rank >= world)
        return fail(ctx, SOTS_ERR_INVALID, "inject_gathered: bad arguments (world %u, rank %u)", world, rank);
    const uint64_t n_rows = (uint64_t)(world - 1) * elites;
    if (n_rows > breeding_rows(ctx))
        return fail(ctx, SOTS_ERR_INVALID, "inject_gathered: %llu immigrant rows do not fit the %u parent rows recombination reads",
                    (unsigned long long)n_rows, breeding_rows(ctx));
    if (int rc = bind_device(ctx)) return rc;
    SOTS_HIP(ctx, launch_unpack_rows(ctx->stream, ctx->val(ctx->rot), ctx->stp(ctx->rot), ctx->fit(ctx->rot),
                                     (const float *)gathered_rows, breeding_rows(ctx) - (uint32_t)n_rows, (uint32_t)n_rows,
                                     ctx->D, rank * elites, elites));
    return SOTS_OK;
}

int sots_fuse_exchange_next_sort(sots_ctx *ctx, void *elite_rows, uint32_t n_elite_rows, const void *gathered_rows,
                                 uint32_t world, uint32_t rank, uint32_t elites, void *host_gate_event)
{
    SOTS_REQUIRE_CTX(ctx);
    ctx->next_exchange_set = false;
    ctx->next_exchange_gate = nullptr;
    SortExchange ex{};
    if (elite_rows) {
        if (n_elite_rows == 0 || n_elite_rows > ctx->P) return fail(ctx, SOTS_ERR_INVALID, "fused exchange: bad elite row count %u", n_elite_rows);
        ex.sink = (float *)elite_rows;
        ex.sink_rows = n_elite_rows;
    }
    if (gathered_rows) {
        if (world == 0 || rank >= world) return fail(ctx, SOTS_ERR_INVALID, "fused exchange: bad arguments (world %u, rank %u)", world, rank);
        const uint64_t n_rows = (uint64_t)(world - 1) * elites;
        if (n_rows > breeding_rows(ctx))
            return fail(ctx, SOTS_ERR_INVALID, "fused exchange: %llu immigrant rows do not fit the %u parent rows recombination reads",
                        (unsigned long long)n_rows, breeding_rows(ctx));
        if (n_rows) {
            ex.imm = (const float *)gathered_rows;
            ex.imm_first = breeding_rows(ctx) - (uint32_t)n_rows;
            ex.imm_rows = (uint32_t)n_rows;
            ex.skip_first = rank * elites;
            ex.skip_count = elites;
        }
    }
    if (!ex.sink && !ex.imm) return SOTS_OK; // nothing to fold in
    ctx->next_exchange = ex;
    ctx->next_exchange_set = true;
    ctx->next_exchange_gate = (hipEvent_t)host_gate_event;
    return SOTS_OK;
}

int sots_pack_elites_host(sots_ctx *ctx, float *rows, uint32_t n_rows)
{
    SOTS_REQUIRE_CTX(ctx);
    if (!rows) return fail(ctx, SOTS_ERR_INVALID, "pack_elites: null rows");
    if (int rc = bind_device(ctx)) return rc;
    if (int rc = ensure_rows(ctx, n_rows)) return rc;
    if (int rc = sots_pack_elites_device(ctx, ctx->rows, n_rows)) return rc;
    SOTS_HIP(ctx, hipMemcpyAsync(rows, ctx->rows, (size_t)n_rows * (2 * ctx->D + 1) * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SOTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SOTS_OK;
}

int sots_inject_immigrants_host(sots_ctx *ctx, const float *rows, uint32_t n_rows)
{
    SOTS_REQUIRE_CTX(ctx);
    if (!rows) return fail(ctx, SOTS_ERR_INVALID, "inject_immigrants: null rows");
    if (int rc = bind_device(ctx)) return rc;
    if (int rc = ensure_rows(ctx, n_rows)) return rc;
    SOTS_HIP(ctx, hipMemcpyAsync(ctx->rows, rows, (size_t)n_rows * (2 * ctx->D + 1) * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = sots_inject_immigrants_device(ctx, ctx->rows, n_rows)) return rc;
    SOTS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SOTS_OK;
}

int sots_get_info(const sots_ctx *ctx, sots_info *info)
{
    SOTS_REQUIRE_CTX(ctx);
    if (!info) return fail(ctx, SOTS_ERR_INVALID, "null info");
    memset(info, 0, sizeof *info);
    info->population_length = ctx->P;
    info->num_dimensions = ctx->D;
    info->audio_length = ctx->N;
    info->spectrum_row_floats = ctx->N + 8;
    info->rotation_index = ctx->rot;
    info->generation = ctx->generation;
    info->compute_units = ctx->num_cus;
    snprintf(info->device_name, sizeof info->device_name, "%s", ctx->device_name);
    snprintf(info->arch, sizeof info->arch, "%s", ctx->arch);
    return SOTS_OK;
}

} // extern "C"
