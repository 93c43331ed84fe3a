// sots_group.hip -- the island model inside the library: one process, one island (sots_ctx) per
// listed device, elites exchanged by an RCCL all-gather over xGMI ("one process,
// ncclCommInitAll over the devices, ncclAllGather of each island's best rows").  This is what the
// C++ drop-in (Evolutionary_Strategy_HIP_Arguments::numDevices) and sots_match's
// type.HIP.{numDevices,numElites,migrationInterval} drive; bench.py keeps one PROCESS per GPU over
// torch.distributed, which is the same exchange seen from the other side.
//
//   * every island has its own PERSISTENT host thread (island 0: the caller's): a generation is 4-6 launches, and
//     eight islands launched from one thread would be host-bound (about 4 us per launch against ~150 us of GPU
//     work per generation).  The threads are made once per group, spin for a job for a few hundred microseconds
//     and then sleep on a condition variable, so sots_group_execute_generations(1) in a loop (the C++ class's
//     executeGeneration) does not pay a thread creation per call;
//   * exchange, every `interval` generations: pack (island stream) -> all-gather -> inject the other
//     islands' rows into the tail of the breeding rows (sots_inject_gathered_device).  Buffers alternate with
//     the exchange count, so the only cross-island ordering the HOST has to provide is "every `packed` event of this
//     exchange is recorded before anybody waits for it": one host barrier per exchange with the copy backend, none
//     with RCCL (every thread issues its own device's ncclAllGather; the devices meet inside the collective);
//   * schedules as in island.py: same generation, or overlapped (the all-gather started after
//     generation g runs on a side stream underneath generation g+1 and is injected after g+1's sort);
//   * backends: RCCL (distinct devices; librccl is opened with dlopen when the first multi-device
//     group is made, so single-GPU users never load it), or device-to-device copies ordered by HIP
//     events when islands share a device (rehearsals and the one-GPU tests).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sots_hip.h"
#include "sots_host_sync.h"

using sots_host::JobGate;
using sots_host::SpinBarrier;

namespace {

thread_local std::string g_group_create_error;

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

// opened once per process, on first use
Rccl *load_rccl(std::string &err)
{
    static std::mutex mu;
    static Rccl lib;
    static bool tried = false;
    std::lock_guard<std::mutex> lock(mu);
    if (tried) {
        if (!lib.handle) err = "librccl could not be loaded earlier in this process";
        return lib.handle ? &lib : nullptr;
    }
    tried = true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        lib.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (lib.handle) break;
    }
    if (!lib.handle) {
        err = std::string("dlopen(librccl): ") + dlerror();
        return nullptr;
    }
#define SOTS_SYM(field, sym)                                                   \
    lib.field = reinterpret_cast<decltype(lib.field)>(dlsym(lib.handle, sym)); \
    if (!lib.field) {                                                          \
        err = std::string("librccl lacks ") + sym;                             \
        dlclose(lib.handle);                                                   \
        lib.handle = nullptr;                                                  \
        return nullptr;                                                        \
    }
    SOTS_SYM(CommInitAll, "ncclCommInitAll")
    SOTS_SYM(CommDestroy, "ncclCommDestroy")
    SOTS_SYM(AllGather, "ncclAllGather")
    SOTS_SYM(GetErrorString, "ncclGetErrorString")
#undef SOTS_SYM
    return &lib;
}

struct Island {
    sots_ctx *ctx = nullptr;
    int device = 0;
    hipStream_t stream = nullptr; // the island's compute stream (owned here, handed to the context)
    hipStream_t side = nullptr;   // collective stream of the overlapped schedule
    float *mine[2] = {nullptr, nullptr};     // packed elites, double-buffered
    float *gathered[2] = {nullptr, nullptr}; // all islands' elites in island order
    hipEvent_t packed[2] = {nullptr, nullptr};   // mine[b] is complete
    hipEvent_t arrived[2] = {nullptr, nullptr};  // gathered[b] is complete
    ncclComm_t comm = nullptr;
};

} // namespace

struct sots_group {
    std::vector<Island> islands;
    sots_config island_cfg{};
    uint32_t elites = 0, interval = 1, flags = 0, width = 0;
    bool use_rccl = false;
    bool fused = true; // pack and inject ride in the sort kernels (sots_fuse_exchange_next_sort) instead of two launches each
    // overlapped schedule: an island's HOST THREAD waits for the side stream's gather (sots_fuse_exchange_next_sort,
    // host_gate_event) instead of its compute stream: a cross-stream hipStreamWaitEvent costs the waiting stream ~18 us
    // per generation on this runtime, even for an event that completed long ago (tools/ubench/cross_stream.hip)
    bool host_gated = false;
    Rccl *rccl = nullptr;
    uint32_t generation = 0; // generations run since the last init_population
    uint32_t exchanges = 0;  // exchanges done since then: exchange x uses buffers x & 1
    int pending = -1;        // buffer index of the all-gather in flight (overlapped schedule), -1 = none
    std::mutex err_mu;       // island threads may fail at the same time
    std::string err;
    // persistent island threads (islands 1..n-1; island 0 runs on the caller's thread)
    std::vector<std::thread> workers;
    JobGate gate;
    SpinBarrier *barrier = nullptr;
    std::vector<int> rcs;    // per island, result of the current job
};

namespace {

int gfail(sots_group *g, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) {
        std::lock_guard<std::mutex> lock(g->err_mu);
        g->err = buf;
    } else {
        g_group_create_error = buf;
    }
    return code;
}

void destroy_group(sots_group *g)
{
    if (!g) return;
    if (!g->workers.empty()) {
        g->gate.post(0, -1, true);
        for (auto &t : g->workers) t.join();
        g->workers.clear();
    }
    delete g->barrier;
    g->barrier = nullptr;
    for (Island &is : g->islands) {
        if (!is.ctx && !is.stream) continue; // never got as far as its device
        (void)hipSetDevice(is.device);
        if (is.stream) (void)hipStreamSynchronize(is.stream);
        if (is.side) (void)hipStreamSynchronize(is.side);
        if (is.comm && g->rccl) (void)g->rccl->CommDestroy(is.comm);
        if (is.ctx) sots_destroy(is.ctx);
        for (int b = 0; b < 2; ++b) {
            if (is.mine[b]) (void)hipFree(is.mine[b]);
            if (is.gathered[b]) (void)hipFree(is.gathered[b]);
            if (is.packed[b]) (void)hipEventDestroy(is.packed[b]);
            if (is.arrived[b]) (void)hipEventDestroy(is.arrived[b]);
        }
        if (is.side) (void)hipStreamDestroy(is.side);
        if (is.stream) (void)hipStreamDestroy(is.stream);
    }
    (void)hipGetLastError(); // a failed creation must not leave its HIP error behind for the next launch to find
    delete g;
}

#define GROUP_HIP(g, call)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) return gfail(g, SOTS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// Before mine[b] is written again: the previous use of this buffer pair (two exchanges ago) must be over on the
// device.  With RCCL this island's own collective read it (arrived[b] of this island); with the copy backend every
// island copied from it (everybody's arrived[b]).  Those events were recorded before the barrier of the exchange in
// between, so this thread sees them; they are an exchange old, so the wait gates nothing.
int exchange_reuse_wait(sots_group *g, uint32_t i, int b)
{
    Island &is = g->islands[i];
    GROUP_HIP(g, hipSetDevice(is.device));
    if (g->use_rccl) {
        GROUP_HIP(g, hipStreamWaitEvent(is.stream, is.arrived[b], 0));
    } else {
        for (Island &reader : g->islands) GROUP_HIP(g, hipStreamWaitEvent(is.stream, reader.arrived[b], 0));
    }
    return SOTS_OK;
}

// Step 1 of an exchange, island i: its best rows into mine[b] on its own stream.  Fused form, BEFORE the generations
// are enqueued: the last generation's sort writes them (and, in the overlapped schedule, takes the rows gathered at
// the previous exchange straight from gathered[pending]: step 3 without its launch).
int exchange_prepare_fused(sots_group *g, uint32_t i, int b, int pending)
{
    Island &is = g->islands[i];
    // Who may still read mine[b] (filled at exchange x - 2)?  With host gating nobody: this thread has waited for its own
    // gather of exchange x - 2 at the gate of exchange x - 1, and (copy backend) every other thread for theirs before
    // the host barrier of exchange x - 1, which this thread has passed.  Otherwise the stream waits for the events.
    hipEvent_t gate = nullptr;
    if (g->host_gated) {
        if (pending >= 0) gate = is.arrived[pending];
    } else {
        if (int rc = exchange_reuse_wait(g, i, b)) return rc;
        if (pending >= 0) GROUP_HIP(g, hipStreamWaitEvent(is.stream, is.arrived[pending], 0));
    }
    if (int rc = sots_fuse_exchange_next_sort(is.ctx, is.mine[b], g->elites, pending >= 0 ? is.gathered[pending] : nullptr,
                                              (uint32_t)g->islands.size(), i, g->elites, gate))
        return gfail(g, rc, "island %u: %s", i, sots_last_error(is.ctx));
    return SOTS_OK;
}

// ... and the launch it replaces
int exchange_pack(sots_group *g, uint32_t i, int b)
{
    Island &is = g->islands[i];
    if (int rc = exchange_reuse_wait(g, i, b)) return rc;
    if (int rc = sots_pack_elites_device(is.ctx, is.mine[b], g->elites)) return gfail(g, rc, "island %u: %s", i, sots_last_error(is.ctx));
    return SOTS_OK;
}

int exchange_packed(sots_group *g, uint32_t i, int b)
{
    Island &is = g->islands[i];
    GROUP_HIP(g, hipSetDevice(is.device));
    GROUP_HIP(g, hipEventRecord(is.packed[b], is.stream));
    return SOTS_OK;
}

// Step 2, island i's share: gathered[b] of island i <- every island's mine[b].  `on_side`: on the side stream.
// RCCL: this device's rank of the all-gather, called from this island's thread (one thread per device: no group
// call; the ranks meet on the devices).  Copy backend: one copy per island, each behind the owner's `packed` event -
// which the owner's thread recorded before the host barrier the caller has just passed.
int exchange_gather(sots_group *g, uint32_t i, int b, bool on_side)
{
    Island &is = g->islands[i];
    GROUP_HIP(g, hipSetDevice(is.device));
    hipStream_t st = on_side ? is.side : is.stream;
    const size_t count = (size_t)g->elites * g->width, bytes = count * sizeof(float);
    if (g->use_rccl) {
        if (on_side) GROUP_HIP(g, hipStreamWaitEvent(st, is.packed[b], 0));
        const ncclResult_t r = g->rccl->AllGather(is.mine[b], is.gathered[b], count, ncclFloat, is.comm, st);
        if (r != ncclSuccess) return gfail(g, SOTS_ERR_HIP, "island %u: ncclAllGather: %s", i, g->rccl->GetErrorString(r));
    } else {
        const uint32_t n = (uint32_t)g->islands.size();
        for (uint32_t j = 0; j < n; ++j) {
            Island &from = g->islands[j];
            if (j != i || on_side) GROUP_HIP(g, hipStreamWaitEvent(st, from.packed[b], 0));
            char *dst = reinterpret_cast<char *>(is.gathered[b]) + (size_t)j * bytes;
            if (from.device == is.device) GROUP_HIP(g, hipMemcpyAsync(dst, from.mine[b], bytes, hipMemcpyDeviceToDevice, st));
            else GROUP_HIP(g, hipMemcpyPeerAsync(dst, is.device, from.mine[b], from.device, bytes, st));
        }
    }
    GROUP_HIP(g, hipEventRecord(is.arrived[b], st));
    return SOTS_OK;
}

// Step 3, island i: the other islands' rows into the tail of its breeding rows.
int exchange_inject(sots_group *g, uint32_t i, int b, bool wait_event)
{
    Island &is = g->islands[i];
    GROUP_HIP(g, hipSetDevice(is.device));
    if (wait_event) GROUP_HIP(g, hipStreamWaitEvent(is.stream, is.arrived[b], 0)); // (the same stream otherwise)
    if (int rc = sots_inject_gathered_device(is.ctx, is.gathered[b], (uint32_t)g->islands.size(), i, g->elites))
        return gfail(g, rc, "island %u: %s", i, sots_last_error(is.ctx));
    return SOTS_OK;
}

// n generations of island i with the exchanges that fall due.  Every island's thread runs this with the same
// arguments and therefore takes the same branches; between two exchanges a thread only talks to its own device.
// The only wait for the GPU is the host gate of the overlapped schedule: before the sort that takes the rows of the
// previous exchange is enqueued, the thread waits for that exchange's `arrived` event (inside sots_execute_generations,
// with the generation's other kernels already on the stream).  A failed island keeps taking part in barriers and
// collectives (the others must not hang): it stops running generations, its context is disarmed, and what it sends are
// the rows of its last good exchange or - before any - rows of +inf fitness (mine[] is created that way), which no
// recombination prefers.  After an error the group's populations are unspecified; the call returns the error.
int run_island(sots_group *g, uint32_t i, uint32_t n, int pending, int *pending_out)
{
    const uint32_t islands = (uint32_t)g->islands.size();
    const bool exchange = g->elites > 0 && (islands > 1 || (g->flags & SOTS_GROUP_FORCE_RCCL));
    const bool overlap = (g->flags & SOTS_GROUP_OVERLAP) != 0;
    int rc = SOTS_OK;
    if (!exchange) {
        rc = sots_execute_generations(g->islands[i].ctx, n);
        if (rc) rc = gfail(g, rc, "island %u: %s", i, sots_last_error(g->islands[i].ctx));
        *pending_out = -1;
        return rc;
    }
    const bool host_barrier = !g->use_rccl && islands > 1;
    auto keep = [&](int r) { if (!rc && r) rc = r; };
    uint32_t x = g->exchanges; // exchanges done so far (the same value in every thread)
    // generations up to the next exchange go out in one call
    uint32_t k = 0;
    while (k < n) {
        uint32_t run = g->interval - (g->generation + k) % g->interval; // generations until one is due
        if (run > n - k) run = n - k;
        const bool due = (g->generation + k + run) % g->interval == 0;
        const int b = (int)(x & 1u);
        if (due && g->fused && !rc) keep(exchange_prepare_fused(g, i, b, overlap ? pending : -1));
        if (!rc) {
            const int r = sots_execute_generations(g->islands[i].ctx, run);
            if (r) keep(gfail(g, r, "island %u: %s", i, sots_last_error(g->islands[i].ctx)));
        }
        // a failed island's context must not keep an armed exchange for some later, unrelated call
        if (rc) (void)sots_fuse_exchange_next_sort(g->islands[i].ctx, nullptr, 0, nullptr, 0, 0, 0, nullptr);
        k += run;
        if (!due) break; // (the job ended before the next exchange)
        ++x;
        if (!g->fused) {
            // rows gathered at the previous exchange arrive now (overlapped schedule), then this exchange's are packed
            if (overlap && !rc && pending >= 0) keep(exchange_inject(g, i, pending, true));
            keep(exchange_pack(g, i, b));
        }
        keep(exchange_packed(g, i, b));
        if (host_barrier) g->barrier->arrive_and_wait(); // every island has recorded `packed`
        keep(exchange_gather(g, i, b, overlap));          // overlapped: on the side stream, underneath the next generations
        if (overlap) pending = b;
        else if (!rc) keep(exchange_inject(g, i, b, false));
    }
    *pending_out = overlap ? pending : -1;
    return rc;
}

void worker_main(sots_group *g, uint32_t i)
{
    uint64_t seen = 0;
    (void)hipSetDevice(g->islands[i].device);
    for (;;) {
        g->gate.wait_job(seen);
        seen = g->gate.seq.load(std::memory_order_acquire);
        if (g->gate.quit) return;
        int unused = -1;
        g->rcs[i] = run_island(g, i, g->gate.n, g->gate.pending_in, &unused);
        g->gate.done.fetch_add(1, std::memory_order_release);
    }
}

} // namespace

// =======================================================================================
extern "C" {

int sots_group_create(const sots_config *island_cfg, const int32_t *devices, uint32_t num_devices, uint32_t num_elites,
                      uint32_t migration_interval, uint32_t flags, sots_group **out)
{
    if (!island_cfg || !out || (num_devices && !devices)) return gfail(nullptr, SOTS_ERR_INVALID, "sots_group_create: null argument");
    *out = nullptr;
    if (num_devices == 0 || num_devices > SOTS_MAX_GROUP_DEVICES)
        return gfail(nullptr, SOTS_ERR_INVALID, "numDevices %u outside 1..%u", num_devices, SOTS_MAX_GROUP_DEVICES);
    if (island_cfg->struct_size != sizeof(sots_config)) return gfail(nullptr, SOTS_ERR_INVALID, "sots_config.struct_size mismatch");
    if (migration_interval == 0) migration_interval = 1;
    const uint64_t p64 = (uint64_t)island_cfg->num_parents + island_cfg->num_offspring;
    if (p64 * num_devices + island_cfg->gid_base > 0xFFFFFFFFull)
        return gfail(nullptr, SOTS_ERR_INVALID, "global individual ids exceed 32 bits");

    sots_group *g = new sots_group();
    g->island_cfg = *island_cfg;
    g->elites = num_devices > 1 ? num_elites : 0; // one island has nobody to exchange with
    g->interval = migration_interval;
    g->flags = flags;
    g->width = 2 * island_cfg->num_dimensions + 1;
    g->islands.resize(num_devices);

    bool distinct = true;
    for (uint32_t i = 0; i < num_devices; ++i)
        for (uint32_t j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
    g->use_rccl = (num_devices > 1 && distinct) || (flags & SOTS_GROUP_FORCE_RCCL);
    if ((flags & SOTS_GROUP_FORCE_RCCL) && !distinct) {
        destroy_group(g);
        return gfail(nullptr, SOTS_ERR_INVALID, "RCCL needs one distinct device per island");
    }
    if ((flags & SOTS_GROUP_FORCE_RCCL) && num_devices == 1) g->elites = num_elites; // exercises the collective on one rank

#define CREATE_FAIL(code, ...)                      \
    do {                                            \
        int rc_ = gfail(nullptr, code, __VA_ARGS__); \
        destroy_group(g);                           \
        return rc_;                                 \
    } while (0)
#define CREATE_HIP(call)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) CREATE_FAIL(SOTS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

    const unsigned ev_flags = hipEventDisableTiming; // (hipEventReleaseToDevice changes nothing measurable: profiles/r03_experiments.md)
    for (uint32_t i = 0; i < num_devices; ++i) {
        Island &is = g->islands[i];
        is.device = devices[i];
        sots_config cfg = *island_cfg;
        cfg.device = devices[i];
        cfg.gid_base = island_cfg->gid_base + (uint32_t)(p64 * i); // global ids: an island's stream does not depend on the group size
        if (int rc = sots_create(&cfg, &is.ctx)) CREATE_FAIL(rc, "island %u: %s", i, sots_last_error(nullptr));
        CREATE_HIP(hipSetDevice(is.device));
        CREATE_HIP(hipStreamCreateWithFlags(&is.stream, hipStreamNonBlocking));
        CREATE_HIP(hipStreamCreateWithFlags(&is.side, hipStreamNonBlocking));
        if (int rc = sots_set_stream(is.ctx, is.stream)) CREATE_FAIL(rc, "island %u: %s", i, sots_last_error(is.ctx));
        // immigrants overwrite the tail of the rows recombination reads: whole parent blocks (sots_inject_gathered_device)
        const uint64_t immigrants = (uint64_t)(num_devices - 1) * g->elites;
        const uint32_t block = island_cfg->workgroup_size ? island_cfg->workgroup_size : 1u;
        const uint32_t npb = island_cfg->num_parents / block ? island_cfg->num_parents / block : 1u;
        if (g->elites > p64 || immigrants > (uint64_t)npb * block)
            CREATE_FAIL(SOTS_ERR_INVALID, "%u elites from each of %u other islands do not fit the %u parent rows recombination reads",
                        g->elites, num_devices - 1, npb * block);
        const size_t mine_bytes = (size_t)(g->elites ? g->elites : 1) * g->width * sizeof(float);
        // rows nobody has written yet read as [+inf fitness, +inf ...]: an island that fails before its first exchange
        // still joins the gather, and such rows sort behind everything (0x7F800000 in every float)
        std::vector<uint32_t> inf_rows(mine_bytes / sizeof(uint32_t) * num_devices, 0x7F800000u);
        for (int b = 0; b < 2; ++b) {
            CREATE_HIP(hipMalloc((void **)&is.mine[b], mine_bytes));
            CREATE_HIP(hipMalloc((void **)&is.gathered[b], mine_bytes * num_devices));
            CREATE_HIP(hipMemcpy(is.mine[b], inf_rows.data(), mine_bytes, hipMemcpyHostToDevice));
            CREATE_HIP(hipMemcpy(is.gathered[b], inf_rows.data(), mine_bytes * num_devices, hipMemcpyHostToDevice));
            CREATE_HIP(hipEventCreateWithFlags(&is.packed[b], ev_flags));
            CREATE_HIP(hipEventCreateWithFlags(&is.arrived[b], ev_flags));
        }
    }
    if (g->use_rccl) {
        std::string err;
        g->rccl = load_rccl(err);
        if (!g->rccl) CREATE_FAIL(SOTS_ERR_HIP, "RCCL unavailable: %s", err.c_str());
        std::vector<ncclComm_t> comms(num_devices);
        std::vector<int> devs(devices, devices + num_devices);
        const ncclResult_t r = g->rccl->CommInitAll(comms.data(), (int)num_devices, devs.data());
        if (r != ncclSuccess) CREATE_FAIL(SOTS_ERR_HIP, "ncclCommInitAll: %s", g->rccl->GetErrorString(r));
        for (uint32_t i = 0; i < num_devices; ++i) g->islands[i].comm = comms[i];
    }
#undef CREATE_HIP
#undef CREATE_FAIL
    {
        // sortPopulation places rows 0..S-1 per generation where the selection applies (enum sots_sort_mode); elites beyond
        // them need the completed order, i.e. the separate pack launch
        const uint32_t block = island_cfg->workgroup_size ? island_cfg->workgroup_size : 1u;
        const uint32_t npb = island_cfg->num_parents / block ? island_cfg->num_parents / block : 1u;
        const uint32_t placed = npb * block > island_cfg->num_parents ? npb * block : island_cfg->num_parents;
        g->fused = g->elites <= placed && !(flags & SOTS_GROUP_UNFUSED);
        g->host_gated = g->fused && (flags & SOTS_GROUP_OVERLAP) && !(flags & SOTS_GROUP_EVENT_WAITS);
    }
    g->rcs.assign(num_devices, SOTS_OK);
    g->barrier = new SpinBarrier(num_devices);
    for (uint32_t i = 1; i < num_devices; ++i) g->workers.emplace_back(worker_main, g, i);
    *out = g;
    return SOTS_OK;
}

void sots_group_destroy(sots_group *g) { destroy_group(g); }

const char *sots_group_last_error(const sots_group *g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

uint32_t sots_group_size(const sots_group *g) { return g ? (uint32_t)g->islands.size() : 0u; }

sots_ctx *sots_group_island(sots_group *g, uint32_t i) { return g && i < g->islands.size() ? g->islands[i].ctx : nullptr; }

int sots_group_uses_rccl(const sots_group *g) { return g && g->use_rccl ? 1 : 0; }

int sots_group_set_target_audio(sots_group *g, const float *audio, uint32_t num_samples)
{
    if (!g) return gfail(nullptr, SOTS_ERR_INVALID, "null group");
    for (size_t i = 0; i < g->islands.size(); ++i)
        if (int rc = sots_set_target_audio(g->islands[i].ctx, audio, num_samples)) return gfail(g, rc, "island %zu: %s", i, sots_last_error(g->islands[i].ctx));
    return SOTS_OK;
}

int sots_group_set_target_spectrum(sots_group *g, const float *magnitudes, uint32_t num_bins)
{
    if (!g) return gfail(nullptr, SOTS_ERR_INVALID, "null group");
    for (size_t i = 0; i < g->islands.size(); ++i)
        if (int rc = sots_set_target_spectrum(g->islands[i].ctx, magnitudes, num_bins))
            return gfail(g, rc, "island %zu: %s", i, sots_last_error(g->islands[i].ctx));
    return SOTS_OK;
}

int sots_group_synchronize(sots_group *g)
{
    if (!g) return gfail(nullptr, SOTS_ERR_INVALID, "null group");
    for (Island &is : g->islands) {
        GROUP_HIP(g, hipSetDevice(is.device));
        GROUP_HIP(g, hipStreamSynchronize(is.side));
        GROUP_HIP(g, hipStreamSynchronize(is.stream));
    }
    return SOTS_OK;
}

int sots_group_init_population(sots_group *g, uint32_t chunk_index)
{
    if (!g) return gfail(nullptr, SOTS_ERR_INVALID, "null group");
    if (int rc = sots_group_synchronize(g)) return rc; // an all-gather in flight is dropped with the old population
    g->pending = -1;
    g->generation = 0;
    g->exchanges = 0;
    for (size_t i = 0; i < g->islands.size(); ++i)
        if (int rc = sots_init_population(g->islands[i].ctx, chunk_index)) return gfail(g, rc, "island %zu: %s", i, sots_last_error(g->islands[i].ctx));
    return SOTS_OK;
}

int sots_group_execute_generations(sots_group *g, uint32_t n)
{
    if (!g) return gfail(nullptr, SOTS_ERR_INVALID, "null group");
    if (n == 0) return SOTS_OK;
    const uint32_t islands = (uint32_t)g->islands.size();
    // the job goes to the persistent island threads; island 0 runs here
    if (islands > 1) g->gate.post(n, g->pending, false);
    int pending_out = -1;
    g->rcs[0] = run_island(g, 0, n, g->pending, &pending_out);
    if (islands > 1) g->gate.wait_done(islands - 1);
    const bool exchange = g->elites > 0 && (islands > 1 || (g->flags & SOTS_GROUP_FORCE_RCCL));
    if (exchange) g->exchanges += (g->generation + n) / g->interval - g->generation / g->interval;
    g->pending = pending_out;
    g->generation += n;
    for (uint32_t i = 0; i < islands; ++i)
        if (g->rcs[i]) return g->rcs[i]; // g->err holds the text of one of the failures
    return SOTS_OK;
}

int sots_group_best(sots_group *g, uint32_t *island, float *fitness)
{
    if (!g) return gfail(nullptr, SOTS_ERR_INVALID, "null group");
    uint32_t best_i = 0;
    float best_f = 0.0f;
    for (uint32_t i = 0; i < g->islands.size(); ++i) {
        Island &is = g->islands[i];
        // row 0 of every island is its best unless immigrants landed after the last sort; their fitness
        // travelled with them, so the minimum over the breeding rows is read
        sots_info info;
        if (int rc = sots_get_info(is.ctx, &info)) return gfail(g, rc, "island %u: %s", i, sots_last_error(is.ctx));
        std::vector<float> f(info.population_length);
        if (int rc = sots_read_population(is.ctx, nullptr, 0, nullptr, 0, f.data(), f.size() * sizeof(float)))
            return gfail(g, rc, "island %u: %s", i, sots_last_error(is.ctx));
        float m = f[0];
        for (float x : f)
            if (x < m) m = x;
        if (i == 0 || m < best_f) {
            best_f = m;
            best_i = i;
        }
    }
    if (island) *island = best_i;
    if (fitness) *fitness = best_f;
    return SOTS_OK;
}

} // extern "C"
