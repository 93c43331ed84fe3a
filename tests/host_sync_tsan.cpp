// host_sync_tsan.cpp -- the island group's host-side synchronisation (csrc/sots_host_sync.h) under ThreadSanitizer, used
// the way sots_group.hip uses it: persistent workers that wait at a JobGate, run "generations" with a SpinBarrier per
// exchange, and report through a done counter; the caller posts jobs of varying length, with and without pauses long
// enough for the workers to fall asleep, then quits them.  Plain (non-atomic) data is handed back and forth on purpose:
// the job parameters, per-island results and an exchange buffer every thread writes before the barrier and reads after
// it - ThreadSanitizer reports any of it that the primitives do not order.  CPU only; built and run by
// tests/test_sanitizers.py.
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "../survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd/csrc/sots_host_sync.h"

using sots_host::JobGate;
using sots_host::SpinBarrier;

struct Group {
    uint32_t islands;
    JobGate gate;
    SpinBarrier barrier;
    std::vector<int> rcs;                 // per island, written by its thread, read by the caller after wait_done
    std::vector<uint64_t> packed;         // "packed[b][i]": written by island i before the barrier, read by all after it
    uint64_t generation = 0, exchanges = 0; // caller-owned between jobs, read by the workers inside one
    std::vector<uint64_t> checksum;       // per island
    explicit Group(uint32_t n) : islands(n), barrier(n), rcs(n, 0), packed(2 * n, 0), checksum(n, 0) {}
};

static int run_island(Group &g, uint32_t i, uint32_t n)
{
    uint64_t x = g.exchanges;
    for (uint32_t k = 0; k < n; ++k) {
        const int b = (int)(x & 1u);
        ++x;
        g.packed[(size_t)b * g.islands + i] = (g.generation + k) * 1000 + i; // "pack"
        g.barrier.arrive_and_wait();                                          // every island has packed
        uint64_t sum = 0;
        for (uint32_t j = 0; j < g.islands; ++j) sum += g.packed[(size_t)b * g.islands + j]; // "gather"
        g.checksum[i] += sum;
        // the buffer of parity b is written again two exchanges later: the barrier of the exchange in between orders that
    }
    return (int)i;
}

static void worker(Group *g, uint32_t i)
{
    uint64_t seen = 0;
    for (;;) {
        g->gate.wait_job(seen);
        seen = g->gate.seq.load(std::memory_order_acquire);
        if (g->gate.quit) return;
        g->rcs[i] = run_island(*g, i, g->gate.n);
        g->gate.done.fetch_add(1, std::memory_order_release);
    }
}

int main()
{
    for (uint32_t islands : {2u, 3u, 8u}) {
        Group g(islands);
        std::vector<std::thread> threads;
        for (uint32_t i = 1; i < islands; ++i) threads.emplace_back(worker, &g, i);
        uint64_t expect = 0;
        for (int job = 0; job < 300; ++job) {
            const uint32_t n = 1 + (uint32_t)(job % 5);
            g.gate.post(n, -1, false);
            g.rcs[0] = run_island(g, 0, n);
            g.gate.wait_done(islands - 1);
            for (uint32_t i = 0; i < islands; ++i)
                if (g.rcs[i] != (int)i) return printf("island %u reported %d\n", i, g.rcs[i]), 1;
            for (uint32_t k = 0; k < n; ++k) {
                uint64_t sum = 0;
                for (uint32_t j = 0; j < islands; ++j) sum += (g.generation + k) * 1000 + j;
                expect += sum;
            }
            g.generation += n;
            g.exchanges += n;
            if (job % 97 == 96) std::this_thread::sleep_for(std::chrono::milliseconds(30)); // the workers fall asleep on the condition variable
        }
        g.gate.post(0, -1, true);
        for (auto &t : threads) t.join();
        for (uint32_t i = 0; i < islands; ++i)
            if (g.checksum[i] != expect) return printf("islands %u: island %u saw %llu, expected %llu\n", islands, i,
                                                       (unsigned long long)g.checksum[i], (unsigned long long)expect), 1;
        printf("islands %u ok: %llu generations, checksum %llu\n", islands, (unsigned long long)g.generation, (unsigned long long)expect);
    }
    return 0;
}
