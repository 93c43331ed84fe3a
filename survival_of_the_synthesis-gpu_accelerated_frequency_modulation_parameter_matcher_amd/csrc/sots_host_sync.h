// sots_host_sync.h -- the two host-side synchronisation primitives of the island group (sots_group.hip): a spinning
// barrier for the island threads and the gate through which the caller's thread hands them a job.  No HIP in here, so
// that the CPU test suite can run them under ThreadSanitizer (tests/host_sync_tsan.cpp, tests/test_sanitizers.py).
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <thread>

namespace sots_host {

inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

// Reusable barrier for the island threads.  They meet once per exchange, a few microseconds apart (each has just
// enqueued the same few launches), so they spin; a thread that has spun for long (an island fewer CPUs than threads)
// yields its time slice.
class SpinBarrier
{
    std::atomic<uint32_t> waiting_{0}, phase_{0};
    uint32_t n_;

public:
    explicit SpinBarrier(uint32_t n) : n_(n) {}
    void arrive_and_wait()
    {
        const uint32_t phase = phase_.load(std::memory_order_acquire);
        if (waiting_.fetch_add(1, std::memory_order_acq_rel) + 1 == n_) {
            waiting_.store(0, std::memory_order_relaxed);
            phase_.store(phase + 1, std::memory_order_release);
            return;
        }
        for (uint32_t spins = 0; phase_.load(std::memory_order_acquire) == phase; ++spins) {
            if (spins < 4096) cpu_relax();
            else std::this_thread::yield();
        }
    }
};

// What the caller's thread tells the workers.  Workers spin on `seq` for a while after a job (the next call usually
// follows at once) and then sleep on the condition variable; `seq` only changes under the mutex, so no wake-up is lost.
struct JobGate {
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<uint64_t> seq{0};
    std::atomic<uint32_t> done{0};
    uint32_t n = 0;        // generations of the current job
    int pending_in = -1;   // the group's `pending` when the job was posted
    bool quit = false;

    void post(uint32_t n_generations, int pending, bool quit_now)
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            n = n_generations;
            pending_in = pending;
            quit = quit_now;
            done.store(0, std::memory_order_relaxed);
            seq.fetch_add(1, std::memory_order_release);
        }
        cv.notify_all();
    }
    // returns once seq != seen
    void wait_job(uint64_t seen)
    {
        for (uint32_t spins = 0; spins < 50000; ++spins) { // a millisecond or two: the next call of a generation loop comes within ~0.1 ms
            if (seq.load(std::memory_order_acquire) != seen) return;
            cpu_relax();
        }
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return seq.load(std::memory_order_acquire) != seen; });
    }
    void wait_done(uint32_t workers)
    {
        for (uint32_t spins = 0; done.load(std::memory_order_acquire) != workers; ++spins) {
            if (spins < 4096) cpu_relax();
            else std::this_thread::yield();
        }
    }
};

} // namespace sots_host
