// Diagnostic micro-benchmark (not part of the product): what per-kernel timing costs on the stream.
// A chain of dependent ~50 us kernels, timed (a) not at all, (b) with a hipEventRecord pair per kernel, (c) with one shared
// event per kernel boundary, (d) with the start/stop events of hipExtLaunchKernelGGL (timestamps of the dispatch itself).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(float *x, int iters)
{
    float a = x[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; ++i) a = __builtin_fmaf(a, 0.999f, 0.001f);
    x[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

int main()
{
    float *x; hipMalloc(&x, 1024 * 256 * 4); hipMemset(x, 0, 1024 * 256 * 4);
    hipStream_t st; hipStreamCreate(&st);
    const int kernels = 3, reps = 2000, iters = 1700;
    std::vector<hipEvent_t> ev(2 * kernels * reps + 2);
    for (auto &e : ev) hipEventCreate(&e);
    for (int mode = 0; mode < 4; ++mode) {
        for (int warm = 0; warm < 2; ++warm) {
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            int e = 0;
            for (int r = 0; r < reps; ++r)
                for (int k = 0; k < kernels; ++k) {
                    if (mode == 1) hipEventRecord(ev[e++], st);
                    if (mode == 2 && k == 0) hipEventRecord(ev[e++], st);
                    if (mode == 3) hipExtLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, st, ev[e], ev[e + 1], 0, x, iters), e += 2;
                    else spin<<<1024, 256, 0, st>>>(x, iters);
                    if (mode == 1 || mode == 2) hipEventRecord(ev[e++], st);
                }
            hipStreamSynchronize(st);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * kernels);
            if (warm) {
                float ms = 0;
                if (mode == 1) hipEventElapsedTime(&ms, ev[0], ev[1]);
                if (mode == 2) hipEventElapsedTime(&ms, ev[0], ev[1]);
                if (mode == 3) hipEventElapsedTime(&ms, ev[0], ev[1]);
                printf("mode %d (%s): %.2f us per kernel on the wall; first kernel by its events %.2f us\n", mode,
                       mode == 0 ? "no events" : mode == 1 ? "event pair per kernel" : mode == 2 ? "one event per boundary" : "hipExtLaunchKernelGGL events",
                       us, ms * 1e3);
            }
        }
    }
    return 0;
}
