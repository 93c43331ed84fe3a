// Diagnostic micro-benchmark (not part of the product): what an exchange on a SIDE stream costs the compute stream.
// Stream A runs a chain of three dependent ~45 us kernels per "generation"; per generation
//   mode 0: nothing else;
//   mode 1: + hipEventRecord(packed) on A;
//   mode 2: + stream B waits for it, runs a tiny kernel (the collective's stand-in), records `arrived`;
//   mode 3: + A waits for the PREVIOUS generation's `arrived` before it starts (the overlapped schedule);
//   mode 4: as 3 with the tiny kernel on A itself, no second stream (the same-generation schedule);
//   mode 5: as 3 with stream memory operations instead of events for B -> A: B ends with hipStreamWriteValue32(flag, r),
//           A starts with hipStreamWaitValue32(flag >= r - 1) (signal memory, hipExtMallocWithFlags);
//   mode 6: as 5, and A -> B too (no events at all).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void spin(float *x, int iters)
{
    float a = x[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; ++i) a = __builtin_fmaf(a, 0.999f, 0.001f);
    x[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void tiny(float *y) { y[threadIdx.x] += 1.0f; }

int main()
{
    float *x, *y;
    hipMalloc(&x, 1024 * 256 * 4); hipMemset(x, 0, 1024 * 256 * 4);
    hipMalloc(&y, 4096); hipMemset(y, 0, 4096);
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipEvent_t packed[2], arrived[2];
    for (int i = 0; i < 2; ++i) {
        hipEventCreateWithFlags(&packed[i], hipEventDisableTiming);
        hipEventCreateWithFlags(&arrived[i], hipEventDisableTiming);
    }
    uint32_t *flag = nullptr, *flag2 = nullptr;
    if (hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory) != hipSuccess || hipExtMallocWithFlags((void **)&flag2, 8, hipMallocSignalMemory) != hipSuccess) {
        printf("no signal memory\n");
        return 1;
    }
    const int reps = 2000, iters = 1500;
    for (int mode = 0; mode < 7; ++mode) {
        *flag = 0; *flag2 = 0;
        for (int warm = 0; warm < 2; ++warm) {
            hipStreamSynchronize(a); hipStreamSynchronize(b);
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) {
                const int cur = r & 1, prev = cur ^ 1;
                if (mode == 3 && r > 0) hipStreamWaitEvent(a, arrived[prev], 0);
                if (mode >= 5 && r > 0) hipStreamWaitValue32(a, flag, (uint32_t)r, hipStreamWaitValueGte, 0xFFFFFFFFu); // written at the end of generation r - 1
                for (int k = 0; k < 3; ++k) spin<<<1024, 256, 0, a>>>(x, iters);
                if (mode >= 1 && mode <= 3) hipEventRecord(packed[cur], a);
                if (mode == 2 || mode == 3) {
                    hipStreamWaitEvent(b, packed[cur], 0);
                    tiny<<<1, 64, 0, b>>>(y);
                    hipEventRecord(arrived[cur], b);
                }
                if (mode == 4) tiny<<<1, 64, 0, a>>>(y);
                if (mode == 5) {
                    hipEventRecord(packed[cur], a);
                    hipStreamWaitEvent(b, packed[cur], 0);
                    tiny<<<1, 64, 0, b>>>(y);
                    hipStreamWriteValue32(b, flag, (uint32_t)(r + 1), 0);
                }
                if (mode == 6) {
                    hipStreamWriteValue32(a, flag2, (uint32_t)(r + 1), 0);
                    hipStreamWaitValue32(b, flag2, (uint32_t)(r + 1), hipStreamWaitValueGte, 0xFFFFFFFFu);
                    tiny<<<1, 64, 0, b>>>(y);
                    hipStreamWriteValue32(b, flag, (uint32_t)(r + 1), 0);
                }
            }
            hipStreamSynchronize(a); hipStreamSynchronize(b);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            if (warm) printf("mode %d: %.2f us per generation\n", mode, us);
        }
    }
    return 0;
}
