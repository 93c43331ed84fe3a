// Diagnostic micro-benchmark (not part of the product): for how many cycles does a store instruction hold the
// wavefront that issues it?  One wavefront per SIMD (the LDS array leaves room for one workgroup per CU), each
// wavefront runs groups of G dependent v_fma_f32 followed by one store of W dwords per lane; the VALU-only time
// (stores compiled out) is subtracted.  Address patterns: "lines" = 8 lanes per 128-byte line, 8 lines per
// instruction (the synthesis tile flush), "flat" = 64 lanes contiguous.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int W, int G, bool STORE, bool LINES>
__global__ __launch_bounds__(256) void k(float *buf, unsigned long long *cyc, float *sink, int iters, unsigned ring_mask)
{
    __shared__ float pad[40 * 1024];
    pad[threadIdx.x] = 0.f;
    const unsigned lane = threadIdx.x & 63, wave = blockIdx.x * 4 + threadIdx.x / 64;
    float a = threadIdx.x * 0.001f;
    float d[4] = {a, a + 1, a + 2, a + 3};
    // per wavefront a private ring of (ring_mask + 1) bytes
    char *base = reinterpret_cast<char *>(buf) + (size_t)wave * (ring_mask + 1);
    const unsigned lane_off = LINES ? (lane >> 3) * 4224u + (lane & 7) * (W * 4u) : lane * (W * 4u);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned pos = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int g = 0; g < G; ++g) a = __builtin_fmaf(a, 0.999f, 0.001f);
            if constexpr (STORE) {
                __builtin_amdgcn_sched_barrier(0);
                char *p = base + ((pos + lane_off) & ring_mask);
                if constexpr (W == 1) *reinterpret_cast<float *>(p) = d[0];
                else if constexpr (W == 2) *reinterpret_cast<float2 *>(p) = make_float2(d[0], d[1]);
                else *reinterpret_cast<float4 *>(p) = make_float4(d[0], d[1], d[2], d[3]);
                __builtin_amdgcn_sched_barrier(0);
                pos += LINES ? 8 * 4224u : 64 * W * 4u;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * 256 + threadIdx.x] = a + pad[threadIdx.x];
    if (lane == 0) cyc[wave] = t1 - t0;
}

template <int W, int G, bool LINES>
void run(unsigned ring_bytes)
{
    const unsigned ring = ring_bytes;
    float *buf, *sink; unsigned long long *cyc;
    hipMalloc(&buf, (size_t)1024 * ring + 65536); hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    const int iters = 500;
    double t[2];
    for (int s = 0; s < 2; ++s) {
        for (int rep = 0; rep < 3; ++rep) {
            if (s) k<W, G, true, LINES><<<256, 256>>>(buf, cyc, sink, iters, ring - 1);
            else k<W, G, false, LINES><<<256, 256>>>(buf, cyc, sink, iters, ring - 1);
        }
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(1024);
        hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : h) sum += v;
        t[s] = sum / 1024 / (iters * 8.0);
    }
    printf("%-6s dwordx%d  %2d fma between stores: %6.1f cycles per group with the store, %6.1f without -> %5.1f per store (%.1f B/cycle)\n",
           LINES ? "lines" : "flat", W, G, t[1], t[0], t[1] - t[0], 64.0 * W * 4 / (t[1] - t[0]));
    hipFree(buf); hipFree(sink); hipFree(cyc);
}

int main(int argc, char **argv)
{
    const unsigned ring = argc > 1 ? (unsigned)atoi(argv[1]) : 8192u; // bytes per wavefront: 8 KiB x 1024 wavefronts stay in L2
    printf("ring %u bytes per wavefront\n", ring);
    run<4, 0, true>(ring); run<4, 8, true>(ring); run<4, 16, true>(ring); run<4, 32, true>(ring); run<4, 64, true>(ring);
    run<2, 0, true>(ring); run<2, 8, true>(ring); run<2, 16, true>(ring);
    run<1, 0, true>(ring); run<1, 8, true>(ring);
    run<4, 0, false>(ring); run<4, 16, false>(ring); run<2, 8, false>(ring); run<1, 4, false>(ring);
    return 0;
}
