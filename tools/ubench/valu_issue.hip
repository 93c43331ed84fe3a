// Diagnostic micro-benchmark (not part of the product): how many cycles does one SIMD of gfx950 spend
// per wave64 VALU instruction with 1, 2, 3, 4 wavefronts resident on it?  Plain v_fma_f32, packed
// v_pk_fma_f32, and v_min3_u32 / v_cvt streams; 8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, int iters)
{
    __shared__ float pad[40 * 1024]; // 160 KiB: one workgroup per CU, so blockDim/256 waves per SIMD
    pad[threadIdx.x] = 0.f;
    float a[8];
    v2f p[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i, p[i] = v2f{a[i], a[i] + 1.f};
    const float m = 0.999f, c = 0.001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);
                else if constexpr (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], v2f{m, m}, v2f{c, c});
                else if constexpr (MODE == 2) a[i] = __uint_as_float(min(min(__float_as_uint(a[i] - 1.0f), __float_as_uint(a[i] + 1.0f)), __float_as_uint(a[i])));
                else a[i] = (float)(int)(a[i] * m) + c;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = pad[threadIdx.x];
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char *name, int instr_per_inner)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    const int iters = 2000;
    for (int waves_per_simd = 1; waves_per_simd <= 4; ++waves_per_simd) {
        const int threads = 256 * waves_per_simd;
        for (int rep = 0; rep < 3; ++rep) k<MODE><<<256, threads>>>(out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * threads / 64);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : h) sum += v;
        const double per_wave = sum / h.size();
        const double n_instr = (double)iters * 64 * instr_per_inner; // per wave
        printf("%-28s waves/SIMD=%d  cycles per instr per wave=%.2f  per SIMD=%.2f\n", name, waves_per_simd,
               per_wave / n_instr, per_wave / n_instr / waves_per_simd);
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("v_fma_f32", 1);
    run<1>("v_pk_fma_f32", 1);
    run<2>("sub,add,min3 (3 instr)", 3);
    run<3>("mul,cvt,cvt,add (4 instr)", 4);
    return 0;
}
