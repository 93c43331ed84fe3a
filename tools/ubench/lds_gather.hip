// Diagnostic micro-benchmark (not part of the product): what does one wave64 LDS GATHER cost on gfx950?
// k_synth reads its 128 KiB wavetable with 64 random addresses per ds_read_b32; every synthesis kernel of this
// repository ends up at 92-94 cycles per sample and CU with 8 such reads per sample, whatever else it does.
// Patterns (16 address registers per lane, cycled; 16 reads, then s_waitcnt lgkmcnt(0)):
//   linear     lane l reads word l + 64 r                       (conflict-free: the 2-cycle baseline)
//   halves     lanes l and l + 32 read the same bank, different words   (do lanes 0-31 and 32-63 conflict?)
//   mod32      lanes of one half read words that are equal mod 32 in PAIRS  (2-way conflict inside a half)
//   mod64      as mod32 but the pair differs by 32 words: equal mod 32, different mod 64 (32 or 64 banks?)
//   random     uniformly random words of a 32768-word table (the wavetable gather)
//   random32   the same with lanes 32-63 switched off (EXEC = low half)
//   random_b64 8-byte reads of random 8-byte slots
//   clones     random, but groups of 4 neighbouring lanes share the address (a partly converged population)
// waves per SIMD = 1, 2, 4 (one workgroup per CU).  Prints cycles per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

template <int BYTES>
__global__ __launch_bounds__(1024) void k(const uint32_t *addr, float *out, unsigned long long *cyc, int iters, int half)
{
    extern __shared__ float tab[]; // 128 KiB table + nothing else
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) tab[i] = (float)i;
    uint32_t a[16];
    const int lane = threadIdx.x & 63;
    for (int r = 0; r < 16; ++r) a[r] = addr[r * 64 + lane];
    __syncthreads();
    float s = 0.f;
    if (half && lane >= 32) { // EXEC = low half
        out[blockIdx.x * blockDim.x + threadIdx.x] = 0.f;
        return;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (BYTES == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(v[r]) : "v"(a[r]));
            else {
                float2 t;
                asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(a[r]));
                v[r] = t.x + t.y;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) s += v[r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// Gathers BESIDE vector work: per ds_read_b32 (random pattern) V independent v_fma_f32 of the same wavefront.  If the LDS and the
// vector unit overlap, a step costs max(gather, V x issue); if a gather in flight holds up the wavefront's (or the SIMD's)
// vector issue, it costs their sum.
template <int V>
__global__ __launch_bounds__(1024) void k_mix(const uint32_t *addr, float *out, unsigned long long *cyc, int iters)
{
    extern __shared__ float tab[];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) tab[i] = (float)i;
    uint32_t a[16];
    const int lane = threadIdx.x & 63;
    for (int r = 0; r < 16; ++r) a[r] = addr[r * 64 + lane];
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = lane * 0.001f + i;
    __syncthreads();
    float s = 0.f;
    float v[2][16];
    for (int r = 0; r < 16; ++r) v[0][r] = v[1][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int j = 0; j < V; ++j) f[(r * V + j) & 7] = __builtin_fmaf(f[(r * V + j) & 7], 0.999f, 0.001f);
                asm volatile("ds_read_b32 %0, %1" : "=v"(v[h][r]) : "v"(a[r]));
            }
            // the values read in the PREVIOUS half are consumed now (a whole half later: their latency is covered)
            if (h == 0) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory"); // (never more than 16 outstanding anyway)
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(v[h ^ 1][r]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) s += v[0][r] + v[1][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int V>
void run_mix(const uint32_t *d_addr, float *out, unsigned long long *cyc)
{
    hipFuncSetAttribute((const void *)k_mix<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int threads = 256 * wps;
        for (int rep = 0; rep < 2; ++rep) k_mix<V><<<256, threads, 131072>>>(d_addr, out, cyc, iters);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
        std::vector<unsigned long long> hc(256 * threads / 64);
        hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : hc) sum += v;
        const double per_step = sum / hc.size() / ((double)iters * 16); // one gather + V fma, as one wave sees it
        printf("mix V=%2d   waves/SIMD=%d  cycles per (gather + V fma) per wave=%6.2f  per SIMD=%6.2f  per CU and gather=%6.2f\n", V, wps,
               per_step, per_step / wps, per_step / (4 * wps));
    }
}

int main()
{
    uint32_t *d_addr; float *out; unsigned long long *cyc;
    hipMalloc(&d_addr, 16 * 64 * 4); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    std::mt19937 rng(12345);
    const char *names[] = {"linear", "halves", "mod32", "mod64", "random", "random32", "random_b64", "clones4", "clones16"};
    hipFuncSetAttribute((const void *)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int pat = 0; pat < 9; ++pat) {
        std::vector<uint32_t> h(16 * 64);
        for (int r = 0; r < 16; ++r)
            for (int l = 0; l < 64; ++l) {
                uint32_t w;
                switch (pat) {
                case 0: w = l + 64 * r; break;
                case 1: w = (l & 31) + 32 * ((l >> 5) * 7 + 2 * r); break;              // halves: same bank, different word
                case 2: w = (l & ~1) % 32 + (l >> 5) * 32 + 64 * ((l & 1) + 2 * r); break; // pairs equal mod 64 (and 32)
                case 3: w = (l & ~1) % 32 + 32 * ((l & 1)) + 64 * (l >> 5) + 128 * r; break; // pairs equal mod 32, differ mod 64
                case 6: w = (rng() % 16384) * 2; break;
                case 7: w = 0; break;
                default: w = rng() % 32768; break;
                }
                h[r * 64 + l] = w * 4;
            }
        if (pat == 7 || pat == 8) {
            const int g = pat == 7 ? 4 : 16;
            for (int r = 0; r < 16; ++r)
                for (int l = 0; l < 64; ++l) h[r * 64 + l] = (l % g == 0) ? (rng() % 32768) * 4 : h[r * 64 + l - l % g];
        }
        hipMemcpy(d_addr, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        const int iters = 4000;
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int threads = 256 * wps;
            for (int rep = 0; rep < 2; ++rep) {
                if (pat == 6) k<8><<<256, threads, 131072>>>(d_addr, out, cyc, iters, 0);
                else k<4><<<256, threads, 131072>>>(d_addr, out, cyc, iters, pat == 5);
            }
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            std::vector<unsigned long long> hc(256 * threads / 64);
            hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost);
            double sum = 0; for (auto v : hc) sum += v;
            const double per_wave = sum / hc.size() / ((double)iters * 16);  // cycles per instruction as one wave sees it
            printf("%-10s waves/SIMD=%d  cycles per read as one wave sees it=%6.2f  per CU (all %2d waves share the LDS)=%6.2f\n",
                   names[pat], wps, per_wave, 4 * wps, per_wave / (4 * wps));
        }
    }
    // random pattern again for the mixes
    {
        std::vector<uint32_t> h(16 * 64);
        for (auto &w : h) w = (rng() % 32768) * 4;
        hipMemcpy(d_addr, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    }
    run_mix<0>(d_addr, out, cyc);
    run_mix<2>(d_addr, out, cyc);
    run_mix<4>(d_addr, out, cyc);
    run_mix<8>(d_addr, out, cyc);
    run_mix<12>(d_addr, out, cyc);
    return 0;
}
