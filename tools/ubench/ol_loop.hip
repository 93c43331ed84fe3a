// Diagnostic micro-benchmark (not part of the product): the steady-state trip of k_synth_ol (operators in the lanes) in
// isolation - 8 samples per trip: table read at the phase, phase += increment, wrap, the hand-over arithmetic of the block read one
// trip ago, the row shift - with pieces switched off by MODE bits, at 1, 2, 4 wavefronts per SIMD:
//   1 no table reads   2 no row shift   8 no wraps   16 no index arithmetic (conflict-free reads)   64 plain instead of packed arithmetic
//   128 no hand-over arithmetic
// Prints shader cycles per SAMPLE per wavefront (all wavefronts of the CU run in parallel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr float kW = 32768.f;

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, int trips, float seed)
{
    extern __shared__ float tab[];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) tab[i] = __sinf(i * 1.9e-4f);
    const int lane = threadIdx.x & 63;
    const float c = 0.743f, pm = 30.f + lane + seed, po = 200.f + 3 * lane, wlo = (lane & 12) ? kW : 0.f;
    float pos = 0.f, inc[8], T[2][8];
    for (int u = 0; u < 8; ++u) inc[u] = 10.f + lane * (u + 1), T[0][u] = 0.f, T[1][u] = 0.f;
    __syncthreads();
    float sink = 0.f;
    auto trip = [&](auto q_tag) {
        constexpr int Q = decltype(q_tag)::value;
        if constexpr (MODE & 1024) {
            // the row shift folded into the phase add (v_add_f32_dpp row_ror:4; the last operator's lanes make operator 0's constant
            // with one more packed multiply per two samples: (t pm z + po) c, z = 0 there): no v_mov_b32_dpp, no increments kept
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int i = (int)pos;
                i = min(max(i, 0), 32767);
                T[Q][u] = tab[i];
                float a;
                asm("v_add_f32_dpp %0, %1, %2 row_ror:4 row_mask:0xf bank_mask:0xf" : "=v"(a) : "v"(inc[u]), "v"(pos));
                const v2f bc = v2f{a, a} + v2f{-kW, wlo};
                pos = __uint_as_float(min(min(__float_as_uint(bc.x), __float_as_uint(bc.y)), __float_as_uint(a)));
            }
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const v2f t = ((v2f{T[Q ^ 1][u], T[Q ^ 1][u + 1]} * pm) * wlo + po) * c;
                inc[u] = t.x, inc[u + 1] = t.y;
            }
            return;
        }
        if constexpr (MODE & 256) {
            // three dependency chains per sample, interleaved instruction by instruction and pinned: index -> read (I), phase add ->
            // wrap (W), hand-over arithmetic -> row shift of the block read one trip ago (H)
#define SB() __builtin_amdgcn_sched_barrier(0)
            v2f h1, h2, h3;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int i1 = (int)pos;                                       // I1
                float a = pos + inc[u];                                   // W1
                if (!(u & 1)) h1 = v2f{T[Q ^ 1][u], T[Q ^ 1][u + 1]} * pm; // H1 (even samples)
                else inc[u - 1] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(inc[u - 1]), __float_as_int(h3.x), 0x114, 0xf, 0xf, false)); // D1
                SB();
                i1 = min(max(i1, 0), 32767);                              // I2
                const v2f bc = v2f{a, a} + v2f{-kW, wlo};                 // W2
                if (!(u & 1)) h2 = h1 + po;                               // H2
                SB();
                uint32_t off = (uint32_t)i1 << 2;                         // I3
                asm volatile("" : "+v"(off));
                pos = __uint_as_float(min(min(__float_as_uint(bc.x), __float_as_uint(bc.y)), __float_as_uint(a))); // W3
                if (!(u & 1)) h3 = h2 * c;                                // H3
                SB();
                T[Q][u] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(tab) + off); // read
                if (u & 1) inc[u] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(inc[u]), __float_as_int(h3.y), 0x114, 0xf, 0xf, false)); // D2: AFTER this sample's W1 has used inc[u]
                SB();
            }
#undef SB
            return;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (MODE & 1) T[Q][u] = pos * 1e-9f;
            else if constexpr (MODE & 16) T[Q][u] = tab[lane + 64 * u];
            else {
                int i = (int)pos;
                i = min(max(i, 0), 32767);
                T[Q][u] = tab[i];
            }
            pos += inc[u];
            if constexpr (!(MODE & 8)) {
                const v2f bc = v2f{pos, pos} + v2f{-kW, wlo};
                pos = __uint_as_float(min(min(__float_as_uint(bc.x), __float_as_uint(bc.y)), __float_as_uint(pos)));
            }
        }
        float o[8];
        if constexpr (MODE & 128) {
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = T[Q ^ 1][u];
        } else if constexpr (MODE & 64) {
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = (T[Q ^ 1][u] * pm + po) * c;
        } else {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const v2f t = (v2f{T[Q ^ 1][u], T[Q ^ 1][u + 1]} * pm + po) * c;
                o[u] = t.x, o[u + 1] = t.y;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (MODE & 2) inc[u] = o[u];
            else inc[u] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(inc[u]), __float_as_int(o[u]), 0x114, 0xf, 0xf, false));
        }
    };
    // MODE & 512: the wavefronts that share a SIMD (w, w + 4, w + 8, w + 12 of the workgroup) take turns at priority 1, a trip each
    const int turn = __builtin_amdgcn_readfirstlane(threadIdx.x / 256), sharers = blockDim.x / 256;
    auto prio = [&](int k) {
        if constexpr (MODE & 512) {
            const int mine = __builtin_amdgcn_readfirstlane((k % sharers) == turn ? 1 : 0);
            asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 1\n\ts_branch 2f\n1:\ts_setprio 0\n2:" ::"s"(mine));
        }
    };
    // MODE & 4096: progress feedback - every 16 trips a wavefront publishes its trip count in LDS and takes priority = the number
    // of wavefronts sharing its SIMD (w, w + 4, w + 8, w + 12 of the workgroup) that are AHEAD of it: the laggard issues first
    __shared__ int progress[16];
    if (threadIdx.x < 16) progress[threadIdx.x] = 0;
    __syncthreads();
    const int my_wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
    auto feedback = [&](int k) {
        if constexpr (MODE & 4096) {
            if ((k & 15) == 0) {
                if (lane == 0) progress[my_wave] = k;
                int ahead = 0;
                for (int q = 1; q < sharers; ++q) {
                    const int other = (my_wave + 4 * q) % (4 * sharers);
                    ahead += *(volatile int *)&progress[other] > k ? 1 : 0;
                }
                ahead = __builtin_amdgcn_readfirstlane(ahead);
                asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 10f\n\ts_cmp_eq_u32 %0, 1\n\ts_cbranch_scc1 11f\n\ts_cmp_eq_u32 %0, 2\n\ts_cbranch_scc1 12f\n\t"
                             "s_setprio 3\n\ts_branch 19f\n10:\ts_setprio 0\n\ts_branch 19f\n11:\ts_setprio 1\n\ts_branch 19f\n12:\ts_setprio 2\n19:" ::"s"(ahead));
            }
        }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < trips; k += 2) {
        feedback(k);
        prio(k);
        if constexpr (MODE & 2048) __builtin_amdgcn_s_setprio(1); // the SAME pattern in every wavefront: even trips at priority 1, odd at 0
        trip(std::integral_constant<int, 0>{});
        prio(k + 1);
        if constexpr (MODE & 2048) __builtin_amdgcn_s_setprio(0);
        trip(std::integral_constant<int, 1>{});
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int u = 0; u < 8; ++u) sink += inc[u] + T[0][u] + T[1][u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink + pos;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char *name, float *out, unsigned long long *cyc)
{
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int trips = 2048;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int threads = 256 * wps;
        for (int rep = 0; rep < 2; ++rep) k<MODE><<<256, threads, 131072>>>(out, cyc, trips, 0.5f);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
        std::vector<unsigned long long> h(256 * threads / 64);
        (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0, mn = 1e30, mx = 0;
        for (auto v : h) { sum += v; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        const double d = (double)trips * 8;
        printf("%-44s waves/SIMD=%d  cycles per sample per wave: mean %6.1f  min %6.1f  max %6.1f   (x %2d waves: per CU and wave-sample %5.2f)\n", name, wps,
               sum / h.size() / d, mn / d, mx / d, 4 * wps, sum / h.size() / d / (4 * wps));
    }
}

int main()
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 256 * 16 * 8);
    run<0>("full trip", out, cyc);
    run<1>("no table reads", out, cyc);
    run<16>("conflict-free reads, no index arithmetic", out, cyc);
    run<2>("no row shift", out, cyc);
    run<8>("no wraps", out, cyc);
    run<64>("plain instead of packed hand-over arithmetic", out, cyc);
    run<128>("no hand-over arithmetic", out, cyc);
    run<1024>("full trip, row shift folded into the phase add", out, cyc);
    run<4096>("full trip, priority by progress feedback", out, cyc);
    run<1024 + 2048>("folded, static priority 1 in even trips, 0 in odd", out, cyc);
    run<512>("full trip, priority by turns", out, cyc);
    run<64 + 512>("plain hand-over arithmetic, priority by turns", out, cyc);
    run<1 + 2 + 8>("phase adds + hand-over arithmetic only", out, cyc);
    run<1 + 2 + 8 + 128>("phase adds only", out, cyc);
    return 0;
}
