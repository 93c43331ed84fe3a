// Objective_tables.hpp -- the host-side tables of the reference's Objective, in ONE place: the wavetable
// (Evolutionary_Strategy.hpp:325-332), the doubled Hann window and its mean (:308-317) and the target
// magnitude spectrum (:524-542).  Included by the C++ drop-in's Evolutionary_Strategy.hpp and by the library
// (csrc/sots_host_math.cpp), which uploads the same tables the caller synthesises its target with.
// Header-only, plain C++17, no HIP.  Compile with -ffp-contract=off (the fp32 expressions round as written).
// The test oracle (oracle/) keeps its own, independent restatement: it is the checker.
#ifndef SOTS_OBJECTIVE_TABLES_HPP
#define SOTS_OBJECTIVE_TABLES_HPP

#include <cmath>
#include <complex>
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace sots_tables {

constexpr double kPi = 3.14159265358979323846; // the reference's M_PI, Evolutionary_Strategy.hpp:5
constexpr uint32_t kWavetableSize = 32768;     // :197

// table[i] = sinf(i / (W - 1) * 2 pi): the period is W - 1 entries, phases wrap at W (:328-331)
inline void wavetable(float *table)
{
    const float inv = 1.0f / ((float)kWavetableSize - 1.0f);
    for (uint32_t i = 0; i < kWavetableSize; ++i) table[i] = sinf((float)i * inv * 2 * (float)kPi);
}

// w[i] = 1 - cos(i (1/N - 1) 2 pi) in double (= 2 x periodic Hann); returns fftWindowFactor = sum(w) / N
// accumulated in a float as the reference does (:225,:296,:308-317)
inline float window(double *w, uint32_t n)
{
    const float one_over = 1.0f / (float)n;
    const double two_pi = 2.0 * kPi;
    float f = 0.0f;
    for (uint32_t i = 0; i < n; ++i) {
        w[i] = (1.0 - std::cos((double)i * (one_over - 1) * two_pi));
        f += w[i];
    }
    return f * one_over;
}

// in-place decimation-in-frequency radix-2 on n complex doubles (n a power of two), natural order out.
// Runs once per audio chunk, so clarity wins over speed.  (FFTW in the reference, :286,:511,:532.)
inline void forward_fft(std::vector<std::complex<double>> &a)
{
    const size_t n = a.size();
    for (size_t len = n; len >= 2; len >>= 1) {
        const size_t half = len / 2;
        for (size_t base = 0; base < n; base += len)
            for (size_t j = 0; j < half; ++j) {
                const double ang = -2.0 * kPi * (double)j / (double)len;
                const std::complex<double> w(std::cos(ang), std::sin(ang));
                const std::complex<double> u = a[base + j], v = a[base + j + half];
                a[base + j] = u + v;
                a[base + j + half] = (u - v) * w;
            }
    }
    size_t bits = 0;
    while (((size_t)1 << bits) < n) ++bits;
    for (size_t i = 0; i < n; ++i) {
        size_t r = 0;
        for (size_t b = 0; b < bits; ++b)
            if (i & ((size_t)1 << b)) r |= (size_t)1 << (bits - 1 - b);
        if (r > i) std::swap(a[i], a[r]);
    }
}

// Objective::calculateFFT (:524-542): double window x fp32 audio -> forward real DFT in fp64 ->
// hypotf((float)re, (float)im) / N / windowFactor for k < N/2
inline void target_spectrum(const float *audio, uint32_t n, const double *w, float window_factor, float *mag)
{
    std::vector<std::complex<double>> a(n);
    for (uint32_t i = 0; i < n; ++i) a[i] = std::complex<double>(audio[i] * w[i], 0.0);
    forward_fft(a);
    const float one_over_size = 1.0f / (float)n;
    const float one_over_wf = 1.f / window_factor;
    for (uint32_t k = 0; k < n / 2; ++k) {
        const float raw = hypotf((float)a[k].real(), (float)a[k].imag());
        mag[k] = raw * one_over_size * one_over_wf;
    }
}

} // namespace sots_tables

#endif
