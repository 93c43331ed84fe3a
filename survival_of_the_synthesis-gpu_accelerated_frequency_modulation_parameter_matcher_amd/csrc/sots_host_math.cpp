// sots_host_math.cpp -- see sots_host_math.h.  Compiled with -ffp-contract=off.
#include "sots_host_math.h"

#include <cmath>
#include <complex>

#include "../../include/sots_hip.h"

namespace sots {

static const double kPi = 3.14159265358979323846; // the reference's M_PI, Evolutionary_Strategy.hpp:5

std::vector<float> make_wavetable()
{
    std::vector<float> t(SOTS_WAVETABLE_SIZE);
    const float inv = 1.0f / ((float)SOTS_WAVETABLE_SIZE - 1.0f);
    for (uint32_t i = 0; i < SOTS_WAVETABLE_SIZE; ++i)
        t[i] = sinf((float)i * inv * 2 * (float)kPi);
    return t;
}

std::vector<double> make_window(uint32_t n, float *factor)
{
    std::vector<double> w(n);
    const float one_over = 1.0f / (float)n; // fftOneOverSize is a float, :296
    const double two_pi = 2.0 * kPi;
    float f = 0.0f;                         // fftWindowFactor is a float, :225
    for (uint32_t i = 0; i < n; ++i) {
        w[i] = (1.0 - std::cos((double)i * (one_over - 1) * two_pi));
        f += w[i];
    }
    f *= one_over;
    *factor = f;
    return w;
}

std::vector<float> make_twiddles(uint32_t n)
{
    std::vector<float> t(2 * (size_t)n);
    for (uint32_t q = 0; q < n; ++q) {
        const double a = -2.0 * kPi * (double)q / (double)n;
        double c = std::cos(a), s = std::sin(a);
        // the four axis points are exact
        if (n % 4 == 0 && q % (n / 4) == 0) {
            const uint32_t quad = q / (n / 4);
            c = quad == 0 ? 1.0 : quad == 2 ? -1.0 : 0.0;
            s = quad == 1 ? -1.0 : quad == 3 ? 1.0 : 0.0;
        }
        t[2 * q] = (float)c;
        t[2 * q + 1] = (float)s;
    }
    return t;
}

// in-place decimation-in-frequency radix-2 on n complex doubles, output bit-reversed,
// then un-permuted.  Runs once per audio chunk, so clarity wins over speed.
static void fft_c2c(std::vector<std::complex<double>> &a)
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

std::vector<float> target_spectrum(const float *audio, uint32_t n, const std::vector<double> &window,
                                   float window_factor)
{
    std::vector<std::complex<double>> a(n);
    for (uint32_t i = 0; i < n; ++i) a[i] = std::complex<double>(audio[i] * window[i], 0.0);
    fft_c2c(a);
    const float one_over_size = 1.0f / (float)n;
    const float one_over_wf = 1.f / window_factor;
    std::vector<float> mag(n / 2);
    for (uint32_t k = 0; k < n / 2; ++k) {
        const float raw = hypotf((float)a[k].real(), (float)a[k].imag());
        mag[k] = raw * one_over_size * one_over_wf;
    }
    return mag;
}

} // namespace sots
