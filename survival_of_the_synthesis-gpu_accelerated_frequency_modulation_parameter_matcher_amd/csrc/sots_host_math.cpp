// sots_host_math.cpp -- see sots_host_math.h.  Compiled with -ffp-contract=off.  The wavetable, window and
// target spectrum are defined once, in host/Objective_tables.hpp; only the device-side twiddle table lives here.
#include "sots_host_math.h"

#include <cmath>
#include <complex>

#include "../../include/sots_hip.h"
#include "../host/Objective_tables.hpp" // the one definition of the Objective's tables, shared with the C++ drop-in

namespace sots {

static const double kPi = sots_tables::kPi;

std::vector<float> make_wavetable()
{
    std::vector<float> t(SOTS_WAVETABLE_SIZE);
    sots_tables::wavetable(t.data());
    return t;
}

std::vector<double> make_window(uint32_t n, float *factor)
{
    std::vector<double> w(n);
    *factor = sots_tables::window(w.data(), n);
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

std::vector<float> target_spectrum(const float *audio, uint32_t n, const std::vector<double> &window,
                                   float window_factor)
{
    std::vector<float> mag(n / 2);
    sots_tables::target_spectrum(audio, n, window.data(), window_factor, mag.data());
    return mag;
}

} // namespace sots
