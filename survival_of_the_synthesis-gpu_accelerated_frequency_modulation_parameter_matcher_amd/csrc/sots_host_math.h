// sots_host_math.h -- host-side tables of the Objective (Evolutionary_Strategy.hpp:175-577)
// that the reference also builds on the CPU and uploads: wavetable, Hann window, window
// factor, target magnitude spectrum.  Product code (not the test oracle).
#pragma once

#include <cstdint>
#include <vector>

namespace sots {

// Objective::initWavetable, Evolutionary_Strategy.hpp:325-332
std::vector<float> make_wavetable();

// Objective::initFFTW window loop, Evolutionary_Strategy.hpp:308-317.
// Returns the double window; *factor = fftWindowFactor (float).
std::vector<double> make_window(uint32_t n, float *factor);

// e^{-2 pi i q / n}, q < n, as interleaved (re, im) floats
std::vector<float> make_twiddles(uint32_t n);

// Objective::calculateFFT, Evolutionary_Strategy.hpp:524-542: double window x fp32 audio ->
// forward real DFT in fp64 -> hypotf(re, im) / N / windowFactor for k < N/2.
std::vector<float> target_spectrum(const float *audio, uint32_t n, const std::vector<double> &window,
                                   float window_factor);

} // namespace sots
