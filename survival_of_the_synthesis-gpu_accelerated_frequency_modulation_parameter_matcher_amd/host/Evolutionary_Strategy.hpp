// Evolutionary_Strategy.hpp -- the strategy API the HIP backend plugs in behind.
//
// Same public surface as the reference's Evolutionary_Strategy.hpp (Population :19-173,
// Objective :175-577, Evolutionary_Strategy_Arguments :579-589, Evolutionary_Strategy
// :591-696), so that a driver written against the reference (main.cpp:105-280) compiles
// against this header unchanged.  The reference header itself cannot be used: it includes
// <fftw_cpp.hh> and links FFTW (Evolutionary_Strategy.hpp:11,286).  Everything here is
// re-implemented: the only FFT the host side needs (the target spectrum, once per audio
// chunk) is a small built-in fp64 transform.
//
// Behavioural fixes relative to the reference, each marked [fix] where it happens:
//   * oscillator phases start at zero for every synthesise call (the reference keeps them in
//     Objective members across calls, :178-180, which makes a call depend on its predecessor);
//   * buffers are sized with new[] (the reference allocates single floats with new float(n),
//     :236-244, and then writes n of them);
//   * scaleParams-style scaling min + v*(max-min) is used by the synthesisers too (the
//     reference's CPU synths multiply by max only, :371; identical for the shipped mins = 0).
#ifndef SOTS_EVOLUTIONARY_STRATEGY_HPP
#define SOTS_EVOLUTIONARY_STRATEGY_HPP

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#include "Objective_tables.hpp" // wavetable, window, target spectrum: one definition, shared with libsots_hip

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <vector>

#if defined(__clang__)
#define SOTS_NO_CONTRACT _Pragma("clang fp contract(off)")
#define SOTS_NO_CONTRACT_ATTR
#elif defined(__GNUC__)
#define SOTS_NO_CONTRACT
#define SOTS_NO_CONTRACT_ATTR __attribute__((optimize("fp-contract=off")))
#else
#define SOTS_NO_CONTRACT
#define SOTS_NO_CONTRACT_ATTR
#endif

struct Individual
{
    uint32_t fitness;
};

// Host-side array-of-structures view of a population: per individual
// [v0, s0, v1, s1, ..., v(D-1), s(D-1), fitness] = 2D+1 floats (Evolutionary_Strategy.hpp:29-43).
struct Population
{
    uint32_t numParents = 197;
    uint32_t numOffspring = 960;
    uint32_t numDimensions = 4;
    uint32_t populationLength = numParents + numOffspring;
    uint32_t populationSize = (numParents + numOffspring) * sizeof(float);

    float *data = nullptr;

    uint32_t rowLength() const { return numDimensions * 2 + 1; }
    float *getValue(uint32_t idxIndividual, uint32_t idxValue) { return &data[idxIndividual * rowLength() + idxValue * 2]; }
    float *getStep(uint32_t idxIndividual, uint32_t idxStep) { return &data[idxIndividual * rowLength() + idxStep * 2 + 1]; }
    float *getFitness(uint32_t idxIndividual) { return &data[idxIndividual * rowLength() + numDimensions * 2]; }

    void swap(int32_t first, int32_t second)
    {
        std::swap_ranges(data + (size_t)first * rowLength(), data + (size_t)(first + 1) * rowLength(),
                         data + (size_t)second * rowLength());
    }

    // Ascending by fitness, equal fitness keeps its order: the result of the reference's
    // bubble sort (:108-124) in O(P log P).
    void bubbleSortPopulation()
    {
        const uint32_t w = rowLength();
        std::vector<uint32_t> order(populationLength);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            const float fa = data[a * w + w - 1], fb = data[b * w + w - 1];
            if (fa != fa) return false; // NaN sorts last
            if (fb != fb) return true;
            return fa < fb;
        });
        std::vector<float> sorted((size_t)populationLength * w);
        for (uint32_t r = 0; r < populationLength; ++r)
            std::copy(data + (size_t)order[r] * w, data + (size_t)(order[r] + 1) * w, sorted.begin() + (size_t)r * w);
        std::copy(sorted.begin(), sorted.end(), data);
    }
    void quickSortPopulation(int /*left*/, int /*right*/) { bubbleSortPopulation(); }
};

// The synthesiser + analysis front end (Evolutionary_Strategy.hpp:175-577).
class Objective
{
public:
    const std::vector<float> paramMins;
    const std::vector<float> paramMaxs;

    std::vector<float> targetParams;

    const uint32_t sampleRate = 44100;
    const uint32_t wavetableSize = 32768;
    float *wavetable = nullptr;

    const float w2srRatio = wavetableSize / (float)sampleRate; // :203
    uint32_t audioLength;
    uint32_t audioLengthLog2;

    uint32_t fftOutSize;  // floats per device spectrum row, N + 8 (:268, ...OpenCL.hpp:167)
    uint32_t fftSize;
    uint32_t fftSizeLog2;
    uint32_t fftHalfSize;
    float fftOneOverSize;
    double *fftWindow = nullptr;
    double *fftWindowedAudio = nullptr;
    float fftWindowFactor;
    float fftOneOverWindowFactor;

    Objective(uint32_t /*aPopulationSize*/, uint32_t /*aNumDimensions*/, const std::vector<float> aParamMins,
              const std::vector<float> aParamMaxs, uint32_t aAudioLengthLog2)
        : paramMins(aParamMins), paramMaxs(aParamMaxs), audioLength(1u << aAudioLengthLog2),
          audioLengthLog2(aAudioLengthLog2)
    {
        fftSizeLog2 = audioLengthLog2;
        fftSize = 1u << fftSizeLog2;
        fftHalfSize = fftSize / 2;
        fftOneOverSize = 1.0f / fftSize;
        fftOutSize = (audioLength / 2 + 4) * 2;
        fftWindow = new double[fftSize];
        fftWindowedAudio = new double[fftSize];
        // Hann window scaled by two, and its mean (:308-317)
        fftWindowFactor = sots_tables::window(fftWindow, fftSize);
        fftOneOverWindowFactor = 1.f / fftWindowFactor;
        initWavetable();
    }
    Objective(const Objective &) = delete;
    Objective &operator=(const Objective &) = delete;
    ~Objective()
    {
        delete[] wavetable;
        delete[] fftWindow;
        delete[] fftWindowedAudio;
    }

    void initWavetable() // :325-332
    {
        if (wavetable) return;
        wavetable = new float[wavetableSize];
        sots_tables::wavetable(wavetable);
    }
    uint32_t getWavetableSize() { return wavetableSize; }

    std::vector<float> scaleParams(const std::vector<float> aParams) // :567-576
    {
        std::vector<float> ret;
        for (size_t i = 0; i < paramMins.size() && i < aParams.size(); i++)
            ret.push_back(scaled(aParams[i], (uint32_t)i));
        return ret;
    }

    // 2-operator voice (:368-402): [mod freq, mod index, carrier freq, amplitude]
    SOTS_NO_CONTRACT_ATTR void synthesiseAudio(const std::vector<float> aParams, float *aAudioBuffer)
    {
        SOTS_NO_CONTRACT
        const float P0 = scaled(aParams[0], 0), P1 = scaled(aParams[1], 1), P2 = scaled(aParams[2], 2), P3 = scaled(aParams[3], 3);
        voice2(P0, P1, P2, P3, aAudioBuffer, false);
    }
    // three operators in series (:403-449)
    SOTS_NO_CONTRACT_ATTR void synthesiseAudioDoubleSeries(const std::vector<float> aParams, float *aAudioBuffer)
    {
        seriesChain(aParams, 3, aAudioBuffer);
    }
    // four operators in series: the same chain one stage longer (no reference counterpart)
    SOTS_NO_CONTRACT_ATTR void synthesiseAudioQuadSeries(const std::vector<float> aParams, float *aAudioBuffer)
    {
        seriesChain(aParams, 4, aAudioBuffer);
    }
    // three 2-operator voices averaged (:450-495); all three scale by entries 0..3
    SOTS_NO_CONTRACT_ATTR void synthesiseAudioTriple(const std::vector<float> aParams, float *aAudioBuffer)
    {
        SOTS_NO_CONTRACT
        std::vector<float> parts[3];
        for (int j = 0; j < 3; ++j) {
            parts[j].resize(audioLength);
            voice2(scaled(aParams[4 * j + 0], 0), scaled(aParams[4 * j + 1], 1), scaled(aParams[4 * j + 2], 2),
                   scaled(aParams[4 * j + 3], 3), parts[j].data(), false);
        }
        for (uint32_t i = 0; i < audioLength; ++i)
            aAudioBuffer[i] = (parts[0][i] + parts[1][i] + parts[2][i]) / 3.0;
    }

    // window -> forward real DFT (fp64) -> |X|/N/windowFactor for k < N/2 (:524-542)
    void calculateFFT(float *input, float *output)
    {
        sots_tables::target_spectrum(input, fftSize, fftWindow, fftWindowFactor, output);
    }
    void calculateJustFFT(float *input, float *output) { calculateFFT(input, output); } // :503-523

private:
    float scaled(float v, uint32_t i) const
    {
        const float lo = i < paramMins.size() ? paramMins[i] : 0.0f;
        const float hi = i < paramMaxs.size() ? paramMaxs[i] : 0.0f;
        return lo + v * (hi - lo);
    }
    float tableAt(float pos) const
    {
        // [fix] the reference can index past the table; clamp decided on the float so that it is
        // defined for every phase value (at or beyond the table length -> last entry, negative or NaN -> 0)
        if (pos >= (float)wavetableSize) return wavetable[wavetableSize - 1];
        if (!(pos > 0.0f)) return wavetable[0];
        return wavetable[(int32_t)pos];
    }
    void advance(float &pos, float by, bool bothEnds) const
    {
        pos += by;
        if (pos >= wavetableSize) pos -= wavetableSize;
        if (bothEnds && pos < 0.0f) pos += wavetableSize;
    }
    SOTS_NO_CONTRACT_ATTR void voice2(float P0, float P1, float P2, float P3, float *out, bool)
    {
        SOTS_NO_CONTRACT
        const float depth = P0 * P1, inc = w2srRatio * P0;
        float posMod = 0.0f, posCar = 0.0f; // [fix] phases restart per call
        for (uint32_t i = 0; i < audioLength; i++) {
            const float freq = tableAt(posMod) * depth + P2;
            advance(posMod, inc, false);
            out[i] = tableAt(posCar) * P3;
            advance(posCar, w2srRatio * freq, true);
        }
    }
    SOTS_NO_CONTRACT_ATTR void seriesChain(const std::vector<float> &aParams, int ops, float *out)
    {
        SOTS_NO_CONTRACT
        float P[8], depth[4], pos[4] = {0, 0, 0, 0};
        for (int g = 0; g < 2 * ops; ++g) P[g] = scaled(aParams[g], (uint32_t)g);
        for (int o = 0; o < ops; ++o) depth[o] = P[2 * o] * P[2 * o + 1];
        const float inc = w2srRatio * P[1];
        for (uint32_t i = 0; i < audioLength; i++) {
            float drive = tableAt(pos[0]) * depth[0] + P[3];
            advance(pos[0], inc, false);
            for (int o = 1; o + 1 < ops; ++o) {
                const float next = tableAt(pos[o]) * depth[o] + P[2 * o + 3];
                advance(pos[o], w2srRatio * drive, true);
                drive = next;
            }
            out[i] = tableAt(pos[ops - 1]) * depth[ops - 1];
            advance(pos[ops - 1], w2srRatio * drive, true);
        }
    }
    // recursive radix-2 decimation in time; n is a power of two
};

struct Evolutionary_Strategy_Arguments
{
    Population pop;
    uint32_t numGenerations = 100;
    std::vector<float> paramMin = {0.0, 0.0, 0.0, 0.0};
    std::vector<float> paramMax = {3520.0, 8.0, 3520.0, 1.0};
    uint32_t audioLengthLog2 = 10;
};

// Virtual base every backend derives from (Evolutionary_Strategy.hpp:591-696).
class Evolutionary_Strategy
{
public:
    uint32_t numGenerations;
    Population population;
    Objective objective;

    // ES constants (:600-628)
    const float mPI;
    const float alpha;
    const float oneOverAlpha;
    const float rootTwoOverPi;
    const float betaScale;
    const float beta;

    Evolutionary_Strategy(const uint32_t aNumGenerations, const uint32_t aNumParents, const uint32_t aNumOffspring,
                          const uint32_t aNumDimensions, const std::vector<float> aParamMin,
                          const std::vector<float> aParamMax, uint32_t aAudioLengthLog2)
        : numGenerations(aNumGenerations),
          objective(aNumParents + aNumOffspring, aNumDimensions, aParamMin, aParamMax, aAudioLengthLog2),
          mPI(3.14159265358979323846), alpha(1.4f), oneOverAlpha(1.f / alpha),
          rootTwoOverPi(sqrtf(2.f / (float)mPI)), betaScale(1.f / (float)aNumDimensions), beta(sqrtf(betaScale))
    {
        population.numParents = aNumParents;
        population.numOffspring = aNumOffspring;
        population.numDimensions = aNumDimensions;
        population.populationLength = aNumParents + aNumOffspring;
        population.populationSize = (aNumParents + aNumOffspring) * sizeof(float);
        storage_.assign((size_t)population.populationLength * population.rowLength(), 0.0f);
        population.data = storage_.data();
    }
    // the reference's default: 1024 parents + 2048 offspring, 4 dimensions, N = 1024 (:607-617)
    Evolutionary_Strategy() : Evolutionary_Strategy(100, 1024, 2048, 4, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, 10) {}
    virtual ~Evolutionary_Strategy() {}

    virtual void init() {}
    virtual void initTargetAudio() {}

    // population transfer; sizes are byte counts (:642-649)
    virtual void writePopulationData(void *, void *, uint32_t, void *, void *, uint32_t, void *, void *, uint32_t) {}
    virtual void readPopulationData(void *, void *, uint32_t, void *, void *, uint32_t, void *, void *, uint32_t) {}
    // synthesiser buffers (:652-659)
    virtual void writeSynthesizerData(void *, uint32_t, void *, void *, uint32_t) {}
    virtual void readSynthesizerData(void *, uint32_t, void *, void *, uint32_t) {}

    virtual void executeGeneration() {}
    virtual void executeAllGenerations() {}
    virtual void parameterMatchAudio(float * /*aTargetAudio*/, uint32_t /*aTargetAudioLength*/) {}

    virtual void readAudioFile() {}
    virtual void generateAudioFile() {}
    virtual void analyseAudio() {}
    virtual void setTargetFFT(float * /*aTargetAudio*/) {}
    virtual void printBest() {}

protected:
    std::vector<float> storage_; // backs population.data
};

#endif
