// host_cpu_test.cpp -- CPU-only checks of the C++ host layer (no GPU, no libsots_hip):
// Objective (wavetable, window, the four voices, calculateFFT, scaleParams), Population's stable
// sort, Benchmarker/CSV_Logger.  Prints raw float arrays that tests/test_host_cpu.py compares
// with the CPU oracle.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>

#include "Benchmarker.hpp"
#include "Evolutionary_Strategy.hpp"

static void dump(const char *path, const float *p, size_t n)
{
    std::ofstream out(path, std::ios::binary);
    out.write(reinterpret_cast<const char *>(p), (std::streamsize)(n * sizeof(float)));
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : ".";
    const uint32_t log2n = 10, n = 1u << log2n;

    // ---- Objective: four voices on the unit-cube parameters the tests also give the oracle ----
    {
        Objective o(8, 4, {0, 0, 0, 0}, {3520.0f, 8.0f, 3520.0f, 1.0f}, log2n);
        std::vector<float> a(n), m(n / 2);
        o.synthesiseAudio({0.411931818f, 0.375f, 0.0568181818f, 1.0f}, a.data());
        dump((dir + "/voice2.f32").c_str(), a.data(), n);
        o.calculateFFT(a.data(), m.data());
        dump((dir + "/voice2_mag.f32").c_str(), m.data(), n / 2);
        o.synthesiseAudio({0.9f, 1.0f, 0.02f, 0.5f}, a.data()); // a second call must not depend on the first
        dump((dir + "/voice2b.f32").c_str(), a.data(), n);
        o.synthesiseAudioTriple({0.41f, 0.375f, 0.057f, 1.0f, 0.2f, 0.5f, 0.11f, 0.7f, 0.6f, 0.1f, 0.3f, 0.4f}, a.data());
        dump((dir + "/voice12.f32").c_str(), a.data(), n);
        dump((dir + "/wavetable.f32").c_str(), o.wavetable, o.wavetableSize);
        std::vector<float> w(n);
        for (uint32_t i = 0; i < n; ++i) w[i] = (float)o.fftWindow[i];
        dump((dir + "/window.f32").c_str(), w.data(), n);
        const std::vector<float> sc = o.scaleParams({0.5f, 0.25f, 1.0f, 0.0f});
        printf("scale %.9g %.9g %.9g %.9g wf %.9g outsize %u\n", sc[0], sc[1], sc[2], sc[3], o.fftWindowFactor, o.fftOutSize);
    }
    {
        Objective o(8, 6, {0, 0, 0, 0, 0, 0}, {3520.0f, 8.0f, 3520.0f, 8.0f, 3520.0f, 8.0f}, log2n);
        std::vector<float> a(n);
        o.synthesiseAudioDoubleSeries({3078 / 3520.0f, 2.0f / 8.0f, 3015 / 3520.0f, 1.5f / 8.0f, 3141 / 3520.0f, 1.0f / 8.0f}, a.data());
        dump((dir + "/voice6.f32").c_str(), a.data(), n);
    }
    {
        Objective o(8, 8, std::vector<float>(8, 0.0f), {3520.0f, 8.0f, 3520.0f, 8.0f, 3520.0f, 8.0f, 3520.0f, 8.0f}, log2n);
        std::vector<float> a(n);
        o.synthesiseAudioQuadSeries({0.3f, 0.25f, 0.85f, 0.19f, 0.89f, 0.125f, 0.5f, 0.1f}, a.data());
        dump((dir + "/voice8.f32").c_str(), a.data(), n);
    }

    // ---- Population: stable ascending sort, NaN last ----
    {
        Evolutionary_Strategy es(1, 4, 4, 2, {0, 0}, {1, 1}, 9);
        const float fit[8] = {3.0f, 1.0f, NAN, 1.0f, 0.0f, -0.0f, 2.0f, 1.0f};
        for (uint32_t i = 0; i < 8; ++i) {
            *es.population.getValue(i, 0) = (float)i;
            *es.population.getStep(i, 1) = 10.0f + i;
            *es.population.getFitness(i) = fit[i];
        }
        es.population.bubbleSortPopulation();
        printf("order");
        for (uint32_t i = 0; i < 8; ++i) printf(" %g", *es.population.getValue(i, 0));
        printf(" consts %.9g %.9g %.9g %.9g %.9g\n", es.alpha, es.oneOverAlpha, es.rootTwoOverPi, es.betaScale, es.beta);
    }

    // ---- Benchmarker / CSV_Logger ----
    {
        const std::string csv = dir + "/bench.csv";
        Benchmarker b(csv, {"Test_Name", "Total_Time", "Average_Time", "Max_Time", "Min_Time", "Max_Difference", "Average_Difference"});
        b.setVerbose(false);
        b.addTimer("stageA", 2.0);
        b.addTimer("stageA", 4.0);
        b.addTimer("stageA", 3.0);
        b.addTimer("once", 7.5);
        b.startTimer("wall");
        b.pauseTimer("wall");
        printf("counts %u %u total %.3f\n", b.count("stageA"), b.count("missing"), b.totalMs("stageA"));
        b.elapsedTimer("stageA");
        b.elapsedTimer("once"); // a single sample still produces a full 7-field row
        b.elapsedTimer("wall");
        printf("after %u\n", b.count("stageA"));
        b.close();
        CSV_Logger l(dir + "/raw.csv", {"a", "b"});
        printf("reject %d accept %d\n", (int)l.addRecord({"1"}), (int)l.addRecord({"1", "2"}));
        l.close();
    }
    return 0;
}
