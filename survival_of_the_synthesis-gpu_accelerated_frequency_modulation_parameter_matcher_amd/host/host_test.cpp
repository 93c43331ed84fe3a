// host_test.cpp -- drives Evolutionary_Strategy_HIP only through the base-class interface,
// the way the reference's main.cpp does (main.cpp:105-163 construct, :233 match, :241 read
// back, :244/:272 public members, :280 print), and prints one JSON object for pytest.
#include <cmath>
#include <cstdio>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>

#include "Evolutionary_Strategy_HIP.hpp"

static Evolutionary_Strategy_HIP_Arguments make_args(uint32_t parents, uint32_t offspring, uint32_t dims, uint32_t log2n,
                                                     uint32_t gens, const std::vector<float> &pmax, const std::string &dir)
{
    Evolutionary_Strategy_HIP_Arguments args;
    args.es_args.pop.numParents = parents;
    args.es_args.pop.numOffspring = offspring;
    args.es_args.pop.numDimensions = dims;
    args.es_args.pop.populationLength = parents + offspring;
    args.es_args.pop.populationSize = (parents + offspring) * sizeof(float);
    args.es_args.numGenerations = gens;
    args.es_args.paramMin = std::vector<float>(dims, 0.0f);
    args.es_args.paramMax = pmax;
    args.es_args.audioLengthLog2 = log2n;
    args.workgroupX = 32;
    args.verbose = false;
    args.logDirectory = dir;
    return args;
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : ".";
    try {
        const std::vector<float> pmax = {3520.0f, 8.0f, 3520.0f, 1.0f};
        const uint32_t parents = 2048, offspring = 6144, gens = 40, log2n = 10, n = 1u << log2n;
        auto args = make_args(parents, offspring, 4, log2n, gens, pmax, dir);
        std::unique_ptr<Evolutionary_Strategy> es(new Evolutionary_Strategy_HIP(args));

        // two audio chunks with different true parameters
        const std::vector<float> truthA = {1450.0f / 3520.0f, 3.0f / 8.0f, 200.0f / 3520.0f, 1.0f};
        const std::vector<float> truthB = {0.25f, 0.5f, 0.125f, 0.75f};
        std::vector<float> target(2 * n);
        es->objective.synthesiseAudio(truthA, target.data());
        es->objective.synthesiseAudio(truthB, target.data() + n);

        es->parameterMatchAudio(target.data(), 2 * n);

        const uint32_t P = es->population.populationLength, D = es->population.numDimensions;
        std::vector<float> v(P * D), s(P * D), f(P), v2(P * D), s2(P * D), f2(P);
        es->readPopulationData(v.data(), v2.data(), P * D * sizeof(float), s.data(), s2.data(), P * D * sizeof(float), f.data(),
                               f2.data(), P * sizeof(float));
        bool sorted = true;
        for (uint32_t i = 1; i < P; ++i) sorted = sorted && !(f[i] < f[i - 1]);
        const bool aos_ok = *es->population.getFitness(0) == f[0] && *es->population.getValue(0, 1) == v[1];

        // fitness of the first chunk's best parameters, re-evaluated on the host Objective
        auto *hip = static_cast<Evolutionary_Strategy_HIP *>(es.get());
        const auto &best = hip->bestParametersPerChunk();
        std::vector<float> audio(n), magA(n / 2), magB(n / 2);
        es->objective.calculateFFT(target.data(), magA.data());
        es->objective.synthesiseAudio(best.at(0), audio.data());
        es->objective.calculateFFT(audio.data(), magB.data());
        double host_fit = 0.0;
        for (uint32_t k = 0; k < n / 2; ++k) host_fit += (double)(magB[k] - magA[k]) * (magB[k] - magA[k]);
        const size_t chunks = best.size(); // `best` lives inside *es, which is destroyed below

        // staged loop must give the same population as the fused loop
        auto argsStaged = args;
        argsStaged.fusedGenerations = false;
        argsStaged.es_args.numGenerations = 3;
        args.es_args.numGenerations = 3;
        Evolutionary_Strategy_HIP a(args), b(argsStaged);
        a.parameterMatchAudio(target.data(), n);
        b.parameterMatchAudio(target.data(), n);
        std::vector<float> va(P * D), vb(P * D), fa(P), fb(P);
        a.readPopulationData(va.data(), nullptr, P * D * sizeof(float), nullptr, nullptr, 0, fa.data(), nullptr, P * sizeof(float));
        b.readPopulationData(vb.data(), nullptr, P * D * sizeof(float), nullptr, nullptr, 0, fb.data(), nullptr, P * sizeof(float));
        const bool same = va == vb && fa == fb;
        const uint32_t staged_fft_calls = b.benchmarker().count("hipFFT"); // reset by elapsedTimer -> 0

        // CSV written by the Benchmarker
        const std::string csv = dir + "/hiplog(pop=" + std::to_string(P) + "gens=" + std::to_string(gens) + "audioBlockSize=" + std::to_string(n) + ").csv";
        es.reset(); // closes the log
        std::ifstream in(csv);
        std::string line, header;
        std::getline(in, header);
        int rows = 0;
        bool has_total = false;
        while (std::getline(in, line)) {
            ++rows;
            if (line.rfind("Total Audio Analysis Time", 0) == 0) has_total = true;
        }

        // un-instrumented mode (general.isBenchmarking = false): no stage rows, the total row, a rate
        int quiet_rows = 0, quiet_stage_rows = 0;
        bool quiet_total = false;
        double quiet_rate = 0.0;
        {
            auto quiet = make_args(parents, offspring, 4, log2n, 5, pmax, dir);
            quiet.benchmarkStages = false;
            {
                Evolutionary_Strategy_HIP q(quiet);
                q.parameterMatchAudio(target.data(), n);
                quiet_rate = q.candidatesPerSecond();
            }
            std::ifstream qin(dir + "/hiplog(pop=" + std::to_string(P) + "gens=5audioBlockSize=" + std::to_string(n) + ").csv");
            std::string ql;
            std::getline(qin, ql);
            while (std::getline(qin, ql)) {
                ++quiet_rows;
                if (ql.rfind("Total Audio Analysis Time", 0) == 0) quiet_total = true;
                else ++quiet_stage_rows;
            }
        }
        // instrumented: the fused-loop rows are per launch (Average = Total / launches)
        double avg_sort_ms = 0.0, total_sort_ms = 0.0;
        {
            std::ifstream cin2(csv);
            std::string l2;
            while (std::getline(cin2, l2))
                if (l2.rfind("sortPopulation,", 0) == 0) sscanf(l2.c_str(), "sortPopulation,%lf,%lf", &total_sort_ms, &avg_sort_ms);
        }

        // island model inside the object: two islands (sharing device 0 here), elites exchanged every generation
        double group_best = 0.0, group_rate = 0.0;
        uint32_t group_islands = 0;
        bool group_sorted = true;
        {
            auto ga = make_args(parents, offspring, 4, log2n, 30, pmax, dir);
            ga.numDevices = 2;
            ga.devices = {0, 0};
            ga.numElites = 16;
            ga.benchmarkStages = false;
            Evolutionary_Strategy_HIP gs(ga);
            gs.parameterMatchAudio(target.data(), n);
            group_islands = gs.numIslands();
            group_rate = gs.candidatesPerSecond();
            std::vector<float> gv(P * D), gf(P);
            gs.readPopulationData(gv.data(), nullptr, P * D * sizeof(float), nullptr, nullptr, 0, gf.data(), nullptr, P * sizeof(float));
            group_best = gf[0];
            for (uint32_t i = parents; i + 1 < P; ++i) group_sorted = group_sorted && !(gf[i + 1] < gf[i]);
        }

        // a bad configuration must throw, not limp on
        bool threw = false;
        try {
            auto bad = make_args(30, 31, 4, log2n, 1, pmax, dir); // 61 is not a multiple of the block
            Evolutionary_Strategy_HIP x(bad);
        } catch (const std::runtime_error &) {
            threw = true;
        }

        printf("{\"sorted\": %s, \"aos_ok\": %s, \"best_fitness_last_chunk\": %.9g, \"host_fitness_chunk0\": %.9g, "
               "\"chunks\": %zu, \"fused_equals_staged\": %s, \"csv_header\": \"%s\", \"csv_rows\": %d, \"csv_has_total\": %s, "
               "\"bad_config_throws\": %s, \"staged_fft_pending\": %u, \"quiet_rows\": %d, \"quiet_stage_rows\": %d, "
               "\"quiet_has_total\": %s, \"quiet_candidates_per_s\": %.6g, \"sort_total_ms\": %.9g, \"sort_avg_ms\": %.9g, "
               "\"group_islands\": %u, \"group_best\": %.9g, \"group_candidates_per_s\": %.6g, \"group_tail_sorted\": %s}\n",
               sorted ? "true" : "false", aos_ok ? "true" : "false", f[0], host_fit, chunks, same ? "true" : "false",
               header.c_str(), rows, has_total ? "true" : "false", threw ? "true" : "false", staged_fft_calls, quiet_rows,
               quiet_stage_rows, quiet_total ? "true" : "false", quiet_rate, total_sort_ms, avg_sort_ms, group_islands, group_best,
               group_rate, group_sorted ? "true" : "false");
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "host_test failed: %s\n", e.what());
        return 1;
    }
}
