// Evolutionary_Strategy_HIP.hpp -- the MI355X backend: a drop-in for
// Evolutionary_Strategy_{OpenCL,CUDA,Vulkan} behind the Evolutionary_Strategy base class.
//
// All device work goes through the C-ABI of include/sots_hip.h (libsots_hip.so); this class
// is the thin C++ host side the reference's main.cpp drives:
//   construct from *_Arguments           (Evolutionary_Strategy_OpenCL.hpp:25-38,122-137)
//   parameterMatchAudio(audio, length)   (:572-610)  chunk loop, timers, CSV
//   readPopulationData(...)              (:417-430)
//   printBest()                          (:613-631)
// Stage timers keep the reference's names (:117) and feed the Benchmarker through
// addTimer(name, ms) with hipEvent-measured durations, like the Vulkan backend feeds its
// timestamp queries (Evolutionary_Strategy_Vulkan.hpp:1169-1210).  Errors, which the
// reference prints and ignores, throw std::runtime_error here (main.cpp:282 catches it).
#ifndef SOTS_EVOLUTIONARY_STRATEGY_HIP_HPP
#define SOTS_EVOLUTIONARY_STRATEGY_HIP_HPP

#include <algorithm>
#include <array>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sots_hip.h"
#include "Benchmarker.hpp"
#include "Evolutionary_Strategy.hpp"

struct Evolutionary_Strategy_HIP_Arguments
{
    // Generic Evolutionary Strategy arguments
    Evolutionary_Strategy_Arguments es_args;

    // HIP details (mirror of the OpenCL block)
    uint32_t workgroupX = 32; // recombination block, ocl_program.cl WRKGRPSIZE
    uint32_t workgroupY = 1;
    uint32_t workgroupZ = 1;
    uint32_t workgroupSize = workgroupX * workgroupY * workgroupZ;

    int32_t deviceOrdinal = 0;        // replaces deviceType / vendor-id matching
    uint64_t seed = 0x5EED0001ull;    // replaces the wall-clock seed
    uint32_t gidBase = 0;             // island offset of individual 0
    int32_t synthKind = -1;           // -1: derive from numDimensions (4, 6, 8, 12)
    bool fusedGenerations = true;     // executeAllGenerations uses the fused kernel loop
    // general.isBenchmarking (parameters.json:7, main.cpp:85).  true: every launch is bracketed by a
    // hipEvent pair and lands in the CSV under the reference's stage names (an event pair costs
    // about 3.5 us per kernel boundary: 170 vs 144 us per generation at pop = 65536).  false: the
    // un-instrumented loop - no events, one wall-clock "Total Audio Analysis Time" row and a
    // candidates-per-second line ("plus an un-instrumented mode").
    bool benchmarkStages = true;
    // true: sortPopulation orders all P rows every generation as the reference does; false: each generation places
    // the rows the next recombination reads and the rest of the order is produced when it is read (same results)
    bool fullSortEveryGeneration = false;
    // false (default): the synthesis arithmetic of the reference's CPU path (fp32 sample-rate ratio, Evolutionary_Strategy.hpp:203);
    // true: that of its OpenCL kernels (double ratio, fused multiply-adds, 3-op offset params[4]: ocl_program.cl:280-443) -
    // bit-identical audio to those kernels, through one plain kernel (enum sots_synth_arith); not for the 4-op voice
    bool deviceKernelArithmetic = false;
    // Island model inside this object (type.HIP.{numDevices,numElites,migrationInterval} in parameters.json;
    // the reference picks exactly one device, ...OpenCL.hpp:194-226).  es_args.pop describes ONE island; with
    // numDevices > 1 the object owns one island per device (devices[i], default deviceOrdinal + i), PRNG ids
    // gidBase + i * populationLength, and every migrationInterval generations the islands' best numElites
    // rows are all-gathered over RCCL and replace the tail of the other islands' parents.  Results are read
    // from the island holding the best individual; stage timers are island 0's.
    uint32_t numDevices = 1;
    std::vector<int32_t> devices;     // empty: deviceOrdinal, deviceOrdinal + 1, ...
    uint32_t numElites = 16;
    uint32_t migrationInterval = 1;
    bool overlapMigration = false;    // the all-gather runs underneath the next generation, rows arrive one exchange later
    bool verbose = true;
    std::string logDirectory = "";    // where hiplog(...).csv goes ("" = cwd)
};

class Evolutionary_Strategy_HIP : public Evolutionary_Strategy
{
private:
    static const uint8_t numKernels_ = 9;
    enum kernelNames_ { initPopulation = 0, recombinePopulation, mutatePopulation, synthesisePopulation, applyWindowPopulation, hipFFT, fitnessPopulation, sortPopulation, rotatePopulation };
    std::array<std::string, numKernels_> kernelNames_;

    sots_ctx *ctx_ = nullptr;       // the context results and timers are read from (island 0 / the best island of a group)
    sots_group *group_ = nullptr;   // owns the islands when numDevices > 1
    sots_config cfg_{};
    Evolutionary_Strategy_HIP_Arguments args_;

    uint32_t numChunks_ = 0;
    uint32_t chunkSize_ = 0;
    uint32_t targetAudioLength = 0;
    std::vector<float> targetFFT_;
    std::vector<std::vector<float>> bestPerChunk_;
    std::vector<float> launchScratch_;
    uint32_t bestIsland_ = 0;
    double candidatesPerSecond_ = 0.0;

    Benchmarker hipBenchmarker_;

    static int kindFromDims(uint32_t d)
    {
        switch (d) {
        case 4: return SOTS_SYNTH_2OP;
        case 6: return SOTS_SYNTH_3OP_SERIES;
        case 8: return SOTS_SYNTH_4OP_SERIES;
        case 12: return SOTS_SYNTH_TRIPLE_PAR;
        default: return -1;
        }
    }
    void check(int rc, const char *what) const
    {
        if (rc != SOTS_OK)
            throw std::runtime_error(std::string("Evolutionary_Strategy_HIP: ") + what + ": " + sots_last_error(ctx_));
    }
    void checkGroup(int rc, const char *what) const
    {
        if (rc != SOTS_OK)
            throw std::runtime_error(std::string("Evolutionary_Strategy_HIP: ") + what + ": " + sots_group_last_error(group_));
    }
    // the island whose best individual is the group's best becomes the one that is read
    void selectBestIsland()
    {
        if (!group_) return;
        uint32_t island = 0;
        float fitness = 0.0f;
        checkGroup(sots_group_best(group_, &island, &fitness), "sots_group_best");
        ctx_ = sots_group_island(group_, island);
        bestIsland_ = island;
    }
    static std::string logName(const Evolutionary_Strategy_HIP_Arguments &a)
    {
        const std::string dir = a.logDirectory.empty() ? "" : a.logDirectory + "/";
        return dir + "hiplog(pop=" + std::to_string(a.es_args.pop.populationLength) + "gens=" + std::to_string(a.es_args.numGenerations) +
               "audioBlockSize=" + std::to_string(1u << a.es_args.audioLengthLog2) + ").csv";
    }
    // hipEvent durations of the launches since the last harvest -> Benchmarker, ONE addTimer per
    // launch, so that the CSV's Average/Max/Min/difference columns are per launch as in the
    // reference (Benchmarker.hpp:33-72); launches beyond the library's per-stage sample store
    // (65536 between harvests) are added as one remainder
    void harvestStage(int stage, const std::string &name)
    {
        double ms = 0.0;
        uint64_t n = 0, got = 0;
        check(sots_stage_time_ms(ctx_, stage, &ms, &n), "sots_stage_time_ms");
        if (!n) return;
        launchScratch_.resize((size_t)std::min<uint64_t>(n, 65536));
        check(sots_stage_launch_times_ms(ctx_, stage, launchScratch_.data(), launchScratch_.size(), &got), "sots_stage_launch_times_ms");
        double listed = 0.0;
        for (uint64_t i = 0; i < got; ++i) {
            hipBenchmarker_.addTimer(name, launchScratch_[i]);
            listed += launchScratch_[i];
        }
        if (got < n) hipBenchmarker_.addTimer(name, ms - listed);
    }
    void harvestTimers()
    {
        if (!args_.benchmarkStages) return;
        if (group_) ctx_ = sots_group_island(group_, 0); // the instrumented island
        static const int stageOf[numKernels_] = {SOTS_STAGE_INIT, SOTS_STAGE_RECOMBINE, SOTS_STAGE_MUTATE, SOTS_STAGE_SYNTHESISE,
                                                 SOTS_STAGE_WINDOW, SOTS_STAGE_FFT, SOTS_STAGE_FITNESS, SOTS_STAGE_SORT, SOTS_STAGE_ROTATE};
        for (uint8_t k = 0; k < numKernels_; ++k) harvestStage(stageOf[k], kernelNames_[k]);
        static const std::pair<int, const char *> fused[] = {{SOTS_STAGE_FUSED_VARIATION, "recombine+mutatePopulation"},
                                                             {SOTS_STAGE_FUSED_SYNTH, "synthesise+applyWindowPopulation"},
                                                             {SOTS_STAGE_FUSED_SPECTRAL, "hipFFT+fitnessPopulation"}};
        for (const auto &f : fused) harvestStage(f.first, f.second);
        check(sots_timing_reset(ctx_), "sots_timing_reset");
    }

public:
    Evolutionary_Strategy_HIP(Evolutionary_Strategy_HIP_Arguments args)
        : Evolutionary_Strategy(args.es_args.numGenerations, args.es_args.pop.numParents, args.es_args.pop.numOffspring, args.es_args.pop.numDimensions, args.es_args.paramMin, args.es_args.paramMax, args.es_args.audioLengthLog2),
          kernelNames_({"initPopulation", "recombinePopulation", "mutatePopulation", "synthesisePopulation", "applyWindowPopulation", "hipFFT", "fitnessPopulation", "sortPopulation", "rotatePopulation"}),
          args_(args),
          hipBenchmarker_(logName(args), {"Test_Name", "Total_Time", "Average_Time", "Max_Time", "Min_Time", "Max_Difference", "Average_Difference"})
    {
        hipBenchmarker_.setVerbose(args.verbose);
        init();
    }
    ~Evolutionary_Strategy_HIP() override
    {
        if (group_) sots_group_destroy(group_); // owns its islands
        else if (ctx_) sots_destroy(ctx_);
        hipBenchmarker_.close();
    }
    Evolutionary_Strategy_HIP(const Evolutionary_Strategy_HIP &) = delete;
    Evolutionary_Strategy_HIP &operator=(const Evolutionary_Strategy_HIP &) = delete;

    sots_ctx *context() { return ctx_; }
    sots_group *group() { return group_; }
    uint32_t numIslands() const { return group_ ? sots_group_size(group_) : 1u; }
    uint32_t bestIsland() const { return bestIsland_; }
    Benchmarker &benchmarker() { return hipBenchmarker_; }
    const std::vector<std::vector<float>> &bestParametersPerChunk() const { return bestPerChunk_; }
    // candidates evaluated per second of the last parameterMatchAudio (population x generations x chunks / wall time)
    double candidatesPerSecond() const { return candidatesPerSecond_; }

    void init() override
    {
        if (ctx_) return;
        memset(&cfg_, 0, sizeof cfg_);
        cfg_.struct_size = sizeof cfg_;
        cfg_.num_parents = population.numParents;
        cfg_.num_offspring = population.numOffspring;
        cfg_.num_dimensions = population.numDimensions;
        cfg_.audio_length_log2 = objective.audioLengthLog2;
        cfg_.num_generations = numGenerations;
        const int kind = args_.synthKind >= 0 ? args_.synthKind : kindFromDims(population.numDimensions);
        if (kind < 0) throw std::runtime_error("Evolutionary_Strategy_HIP: numDimensions must be 4, 6, 8 or 12");
        cfg_.synth_kind = (uint32_t)kind;
        cfg_.workgroup_size = args_.workgroupX * args_.workgroupY * args_.workgroupZ;
        cfg_.device = args_.deviceOrdinal;
        cfg_.gid_base = args_.gidBase;
        cfg_.seed = args_.seed;
        for (size_t i = 0; i < SOTS_MAX_DIMS; ++i) {
            cfg_.param_min[i] = i < objective.paramMins.size() ? objective.paramMins[i] : 0.0f;
            cfg_.param_max[i] = i < objective.paramMaxs.size() ? objective.paramMaxs[i] : 0.0f;
        }
        if (args_.numDevices > 1) {
            std::vector<int32_t> devs = args_.devices;
            if (devs.empty())
                for (uint32_t i = 0; i < args_.numDevices; ++i) devs.push_back(args_.deviceOrdinal + (int32_t)i);
            if (devs.size() != args_.numDevices) throw std::runtime_error("Evolutionary_Strategy_HIP: devices must list numDevices entries");
            const int rc = sots_group_create(&cfg_, devs.data(), args_.numDevices, args_.numElites, args_.migrationInterval,
                                             args_.overlapMigration ? (uint32_t)SOTS_GROUP_OVERLAP : 0u, &group_);
            if (rc != SOTS_OK) throw std::runtime_error(std::string("Evolutionary_Strategy_HIP: sots_group_create: ") + sots_group_last_error(nullptr));
            ctx_ = sots_group_island(group_, 0);
        } else {
            const int rc = sots_create(&cfg_, &ctx_);
            if (rc != SOTS_OK) throw std::runtime_error(std::string("Evolutionary_Strategy_HIP: sots_create: ") + sots_last_error(nullptr));
        }
        targetFFT_.assign(objective.fftHalfSize, 0.0f);
        if (args_.fullSortEveryGeneration)
            for (uint32_t i = 0; i < numIslands(); ++i)
                check(sots_set_sort_mode(group_ ? sots_group_island(group_, i) : ctx_, SOTS_SORT_FULL), "sots_set_sort_mode");
        if (args_.deviceKernelArithmetic)
            for (uint32_t i = 0; i < numIslands(); ++i)
                check(sots_set_synth_arithmetic(group_ ? sots_group_island(group_, i) : ctx_, SOTS_ARITH_DEVICE_KERNELS), "sots_set_synth_arithmetic");
        check(sots_timing_enable(ctx_, args_.benchmarkStages ? 1 : 0), "sots_timing_enable");
    }
    void initTargetAudio() override {}

    // "Input" arrays address the current rotation half, "Output" arrays the other one
    // (the reference transfers both buffers whole, ...OpenCL.hpp:403-430).
    void writePopulationData(void *aInputPopulationValueData, void * /*aOutputPopulationValueData*/, uint32_t aPopulationValueSize, void *aInputPopulationStepData, void * /*aOutputPopulationStepData*/, uint32_t aPopulationStepSize, void *aInputPopulationFitnessData, void * /*aOutputPopulationFitnessData*/, uint32_t aPopulationFitnessSize) override
    {
        check(sots_write_population(ctx_, (const float *)aInputPopulationValueData, aPopulationValueSize, (const float *)aInputPopulationStepData, aPopulationStepSize, (const float *)aInputPopulationFitnessData, aPopulationFitnessSize), "writePopulationData");
    }
    void readPopulationData(void *aInputPopulationValueData, void *aOutputPopulationValueData, uint32_t aPopulationValueSize, void *aInputPopulationStepData, void *aOutputPopulationStepData, uint32_t aPopulationStepSize, void *aInputPopulationFitnessData, void *aOutputPopulationFitnessData, uint32_t aPopulationFitnessSize) override
    {
        selectBestIsland();
        check(sots_read_population(ctx_, (float *)aInputPopulationValueData, aPopulationValueSize, (float *)aInputPopulationStepData, aPopulationStepSize, (float *)aInputPopulationFitnessData, aPopulationFitnessSize), "readPopulationData");
        if (aOutputPopulationValueData || aOutputPopulationStepData || aOutputPopulationFitnessData)
            check(sots_read_population_other(ctx_, (float *)aOutputPopulationValueData, aPopulationValueSize, (float *)aOutputPopulationStepData, aPopulationStepSize, (float *)aOutputPopulationFitnessData, aPopulationFitnessSize), "readPopulationData");
        // keep the host-side AoS view in step with the device (main.cpp:244 reads es->population)
        const float *v = (const float *)aInputPopulationValueData, *s = (const float *)aInputPopulationStepData, *f = (const float *)aInputPopulationFitnessData;
        if (v && s && f)
            for (uint32_t i = 0; i != population.populationLength; ++i) {
                for (uint32_t j = 0; j != population.numDimensions; ++j) {
                    *population.getValue(i, j) = v[i * population.numDimensions + j];
                    *population.getStep(i, j) = s[i * population.numDimensions + j];
                }
                *population.getFitness(i) = f[i];
            }
    }

    // aInputFFTSize: bytes of the device spectrum buffer P*(N+8)*4; the target gets aInputFFTSize/2 in the
    // reference (...OpenCL.hpp:443,452) -- here the target is always N/2 floats.
    void writeSynthesizerData(void *aOutputAudioBuffer, uint32_t aOutputAudioSize, void *aInputFFTDataBuffer, void *aInputFFTTargetBuffer, uint32_t aInputFFTSize) override
    {
        check(sots_write_synth(ctx_, (const float *)aOutputAudioBuffer, aOutputAudioSize, (const float *)aInputFFTDataBuffer, aInputFFTSize), "writeSynthesizerData");
        if (aInputFFTTargetBuffer) setTargetFFT((float *)aInputFFTTargetBuffer);
    }
    void readSynthesizerData(void *aOutputAudioBuffer, uint32_t aOutputAudioSize, void *aInputFFTDataBuffer, void *aInputFFTTargetBuffer, uint32_t aInputFFTSize) override
    {
        check(sots_read_synth(ctx_, (float *)aOutputAudioBuffer, aOutputAudioSize, (float *)aInputFFTDataBuffer, aInputFFTSize, (float *)aInputFFTTargetBuffer, aInputFFTTargetBuffer ? objective.fftHalfSize * sizeof(float) : 0), "readSynthesizerData");
    }

    void initPopulationHIP(uint32_t aChunk = 0)
    {
        if (group_) checkGroup(sots_group_init_population(group_, aChunk), "initPopulation");
        else check(sots_init_population(ctx_, aChunk), "initPopulation");
    }

    // the eight per-generation stages, each as its own launch sequence (...OpenCL.hpp:471-541); a group of
    // islands runs the fused loop (its exchange sits between generations)
    void executeGeneration() override
    {
        if (group_) checkGroup(sots_group_execute_generations(group_, 1), "executeGeneration");
        else check(sots_execute_generation(ctx_), "executeGeneration");
    }
    void executeAllGenerations() override
    {
        if (group_) {
            checkGroup(sots_group_execute_generations(group_, numGenerations), "executeAllGenerations");
        } else if (args_.fusedGenerations) {
            check(sots_execute_generations(ctx_, numGenerations), "executeAllGenerations");
        } else {
            for (uint32_t i = 0; i != numGenerations; ++i) executeGeneration();
        }
    }

    void setTargetAudio(float *aTargetAudio, uint32_t aTargetAudioLength)
    {
        // host: double window -> fp64 DFT -> magnitudes (Objective::calculateFFT), then H2D (...OpenCL.hpp:563-570)
        targetAudioLength = aTargetAudioLength;
        objective.calculateFFT(aTargetAudio, targetFFT_.data());
        if (group_) checkGroup(sots_group_set_target_spectrum(group_, targetFFT_.data(), objective.fftHalfSize), "setTargetAudio");
        else check(sots_set_target_spectrum(ctx_, targetFFT_.data(), objective.fftHalfSize), "setTargetAudio");
    }
    void setTargetFFT(float *aTargetFFT) override
    {
        std::copy(aTargetFFT, aTargetFFT + objective.fftHalfSize, targetFFT_.begin());
        if (group_) checkGroup(sots_group_set_target_spectrum(group_, targetFFT_.data(), objective.fftHalfSize), "setTargetFFT");
        else check(sots_set_target_spectrum(ctx_, targetFFT_.data(), objective.fftHalfSize), "setTargetFFT");
    }

    void parameterMatchAudio(float *aTargetAudio, uint32_t aTargetAudioLength) override
    {
        // every N-sample chunk is matched from a fresh population (...OpenCL.hpp:572-610)
        chunkSize_ = objective.audioLength;
        numChunks_ = aTargetAudioLength / chunkSize_;
        bestPerChunk_.clear();

        hipBenchmarker_.startTimer("Total Audio Analysis Time");
        for (uint32_t i = 0; i < numChunks_; i++) {
            setTargetAudio(&aTargetAudio[chunkSize_ * i], chunkSize_);
            initPopulationHIP(i);
            executeAllGenerations();
            if (group_) checkGroup(sots_group_synchronize(group_), "synchronize");
            else check(sots_synchronize(ctx_), "synchronize");
            if (args_.verbose) printf("Audio chunk %u evaluated:\n", i);
            printBest();
            harvestTimers();
        }
        hipBenchmarker_.pauseTimer("Total Audio Analysis Time");
        const double totalMs = hipBenchmarker_.totalMs("Total Audio Analysis Time");
        candidatesPerSecond_ = totalMs > 0.0 ? (double)population.populationLength * numIslands() * numGenerations * numChunks_ / (totalMs * 1e-3) : 0.0;
        if (args_.verbose) printf("Candidates evaluated per second: %.6g\n", candidatesPerSecond_);

        for (uint8_t k = 1; k < numKernels_; ++k)
            if (hipBenchmarker_.count(kernelNames_[k])) hipBenchmarker_.elapsedTimer(kernelNames_[k]);
        for (const char *name : {"recombine+mutatePopulation", "synthesise+applyWindowPopulation", "hipFFT+fitnessPopulation"})
            if (hipBenchmarker_.count(name)) hipBenchmarker_.elapsedTimer(name);
        hipBenchmarker_.elapsedTimer("Total Audio Analysis Time");
    }

    // rotation-aware (the reference reads offset 0 whatever the rotation index, ...OpenCL.hpp:612-631)
    void printBest() override
    {
        selectBestIsland();
        const uint32_t d = population.numDimensions;
        std::vector<float> v((size_t)population.populationLength * d), f(population.populationLength);
        check(sots_read_population(ctx_, v.data(), v.size() * sizeof(float), nullptr, 0, f.data(), f.size() * sizeof(float)), "printBest");
        std::vector<float> best(v.begin(), v.begin() + d);
        bestPerChunk_.push_back(best);
        if (!args_.verbose) return;
        const std::vector<float> scaled = objective.scaleParams(best);
        printf("Best parameters found:\n");
        for (uint32_t j = 0; j < d && j < scaled.size(); ++j) printf(" p%u = %f\n", j, scaled[j]);
        printf("Best fitness: %g\n\n", f[0]);
    }
};

#endif
