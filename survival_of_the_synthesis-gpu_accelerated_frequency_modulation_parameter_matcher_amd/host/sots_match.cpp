// sots_match.cpp -- command-line driver with the reference's interface (main.cpp:25-305):
//     sots_match -j parameters.json
// Reads the reference's parameters.json schema (general / audio / evolutionary / type), with
// "type": {"implementation": "HIP", "HIP": {"workgroupSize", "device", "seed", "synth", "numDevices", "numElites",
// "migrationInterval", "overlapMigration", "devices", "fullSortEveryGeneration", "deviceKernelArithmetic"}},
// builds the target from "params" (synthesised) or "audio" (a mono WAV file), matches every
// N-sample chunk with Evolutionary_Strategy_HIP, writes inputGenerated.wav and the
// outputAudioPath rendering of the best match, and prints the best parameters.
//
// The JSON reader and the WAV reader/writer are small built-ins: the reference's
// dependencies (nlohmann json, libsndfile, AudioFile) are not vendored and not needed.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "Evolutionary_Strategy_HIP.hpp"
#include "Wav_IO.hpp"

// ---------------------------------------------------------------------------------------
// minimal JSON (objects, arrays, numbers, strings, true/false/null)
// ---------------------------------------------------------------------------------------
struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::map<std::string, Json> obj;

    const Json &operator[](const std::string &k) const
    {
        auto it = obj.find(k);
        if (kind != Object || it == obj.end()) throw std::runtime_error("parameters.json: missing key \"" + k + "\"");
        return it->second;
    }
    bool has(const std::string &k) const { return kind == Object && obj.count(k); }
    double number() const
    {
        if (kind != Number) throw std::runtime_error("parameters.json: number expected");
        return num;
    }
    std::vector<float> floats() const
    {
        std::vector<float> out;
        for (const Json &e : arr) out.push_back((float)e.number());
        return out;
    }
};

class JsonParser
{
    const std::string &s_;
    size_t i_ = 0;
    void ws()
    {
        while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\n' || s_[i_] == '\t' || s_[i_] == '\r')) ++i_;
    }
    [[noreturn]] void bad(const char *what) { throw std::runtime_error(std::string("parameters.json: ") + what + " at offset " + std::to_string(i_)); }
    std::string string()
    {
        std::string out;
        ++i_;
        while (i_ < s_.size() && s_[i_] != '"') {
            if (s_[i_] == '\\' && i_ + 1 < s_.size()) ++i_;
            out.push_back(s_[i_++]);
        }
        if (i_ >= s_.size()) bad("unterminated string");
        ++i_;
        return out;
    }

public:
    explicit JsonParser(const std::string &s) : s_(s) {}
    Json value()
    {
        ws();
        if (i_ >= s_.size()) bad("unexpected end");
        Json j;
        const char c = s_[i_];
        if (c == '{') {
            j.kind = Json::Object;
            ++i_;
            ws();
            if (s_[i_] == '}') { ++i_; return j; }
            for (;;) {
                ws();
                if (s_[i_] != '"') bad("key expected");
                const std::string k = string();
                ws();
                if (s_[i_++] != ':') bad("':' expected");
                j.obj[k] = value();
                ws();
                if (s_[i_] == ',') { ++i_; continue; }
                if (s_[i_] == '}') { ++i_; return j; }
                bad("',' or '}' expected");
            }
        }
        if (c == '[') {
            j.kind = Json::Array;
            ++i_;
            ws();
            if (s_[i_] == ']') { ++i_; return j; }
            for (;;) {
                j.arr.push_back(value());
                ws();
                if (s_[i_] == ',') { ++i_; continue; }
                if (s_[i_] == ']') { ++i_; return j; }
                bad("',' or ']' expected");
            }
        }
        if (c == '"') { j.kind = Json::String; j.str = string(); return j; }
        if (s_.compare(i_, 4, "true") == 0) { j.kind = Json::Bool; j.b = true; i_ += 4; return j; }
        if (s_.compare(i_, 5, "false") == 0) { j.kind = Json::Bool; j.b = false; i_ += 5; return j; }
        if (s_.compare(i_, 4, "null") == 0) { i_ += 4; return j; }
        char *end = nullptr;
        j.num = strtod(s_.c_str() + i_, &end);
        if (end == s_.c_str() + i_) bad("value expected");
        j.kind = Json::Number;
        i_ = (size_t)(end - s_.c_str());
        return j;
    }
};

static void show_usage(const std::string &name)
{
    std::cerr << "Usage: " << name << " -j <parameters.json>\n"
              << "  -h, --help        this text\n"
              << "  -j, --json PATH   configuration in the reference's parameters.json schema;\n"
              << "                    type.implementation must be \"HIP\" (or any value with --force-hip)\n"
              << "  --force-hip       run the HIP backend whatever type.implementation says\n";
}

int main(int argc, char *argv[])
{
    try {
        std::string jsonPath;
        bool forceHip = false;
        for (int i = 1; i < argc; ++i) {
            const std::string arg = argv[i];
            if (arg == "-h" || arg == "--help") { show_usage(argv[0]); return 0; }
            else if ((arg == "-j" || arg == "--json") && i + 1 < argc) jsonPath = argv[++i];
            else if (arg == "--force-hip") forceHip = true;
        }
        if (jsonPath.empty()) { show_usage(argv[0]); return 1; }
        std::ifstream ifs(jsonPath);
        if (!ifs) throw std::runtime_error("cannot open " + jsonPath);
        std::stringstream buf;
        buf << ifs.rdbuf();
        const std::string text = buf.str();
        const Json j = JsonParser(text).value();

        const std::string implementation = j["type"]["implementation"].str;
        if (implementation != "HIP" && !forceHip)
            throw std::runtime_error("type.implementation is \"" + implementation + "\": this build carries the HIP backend only (use --force-hip)");
        const std::string outputAudioPath = j["general"]["outputAudioPath"].str;
        const bool verbose = !j["general"].has("isDebug") || j["general"]["isDebug"].b;
        const uint32_t audioLengthLog2 = (uint32_t)j["audio"]["audioLengthLog2"].number();
        const Json &evo = j["evolutionary"];

        Evolutionary_Strategy_HIP_Arguments args;
        args.es_args.pop.numParents = (uint32_t)evo["numParents"].number();
        args.es_args.pop.numOffspring = (uint32_t)evo["numOffspring"].number();
        args.es_args.pop.numDimensions = (uint32_t)evo["numDimensions"].number();
        args.es_args.pop.populationLength = args.es_args.pop.numParents + args.es_args.pop.numOffspring;
        args.es_args.pop.populationSize = args.es_args.pop.populationLength * sizeof(float);
        args.es_args.numGenerations = (uint32_t)evo["numGenerations"].number();
        args.es_args.paramMin = evo["paramMins"].floats();
        args.es_args.paramMax = evo["paramMaxs"].floats();
        args.es_args.audioLengthLog2 = audioLengthLog2;
        args.verbose = verbose;
        // general.isBenchmarking (parameters.json:7, main.cpp:85): per-stage hipEvent timing on / off
        if (j["general"].has("isBenchmarking")) args.benchmarkStages = j["general"]["isBenchmarking"].b;
        if (j["type"].has("HIP")) {
            const Json &h = j["type"]["HIP"];
            if (h.has("workgroupSize")) args.workgroupX = (uint32_t)h["workgroupSize"].number();
            if (h.has("device")) args.deviceOrdinal = (int32_t)h["device"].number();
            if (h.has("seed")) args.seed = (uint64_t)h["seed"].number();
            // island model: one island of numParents + numOffspring per device, elites all-gathered over RCCL
            if (h.has("numDevices")) args.numDevices = (uint32_t)h["numDevices"].number();
            if (h.has("numElites")) args.numElites = (uint32_t)h["numElites"].number();
            if (h.has("migrationInterval")) args.migrationInterval = (uint32_t)h["migrationInterval"].number();
            if (h.has("overlapMigration")) args.overlapMigration = h["overlapMigration"].b;
            if (h.has("fullSortEveryGeneration")) args.fullSortEveryGeneration = h["fullSortEveryGeneration"].b;
            if (h.has("deviceKernelArithmetic")) args.deviceKernelArithmetic = h["deviceKernelArithmetic"].b;
            if (h.has("devices"))
                for (const Json &dv : h["devices"].arr) args.devices.push_back((int32_t)dv.number());
            if (h.has("synth")) {
                const std::string s = h["synth"].str;
                args.synthKind = s == "2op" ? SOTS_SYNTH_2OP : s == "3op_series" ? SOTS_SYNTH_3OP_SERIES
                               : s == "triple_parallel" ? SOTS_SYNTH_TRIPLE_PAR : s == "4op_series" ? SOTS_SYNTH_4OP_SERIES : -1;
            }
        }
        const uint32_t D = args.es_args.pop.numDimensions;
        std::unique_ptr<Evolutionary_Strategy> es(new Evolutionary_Strategy_HIP(args));
        auto synthesise = [&](Objective &obj, const std::vector<float> &p, float *out) {
            if (D == 4) obj.synthesiseAudio(p, out);
            else if (D == 6) obj.synthesiseAudioDoubleSeries(p, out);
            else if (D == 8) obj.synthesiseAudioQuadSeries(p, out);
            else obj.synthesiseAudioTriple(p, out);
        };

        // target (main.cpp:198-228): a WAV file, or audio generated from known parameters
        const uint32_t N = 1u << audioLengthLog2;
        std::vector<float> targetAudio;
        if (j["type"]["input"].str == "audio") {
            targetAudio = readAudioFile(j["type"]["audio"].str);
            if (targetAudio.size() < N) targetAudio.resize(N, 0.0f);
        } else {
            const std::vector<float> raw = j["type"]["params"].floats();
            if (raw.size() < D) throw std::runtime_error("type.params needs numDimensions entries");
            std::vector<float> unit(D);
            for (uint32_t i = 0; i < D; ++i) {
                const uint32_t s = D == 12 ? (i & 3u) : i;
                const float lo = args.es_args.paramMin[s], hi = args.es_args.paramMax[s];
                unit[i] = (raw[i] - lo) / (hi - lo);
            }
            targetAudio.resize(N);
            synthesise(es->objective, unit, targetAudio.data());
            outputAudioFile("inputGenerated.wav", targetAudio.data(), N);
        }

        const auto start = std::chrono::steady_clock::now();
        es->parameterMatchAudio(targetAudio.data(), (uint32_t)targetAudio.size());
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        std::cout << "Total time to complete: " << secs << "s" << std::endl;
        const double evaluated = (double)es->population.populationLength * args.numDevices * es->numGenerations * (targetAudio.size() / N);
        std::cout << "Candidates evaluated per second: " << evaluated / secs << std::endl;

        const uint32_t P = es->population.populationLength;
        std::vector<float> v(P * D), s(P * D), f(P);
        es->readPopulationData(v.data(), nullptr, P * D * sizeof(float), s.data(), nullptr, P * D * sizeof(float), f.data(), nullptr, P * sizeof(float));
        std::vector<float> best(v.begin(), v.begin() + D);

        // render 2^14 samples of the best match (main.cpp:270-275)
        Objective render(P, D, args.es_args.paramMin, args.es_args.paramMax, 14);
        std::vector<float> audio(1u << 14);
        synthesise(render, best, audio.data());
        outputAudioFile(outputAudioPath, audio.data(), 1u << 14);

        printf("Overall best parameters found\n Fitness = %g\n", f[0]);
        const std::vector<float> scaled = es->objective.scaleParams(best);
        for (uint32_t i = 0; i < scaled.size(); ++i) printf(" p%u = %f\n", i, scaled[i]);
        return EXIT_SUCCESS;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return EXIT_FAILURE;
    }
}
