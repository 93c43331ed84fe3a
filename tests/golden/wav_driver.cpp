// Writes the signal of wav_samples.inc as a 24-bit 44.1 kHz mono WAV to argv[1]:
//   -DSOTS_WAV_REFERENCE: through the REFERENCE's AudioFile exactly as its outputAudioFile does (main.cpp:337-366);
//     built by tests/golden/make_wav_golden.sh with /root/reference/AudioFile.cpp (build container only)
//   otherwise: through host/Wav_IO.hpp, then reads argv[2] (the golden file) back and prints its samples' sum of
//     absolute differences to k / 8388608 of the bytes (tests/test_host_cpu.py)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
#ifdef SOTS_WAV_REFERENCE
#include "AudioFile.h"
#else
#include "Wav_IO.hpp"
#endif

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
#include "wav_samples.inc"
#ifdef SOTS_WAV_REFERENCE
    AudioFile<float> audioFile;
    AudioFile<float>::AudioBuffer buffer;
    buffer.resize(1);
    buffer[0].resize(n);
    audioFile.setBitDepth(24);
    audioFile.setSampleRate(44100);
    for (int k = 0; k != n; ++k) buffer[0][k] = (float)buf[k];
    audioFile.setAudioBuffer(buffer);
    return audioFile.save(argv[1]) ? 0 : 1;
#else
    outputAudioFile(argv[1], buf.data(), (uint32_t)n);
    if (argc > 2) {
        const std::vector<float> back = readAudioFile(argv[2]);
        if ((int)back.size() != n) return 3;
        // what AudioFile::load returns for a 24-bit sample: sign-extended integer / 8388608 (AudioFile.cpp:349-358) - i.e. the
        // truncated value the writer stored
        int bad = 0;
        for (int k = 0; k < n; ++k) {
            const float want = (float)(int32_t)(buf[k] * 8388608.0f) / 8388608.0f;
            bad += back[k] != want;
        }
        printf("read back %d samples, %d differ\n", n, bad);
    }
    return 0;
#endif
}
