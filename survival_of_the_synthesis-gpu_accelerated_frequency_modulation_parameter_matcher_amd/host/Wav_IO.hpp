// Wav_IO.hpp -- the WAV reader and writer of the sots_match driver: mono float read (8/16/24/32-bit PCM or
// 32-bit float; first channel of multichannel files: the reference reads through libsndfile, main.cpp:307-335), 24-bit
// 44.1 kHz mono write as the reference's outputAudioFile does through AudioFile (main.cpp:337-366).
//
// Pinned against the reference's own AudioFile.cpp, compiled here as it stands: tests/golden/audiofile_24bit_v1.wav is the
// file it writes for tests/golden/wav_samples.inc (tests/golden/make_wav_golden.sh); tests/test_host_cpu.py requires
// outputAudioFile to write the same bytes and readAudioFile to return the values AudioFile::load returns.  What the pin
// fixed: the reference quantises by TRUNCATION, (int32_t)(sample * 8388608.0) (AudioFile.cpp:595) - this writer rounded to
// nearest on a scale of 8388607.  One deliberate difference: the reference does not clamp, so a sample of +1.0 (the 2-operator
// voice at amplitude 1 reaches it: table[8192] = 1) becomes 0x800000 = -1.0 in the file and anything beyond wraps around;
// here the integer is clamped to the 24-bit range.
#ifndef SOTS_WAV_IO_HPP
#define SOTS_WAV_IO_HPP

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iterator>
#include <stdexcept>
#include <string>
#include <vector>

static inline uint32_t rd32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

static inline std::vector<float> readAudioFile(const std::string &path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + path);
    std::vector<unsigned char> d((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    if (d.size() < 44 || memcmp(d.data(), "RIFF", 4) || memcmp(d.data() + 8, "WAVE", 4)) throw std::runtime_error(path + ": not a RIFF/WAVE file");
    uint16_t fmt = 1, channels = 1, bits = 16;
    size_t pos = 12;
    std::vector<float> out;
    while (pos + 8 <= d.size()) {
        const uint32_t len = rd32(&d[pos + 4]);
        const unsigned char *body = &d[pos + 8];
        if (!memcmp(&d[pos], "fmt ", 4) && len >= 16) {
            // the fields read below must lie inside the file (a truncated or odd 44-45 byte file ends
            // inside the chunk): 16 bytes of PCM header, 26 when the extensible sub-format is read
            if (pos + 8 + std::min<size_t>(len, 26) > d.size()) throw std::runtime_error(path + ": truncated fmt chunk");
            fmt = rd16(body);
            channels = rd16(body + 2);
            bits = rd16(body + 14);
            if (fmt == 0xFFFE && len >= 26) fmt = rd16(body + 24); // WAVE_FORMAT_EXTENSIBLE
        } else if (!memcmp(&d[pos], "data", 4)) {
            const size_t avail = std::min<size_t>(len, d.size() - pos - 8);
            const size_t bytes = bits / 8, frame = bytes * channels;
            if (frame == 0) throw std::runtime_error(path + ": bad format chunk");
            for (size_t o = 0; o + frame <= avail; o += frame) {
                const unsigned char *s = body + o;
                float v = 0.0f;
                if (fmt == 3 && bits == 32) { uint32_t u = rd32(s); memcpy(&v, &u, 4); }
                else if (bits == 8) v = ((int)s[0] - 128) / 128.0f;
                else if (bits == 16) v = (int16_t)rd16(s) / 32768.0f;
                else if (bits == 24) v = (float)((int32_t)((s[0] << 8) | (s[1] << 16) | ((uint32_t)s[2] << 24)) >> 8) / 8388608.0f;
                else if (bits == 32) v = (float)((int32_t)rd32(s) / 2147483648.0);
                else throw std::runtime_error(path + ": unsupported sample format");
                out.push_back(v);
            }
            break;
        }
        pos += 8 + len + (len & 1);
    }
    if (out.empty()) throw std::runtime_error(path + ": no audio data");
    return out;
}

static inline void outputAudioFile(const std::string &path, const float *audio, uint32_t n)
{
    std::ofstream out(path, std::ios::binary);
    if (!out) throw std::runtime_error("cannot write " + path);
    const uint32_t rate = 44100, bytes = n * 3;
    auto w32 = [&](uint32_t v) { out.put((char)v).put((char)(v >> 8)).put((char)(v >> 16)).put((char)(v >> 24)); };
    auto w16 = [&](uint16_t v) { out.put((char)v).put((char)(v >> 8)); };
    out.write("RIFF", 4); w32(36 + bytes); out.write("WAVEfmt ", 8); w32(16); w16(1); w16(1); w32(rate); w32(rate * 3); w16(3); w16(24);
    out.write("data", 4); w32(bytes);
    for (uint32_t i = 0; i < n; ++i) {
        // AudioFile.cpp:595: (int32_t)(sample * 8388608.) - truncation towards zero; clamped here where the reference wraps
        const float scaled = audio[i] * 8388608.0f;
        const int32_t q = !(scaled > -8388608.0f) ? -8388608 : scaled >= 8388607.0f ? 8388607 : (int32_t)scaled; // (NaN: the low end)
        out.put((char)q).put((char)(q >> 8)).put((char)(q >> 16));
    }
}

#endif
