// C entry points over the reference's own AudioUtils (compiled from /root/reference/src/audio_utils.cpp
// by oracle/Makefile into oracle/_ref/libaudio_ref.so).  TEST INFRASTRUCTURE ONLY: used to generate and
// check golden vectors for the WAV / resample / normalise stage that feeds the voice encoder
// (reference call chain: src/pocket_tts.cpp:253 -> src/audio_utils.cpp:45-159,205-303).
#include "pocket_tts/audio_utils.hpp"

#include <cstring>
#include <string>
#include <vector>

extern "C" {

// Returns the output length; writes at most cap samples.
int audio_ref_resample(const float* in, int n, int in_rate, int out_rate, float* out, int cap) {
    std::vector<float> v(in, in + n);
    auto r = pocket_tts::AudioUtils::resample(v, in_rate, out_rate);
    int m = static_cast<int>(r.size());
    std::memcpy(out, r.data(), sizeof(float) * static_cast<size_t>(m < cap ? m : cap));
    return m;
}

int audio_ref_normalize(const float* in, int n, float* out) {
    std::vector<float> v(in, in + n);
    auto r = pocket_tts::AudioUtils::normalize(v);
    std::memcpy(out, r.data(), sizeof(float) * r.size());
    return static_cast<int>(r.size());
}

// loadWav at the given target rate; returns length, or -1 on exception.
int audio_ref_load_wav(const char* path, int target_rate, float* out, int cap) {
    try {
        auto r = pocket_tts::AudioUtils::loadWav(path, target_rate);
        int m = static_cast<int>(r.size());
        std::memcpy(out, r.data(), sizeof(float) * static_cast<size_t>(m < cap ? m : cap));
        return m;
    } catch (...) {
        return -1;
    }
}

int audio_ref_save_wav(const char* path, const float* in, int n, int rate) {
    try {
        std::vector<float> v(in, in + n);
        pocket_tts::AudioUtils::saveWav(path, v, rate);
        return 0;
    } catch (...) {
        return -1;
    }
}

}  // extern "C"
