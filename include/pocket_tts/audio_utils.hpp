// WAV decode / encode, Lanczos-8 resampling and peak attenuation: the host-side audio steps that sit
// in front of the voice encoder and behind the decoder.  Same static interface as AudioUtils in
// gudrob/pocket-tts.cpp (include/pocket_tts/audio_utils.hpp:12-68); results match the reference
// bit for bit on the same input (tests compare against the reference translation unit itself).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace pocket_tts {

class __attribute__((visibility("default"))) AudioUtils {
public:
    static constexpr int TARGET_SAMPLE_RATE = 24000;

    // File -> mono float at targetSampleRate, peak-attenuated.  PCM16, PCM24 and float32; 1 or 2 channels.
    static std::vector<float> loadWav(const std::string& filepath, int targetSampleRate = TARGET_SAMPLE_RATE);
    // Same pipeline on a RIFF image already in memory.
    static std::vector<float> decodeWav(const uint8_t* bytes, size_t size, int targetSampleRate, const std::string& label);
    // Mono float32 (IEEE) WAV writer.
    static void saveWav(const std::string& filepath, const std::vector<float>& audioData, int sampleRate = TARGET_SAMPLE_RATE);
    // Windowed-sinc (Lanczos, 8 lobes) rate conversion; the kernel widens when down-sampling.
    static std::vector<float> resample(const std::vector<float>& input, int inputSampleRate, int outputSampleRate);
    static std::vector<float> stereoToMono(const std::vector<float>& stereoData);
    // Scales down to peak 0.85 when the peak exceeds it; never amplifies.
    static std::vector<float> normalize(const std::vector<float>& audio);
};

}  // namespace pocket_tts
