// Host-side audio utilities.  Behaviour follows /root/reference/src/audio_utils.cpp (loadWav :45-159,
// saveWav :163-201, resample :205-259, stereoToMono :263-276, normalize :280-303) so that a voice clip
// reaches the encoder with exactly the samples the reference would feed it.
#include "pocket_tts/audio_utils.hpp"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace pocket_tts {

namespace {

constexpr double kPi = 3.14159265358979323846;

struct Cursor {
    const uint8_t* p;
    size_t size, pos = 0;
    bool read(void* dst, size_t n) {
        if (pos + n > size) {
            // short file: deliver what exists, zero the rest, and report failure like a stream would
            size_t have = pos < size ? size - pos : 0;
            if (have) std::memcpy(dst, p + pos, have);
            std::memset(static_cast<uint8_t*>(dst) + have, 0, n - have);
            pos = size;
            return false;
        }
        std::memcpy(dst, p + pos, n);
        pos += n;
        return true;
    }
};

uint16_t le16(const uint8_t* b) { return static_cast<uint16_t>(b[0] | (b[1] << 8)); }
uint32_t le32(const uint8_t* b) { return static_cast<uint32_t>(b[0]) | (static_cast<uint32_t>(b[1]) << 8) | (static_cast<uint32_t>(b[2]) << 16) | (static_cast<uint32_t>(b[3]) << 24); }

// Lanczos window of half-width a evaluated at x.
double lanczos(double x, int a) {
    if (x == 0.0) return 1.0;
    if (x < -a || x > a) return 0.0;
    const double px = kPi * x;
    const double pxa = kPi * x / a;
    return (std::sin(px) / px) * (std::sin(pxa) / pxa);
}

}  // namespace

std::vector<float> AudioUtils::decodeWav(const uint8_t* bytes, size_t size, int targetSampleRate, const std::string& label) {
    Cursor c{bytes, size};
    uint8_t riff[12];
    if (!c.read(riff, 12) || std::memcmp(riff, "RIFF", 4) != 0 || std::memcmp(riff + 8, "WAVE", 4) != 0)
        throw std::runtime_error("Not a valid WAV file: " + label);

    bool have_fmt = false;
    uint16_t format = 0, channels = 0, bits = 0;
    uint32_t rate = 0;
    std::vector<float> mono;

    for (;;) {
        uint8_t hdr[8];
        if (!c.read(hdr, 8)) break;
        const uint32_t csize = le32(hdr + 4);
        const size_t body = c.pos;
        if (std::memcmp(hdr, "fmt ", 4) == 0) {
            if (csize < 16) throw std::runtime_error("Invalid fmt chunk in: " + label);
            uint8_t f[16];
            if (!c.read(f, 16)) throw std::runtime_error("Truncated fmt chunk in: " + label);
            format = le16(f); channels = le16(f + 2); rate = le32(f + 4); bits = le16(f + 14);
            if (format != 1 && format != 3) throw std::runtime_error("Unsupported WAV format (only PCM/float supported): " + label);
            if (channels < 1 || channels > 2) throw std::runtime_error("Unsupported channel count (only mono/stereo): " + label);
            have_fmt = true;
        } else if (std::memcmp(hdr, "data", 4) == 0) {
            if (!have_fmt) throw std::runtime_error("data chunk before fmt chunk in: " + label);
            const size_t bps = bits / 8;
            const bool f32 = format == 3 && bits == 32, i16 = format == 1 && bits == 16, i24 = format == 1 && bits == 24;
            if (!f32 && !i16 && !i24)
                throw std::runtime_error("Unsupported bit depth " + std::to_string(bits) + " for format " + std::to_string(format) + " in: " + label);
            const size_t n = csize / bps;
            std::vector<uint8_t> raw(csize);
            c.read(raw.data(), csize);
            std::vector<float> samples(n);
            if (f32) {
                std::memcpy(samples.data(), raw.data(), n * 4);
            } else if (i16) {
                constexpr float k = 1.0f / 32768.0f;
                for (size_t i = 0; i < n; ++i) samples[i] = static_cast<float>(static_cast<int16_t>(le16(&raw[2 * i]))) * k;
            } else {
                constexpr float k = 1.0f / 8388608.0f;
                for (size_t i = 0; i < n; ++i) {
                    int32_t v = raw[3 * i] | (raw[3 * i + 1] << 8) | (raw[3 * i + 2] << 16);
                    if (v & 0x800000) v |= static_cast<int32_t>(0xFF000000u);
                    samples[i] = static_cast<float>(v) * k;
                }
            }
            mono = channels == 1 ? std::move(samples) : stereoToMono(samples);
            break;
        }
        c.pos = body + csize + (csize & 1u);  // chunks are word aligned
        if (c.pos > c.size) break;
    }
    if (mono.empty()) throw std::runtime_error("No audio data found in: " + label);
    if (have_fmt && static_cast<int>(rate) != targetSampleRate) mono = resample(mono, static_cast<int>(rate), targetSampleRate);
    return normalize(mono);
}

std::vector<float> AudioUtils::loadWav(const std::string& filepath, int targetSampleRate) {
    FILE* f = std::fopen(filepath.c_str(), "rb");
    if (!f) throw std::runtime_error("Failed to open audio file: " + filepath);
    std::vector<uint8_t> bytes;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + got);
    std::fclose(f);
    return decodeWav(bytes.data(), bytes.size(), targetSampleRate, filepath);
}

void AudioUtils::saveWav(const std::string& filepath, const std::vector<float>& audioData, int sampleRate) {
    FILE* f = std::fopen(filepath.c_str(), "wb");
    if (!f) throw std::runtime_error("Failed to create audio file: " + filepath);
    const uint32_t n = static_cast<uint32_t>(audioData.size());
    const uint32_t data_bytes = n * 4u;
    uint8_t h[44];
    auto put32 = [&](int o, uint32_t v) { h[o] = v & 255; h[o + 1] = (v >> 8) & 255; h[o + 2] = (v >> 16) & 255; h[o + 3] = (v >> 24) & 255; };
    auto put16 = [&](int o, uint16_t v) { h[o] = v & 255; h[o + 1] = (v >> 8) & 255; };
    std::memcpy(h, "RIFF", 4); put32(4, 4 + (8 + 16) + (8 + data_bytes)); std::memcpy(h + 8, "WAVE", 4);
    std::memcpy(h + 12, "fmt ", 4); put32(16, 16);
    put16(20, 3); put16(22, 1); put32(24, static_cast<uint32_t>(sampleRate)); put32(28, static_cast<uint32_t>(sampleRate) * 4u);
    put16(32, 4); put16(34, 32);
    std::memcpy(h + 36, "data", 4); put32(40, data_bytes);
    std::fwrite(h, 1, 44, f);
    if (n) std::fwrite(audioData.data(), 4, n, f);
    std::fclose(f);
}

std::vector<float> AudioUtils::resample(const std::vector<float>& input, int inputSampleRate, int outputSampleRate) {
    if (inputSampleRate == outputSampleRate || input.empty()) return input;
    constexpr int A = 8;
    const double ratio = static_cast<double>(outputSampleRate) / inputSampleRate;
    const size_t out_n = static_cast<size_t>(std::ceil(input.size() * ratio));
    const double step = static_cast<double>(inputSampleRate) / outputSampleRate;
    const double fscale = ratio < 1.0 ? ratio : 1.0;  // low-pass widening when down-sampling
    const double radius = A / fscale;
    const int in_n = static_cast<int>(input.size());
    std::vector<float> out(out_n);
    // The tap weights of an output sample depend on its fractional position alone.  They are kept per distinct value of that
    // fraction (compared bit for bit, so a reused weight is the very double the expression yields): when the rate ratio is a
    // dyadic number (48 kHz or 96 kHz to 24 kHz: every fraction is 0) two sines per tap become two per file; for other ratios
    // (44.1 kHz) the products i * step round differently for each i and nearly every sample computes its own.
    struct Phase { uint64_t key; int lo, hi; bool used; std::vector<double> w; };
    constexpr size_t kPhases = 256;
    std::vector<Phase> phases(kPhases);
    for (auto& p : phases) p.used = false;
    for (size_t i = 0; i < out_n; ++i) {
        const double src = i * step;
        const int center = static_cast<int>(std::floor(src));
        const double frac = src - center;
        uint64_t key;
        std::memcpy(&key, &frac, 8);
        Phase& ph = phases[(key ^ (key >> 17) ^ (key >> 43)) % kPhases];
        if (!ph.used || ph.key != key) {
            ph.key = key; ph.used = true;
            ph.lo = static_cast<int>(std::ceil(-radius + frac)); ph.hi = static_cast<int>(std::floor(radius + frac));
            ph.w.resize(static_cast<size_t>(ph.hi - ph.lo + 1));
            for (int j = ph.lo; j <= ph.hi; ++j) ph.w[static_cast<size_t>(j - ph.lo)] = lanczos((j - frac) * fscale, A);
        }
        double acc = 0.0, wsum = 0.0;
        const double* wp = ph.w.data() - ph.lo;
        for (int j = ph.lo; j <= ph.hi; ++j) {
            const int idx = center + j;
            if (idx < 0 || idx >= in_n) continue;
            const double w = wp[j];
            acc += input[idx] * w;
            wsum += w;
        }
        out[i] = wsum > 0.0 ? static_cast<float>(acc / wsum) : 0.0f;
    }
    return out;
}

std::vector<float> AudioUtils::stereoToMono(const std::vector<float>& stereoData) {
    if (stereoData.size() % 2 != 0) throw std::runtime_error("Stereo data must have even number of samples");
    std::vector<float> mono(stereoData.size() / 2);
    for (size_t i = 0; i < mono.size(); ++i) mono[i] = (stereoData[2 * i] + stereoData[2 * i + 1]) * 0.5f;
    return mono;
}

std::vector<float> AudioUtils::normalize(const std::vector<float>& audio) {
    if (audio.empty()) return audio;
    constexpr float kPeak = 0.85f;
    float peak = 0.0f;
    for (float v : audio) peak = std::max(peak, std::abs(v));
    if (peak <= kPeak) return audio;
    const float gain = kPeak / peak;
    std::vector<float> out(audio.size());
    for (size_t i = 0; i < audio.size(); ++i) out[i] = audio[i] * gain;
    return out;
}

}  // namespace pocket_tts
