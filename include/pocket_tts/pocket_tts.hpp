// C++ facade of the MI355X Pocket TTS engine.
//
// Source-compatible with the class of the same name in gudrob/pocket-tts.cpp
// (include/pocket_tts/pocket_tts.hpp:30-163): same config structs, callback types, constants and
// member functions, so the reference CLI (src/main.cpp:86-96) and streaming test
// (test/test_streaming.cpp:16-48) compile against it unchanged.  Behind the facade every network runs
// as HIP kernels on one GPU; there is no ONNX Runtime and no CPU fallback.
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#if defined(_WIN32)
#  define POCKET_TTS_API
#else
#  define POCKET_TTS_API __attribute__((visibility("default")))
#endif

namespace ptts { class Engine; struct Utterance; struct UtteranceResult; }

namespace pocket_tts {

class Tokenizer;

// (samples, sampleCount, isFinal); samples are float32 24 kHz mono and valid only during the call.
using AudioChunkCallback = std::function<void(const float*, int, bool)>;
// (currentFrame, totalFrames); totalFrames is always 0 (unknown), as in the reference.
using ProgressCallback = std::function<void(int, int)>;

struct POCKET_TTS_API StreamingConfig {
    int chunkSizeFrames = 5;               // frames per delivered chunk (5 frames = 400 ms)
    ProgressCallback onProgress = nullptr; // called once per generated frame
    bool enableCancellation = false;       // poll cancelStreaming() once per frame
};

struct POCKET_TTS_API PocketTTSConfig {
    std::string modelsDir = "models/onnx";  // directory holding pocket_tts.ptts
    std::string tokenizerPath = "models/tokenizer.model";
    std::string precision = "int8";          // "int8", "fp32" or "bf16"
    float temperature = 0.7f;
    int lsdSteps = 10;
    int maxFrames = 500;
    int framesAfterEos = 3;
    bool loadVoiceEncoder = true;
    bool verbose = true;
    // ---- additions of this engine (defaults keep reference behaviour) ----
    std::string packPath;                    // explicit weight pack; empty = modelsDir + "/pocket_tts.ptts"
    int device = 0;                          // HIP device ordinal
    int maxTextTokens = 512;
    bool hasSeed = false;                    // seed the start-point noise instead of std::random_device
    uint64_t seed = 0;
    bool loadTokenizer = true;               // false: id-based entry points only
};

class POCKET_TTS_API PocketTTS {
public:
    static constexpr int SAMPLE_RATE = 24000;
    static constexpr int SAMPLES_PER_FRAME = 1920;
    static constexpr float FRAME_DURATION = static_cast<float>(SAMPLES_PER_FRAME) / SAMPLE_RATE;

    explicit PocketTTS(const PocketTTSConfig& config = PocketTTSConfig{});
    ~PocketTTS();
    PocketTTS(const PocketTTS&) = delete;
    PocketTTS& operator=(const PocketTTS&) = delete;
    PocketTTS(PocketTTS&&) noexcept;
    PocketTTS& operator=(PocketTTS&&) noexcept;

    // text + reference WAV -> PCM (encodes the voice on first use of a path, then reuses it)
    std::vector<float> generate(const std::string& text, const std::string& voicePath);
    // WAV -> conditioning rows, flattened [frames * d_model]
    std::vector<float> encodeVoice(const std::string& audioPath);
    // Extension (no counterpart in the reference): encodeVoice over many files, the files prepared on host threads while the
    // GPU encodes those already prepared; one embedding per path, in order, each also cached by path.
    std::vector<std::vector<float>> encodeVoices(const std::vector<std::string>& audioPaths);
    // text + conditioning rows (shape [1, frames, d_model]) -> PCM
    std::vector<float> generateWithEmbeddings(const std::string& text, const std::vector<float>& voiceEmbeddings,
                                              const std::vector<int64_t>& voiceEmbeddingShape);
    // chunked generation; returns the number of samples delivered through `callback`
    int generateStreaming(const std::string& text, const std::vector<float>& voiceEmbeddings,
                          const std::vector<int64_t>& voiceEmbeddingShape, AudioChunkCallback callback,
                          const StreamingConfig& streamConfig = StreamingConfig{});
    void cancelStreaming();

    // ---- additions of this engine ----
    std::vector<float> encodeVoiceFromSamples(const float* samples, int count, int sampleRate);
    std::vector<int64_t> tokenize(const std::string& text) const;
    ptts::Engine& engine();
    // Additive: with POCKET_TTS_DEVICES=0,1,... in the environment the object holds one engine per listed GPU (the
    // reference has one CPU session set per object, src/pocket_tts.cpp:80-113).  pickEngine() is the engine with the fewest
    // utterances pending; generateQueue() deals a batch out over all of them.
    ptts::Engine& pickEngine();
    int engineCount() const;
    ptts::Engine& engineAt(int i);
    void generateQueue(const std::vector<ptts::Utterance>& utts, std::vector<ptts::UtteranceResult>& out, bool wantLatents);
    const PocketTTSConfig& config() const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace pocket_tts
