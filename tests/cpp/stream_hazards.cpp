// Hazards of the streaming entry point that only a C++ caller can provoke, run on a GPU box by
// tests/test_gpu_scheduler.py (built by oracle/Makefile into oracle/_ref/, beside the reference's own test programs):
//   1. a chunk callback that throws: the exception reaches the caller, nothing is used after the call's stack frame is gone,
//      and the handle serves the next stream;
//   2. four concurrent streams, cancelStreaming(), then a fifth stream at once: the four end early, the fifth runs to its
//      end (the reference's single flag would be cleared by the fifth stream's start, src/pocket_tts.cpp:698).
// Usage: stream_hazards <models_dir> <tokenizer.model> <voice.wav>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "pocket_tts/pocket_tts.hpp"

using namespace pocket_tts;

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: stream_hazards <models_dir> <tokenizer.model> <voice.wav>\n"); return 2; }
    PocketTTSConfig cfg;
    cfg.modelsDir = argv[1];
    cfg.tokenizerPath = argv[2];
    cfg.precision = "bf16";
    cfg.temperature = 0.0f;
    cfg.maxFrames = 120;
    PocketTTS tts(cfg);
    const std::vector<float> voice = tts.encodeVoice(argv[3]);
    const std::vector<int64_t> shape = {1, static_cast<int64_t>(voice.size() / 1024), 1024};
    StreamingConfig sc;
    sc.chunkSizeFrames = 5;
    sc.enableCancellation = true;

    // ---- 1. throwing callback
    int chunks = 0;
    bool caught = false;
    try {
        tts.generateStreaming("Hello, this is a test.", voice, shape, [&](const float*, int, bool) {
            if (++chunks == 2) throw std::runtime_error("callback failed on purpose");
        }, sc);
    } catch (const std::runtime_error& e) {
        caught = std::string(e.what()) == "callback failed on purpose";
    }
    if (!caught || chunks != 2) { std::printf("FAIL throwing callback: caught=%d chunks=%d\n", caught, chunks); return 1; }
    int after = 0;
    const int total = tts.generateStreaming("Hello again.", voice, shape, [&](const float*, int n, bool) { after += n; }, sc);
    if (total != after || total != 120 * 1920) { std::printf("FAIL stream after a throwing callback: %d %d\n", total, after); return 1; }
    std::printf("throwing callback: ok (%d samples afterwards)\n", total);

    // ---- 2. cancel four running streams, start a fifth at once
    std::atomic<int> started{0};
    std::vector<int> got(5, 0);
    std::vector<std::thread> th;
    for (int i = 0; i < 4; ++i)
        th.emplace_back([&, i] {
            bool first = true;
            got[i] = tts.generateStreaming("Concurrent stream number " + std::to_string(i) + ".", voice, shape, [&](const float*, int, bool) {
                if (first) { first = false; started.fetch_add(1); }
                std::this_thread::sleep_for(std::chrono::milliseconds(3));   // a slow consumer: the stream is still running when the cancel comes
            }, sc);
        });
    while (started.load() < 4) std::this_thread::sleep_for(std::chrono::microseconds(200));
    tts.cancelStreaming();
    got[4] = tts.generateStreaming("The fifth stream starts right after the cancel.", voice, shape, [&](const float*, int, bool) {}, sc);
    for (auto& t : th) t.join();
    const int full = 120 * 1920;
    for (int i = 0; i < 4; ++i)
        if (got[i] <= 0 || got[i] >= full) { std::printf("FAIL cancelled stream %d delivered %d of %d samples\n", i, got[i], full); return 1; }
    if (got[4] != full) { std::printf("FAIL the fifth stream delivered %d of %d samples\n", got[4], full); return 1; }
    std::printf("per-stream cancel: ok (%d %d %d %d cancelled, fifth %d)\n", got[0], got[1], got[2], got[3], got[4]);
    std::printf("OK\n");
    return 0;
}
