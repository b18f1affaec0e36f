// Command-line front end with the same arguments as the reference CLI
// (/root/reference/src/main.cpp:9-26, 37-67): text, voice WAV, output WAV and the six options.
// It is the only stock entry point that can request temperature 0 (deterministic start point).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>

#include "pocket_tts/audio_utils.hpp"
#include "pocket_tts/pocket_tts.hpp"

static void usage(const char* prog) {
    std::printf("Pocket TTS (MI355X engine) - text-to-speech with voice cloning\n\n"
                "Usage: %s [options] <text> <voice_file> <output_file>\n\n"
                "Options:\n"
                "  --models-dir <path>   directory holding pocket_tts.ptts (default: models/onnx)\n"
                "  --tokenizer <path>    SentencePiece model (default: models/tokenizer.model)\n"
                "  --precision <p>       int8, bf16 or fp32 (default: int8)\n"
                "  --temperature <t>     sampling temperature (default: 0.7)\n"
                "  --lsd-steps <n>       flow matching steps (default: 10)\n"
                "  --max-frames <n>      maximum frames to generate (default: 500)\n"
                "  --device <n>          HIP device ordinal (default: 0)\n"
                "  -h, --help            this message\n",
                prog);
}

int main(int argc, char** argv) {
    pocket_tts::PocketTTSConfig cfg;
    std::string pos[3];
    int npos = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
        if (a == "-h" || a == "--help") { usage(argv[0]); return 0; }
        else if (a == "--models-dir") { if (auto v = next()) cfg.modelsDir = v; }
        else if (a == "--tokenizer") { if (auto v = next()) cfg.tokenizerPath = v; }
        else if (a == "--precision") { if (auto v = next()) cfg.precision = v; }
        else if (a == "--temperature") { if (auto v = next()) cfg.temperature = std::stof(v); }
        else if (a == "--lsd-steps") { if (auto v = next()) cfg.lsdSteps = std::stoi(v); }
        else if (a == "--max-frames") { if (auto v = next()) cfg.maxFrames = std::stoi(v); }
        else if (a == "--device") { if (auto v = next()) cfg.device = std::stoi(v); }
        else if (!a.empty() && a[0] != '-') { if (npos < 3) pos[npos] = a; ++npos; }
        else { std::fprintf(stderr, "Unknown option: %s\n", a.c_str()); return 1; }
    }
    if (npos < 3) {
        std::fprintf(stderr, "Error: Missing required arguments\n\n");
        usage(argv[0]);
        return 1;
    }
    try {
        auto t0 = std::chrono::steady_clock::now();
        pocket_tts::PocketTTS tts(cfg);
        std::printf("Loaded in %.3fs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        auto audio = tts.generate(pos[0], pos[1]);
        pocket_tts::AudioUtils::saveWav(pos[2], audio);
        std::printf("Saved to: %s\n", pos[2].c_str());
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
}
