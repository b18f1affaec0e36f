// Dumps what a Pocket TTS library produces for one (text, voice) pair, through the public C++ class only
// (pocket_tts/pocket_tts.hpp, the same class in the reference tree and in this one), so that the reference
// build and this build can be compared file by file with tools/parity_report.py on a machine that has both.
//
//   this tree:   make -C pocket-tts.cpp_amd ref_dump
//   reference:   g++ -std=c++17 ref_dump.cpp -I<reference>/include -L<reference>/build -lpocket_tts -o ref_dump_ort
//
//   ref_dump --models DIR --tokenizer FILE --voice WAV --text "..." --out DIR
//            [--precision fp32|int8|bf16] [--temperature 0] [--lsd-steps 10] [--max-frames 500] [--chunk 5] [--d-model 1024]
//            [--latents]
//
// Files written to --out: pcm.f32 (offline PCM, raw little-endian float32), voice.f32 (voice embedding rows),
// stream_pcm.f32 (PCM as delivered by generateStreaming), meta.json (settings, counts, chunk sizes and flags).
// --latents adds latents.f32 ([frames][latent_dim]) and eos.f32 (one logit per LM step) where the library under the
// dump offers them (pocket_tts/pocket_tts_amd.h: pocket_tts_generate_ids); against the reference tree, whose class
// keeps the per-frame values private, the flag is accepted and the dump stays PCM-only (meta.json says which).
// Temperature 0 makes the run deterministic (x0 = 0, /root/reference/src/pocket_tts.cpp:606-613).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "pocket_tts/pocket_tts.hpp"
#if __has_include("pocket_tts/pocket_tts_amd.h")
#include "pocket_tts/pocket_tts_amd.h"
#define REF_DUMP_HAS_LATENTS 1
#else
#define REF_DUMP_HAS_LATENTS 0
#endif

static void write_f32(const std::string& path, const std::vector<float>& v) {
    std::ofstream f(path, std::ios::binary);
    if (!f) { std::fprintf(stderr, "cannot write %s\n", path.c_str()); std::exit(2); }
    f.write(reinterpret_cast<const char*>(v.data()), static_cast<std::streamsize>(v.size() * sizeof(float)));
}

static std::string json_escape(const std::string& s) {
    std::string o;
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o += '\\'; o += static_cast<char>(c); }
        else if (c < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o += static_cast<char>(c);
    }
    return o;
}

int main(int argc, char** argv) {
    pocket_tts::PocketTTSConfig cfg;
    cfg.temperature = 0.0f;
    cfg.verbose = false;
    std::string voice, text = "Hello, this is a test.", out = "dump";
    int chunk = 5, d_model = 1024;   // the reference fixes 1024-wide voice rows (src/pocket_tts.cpp:560)
    bool want_latents = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); }
            return argv[++i];
        };
        if (a == "--models") cfg.modelsDir = val();
        else if (a == "--tokenizer") cfg.tokenizerPath = val();
        else if (a == "--precision") cfg.precision = val();
        else if (a == "--temperature") cfg.temperature = std::stof(val());
        else if (a == "--lsd-steps") cfg.lsdSteps = std::stoi(val());
        else if (a == "--max-frames") cfg.maxFrames = std::stoi(val());
        else if (a == "--voice") voice = val();
        else if (a == "--text") text = val();
        else if (a == "--out") out = val();
        else if (a == "--chunk") chunk = std::stoi(val());
        else if (a == "--d-model") d_model = std::stoi(val());
        else if (a == "--latents") want_latents = true;
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (voice.empty()) { std::fprintf(stderr, "--voice WAV is required\n"); return 2; }
    try {
        const auto t0 = std::chrono::steady_clock::now();
        pocket_tts::PocketTTS tts(cfg);
        const auto t1 = std::chrono::steady_clock::now();
        const std::vector<float> emb = tts.encodeVoice(voice);
        if (d_model <= 0 || emb.size() % static_cast<size_t>(d_model)) d_model = 0;   // unknown row width: skip the streaming run
        const std::vector<int64_t> shape = {1, d_model ? static_cast<int64_t>(emb.size() / d_model) : 0, d_model};
        const auto t2 = std::chrono::steady_clock::now();
        const std::vector<float> pcm = tts.generate(text, voice);
        const auto t3 = std::chrono::steady_clock::now();
        std::vector<float> stream_pcm;
        std::vector<int> chunk_n, chunk_final;
        int total = -1;
        if (d_model) {
            pocket_tts::StreamingConfig sc;
            sc.chunkSizeFrames = chunk;
            total = tts.generateStreaming(text, emb, shape,
                                          [&](const float* s, int n, bool fin) {
                                              stream_pcm.insert(stream_pcm.end(), s, s + n);
                                              chunk_n.push_back(n);
                                              chunk_final.push_back(fin ? 1 : 0);
                                          },
                                          sc);
        }
        const std::string mk = "mkdir -p '" + out + "'";
        if (std::system(mk.c_str()) != 0) { std::fprintf(stderr, "cannot create %s\n", out.c_str()); return 2; }
        write_f32(out + "/pcm.f32", pcm);
        write_f32(out + "/voice.f32", emb);
        write_f32(out + "/stream_pcm.f32", stream_pcm);
        int latent_frames = -1, latent_dim = 0, eos_count = 0;
#if REF_DUMP_HAS_LATENTS
        if (want_latents && d_model) {
            // the same settings through the C API: token ids of the text, the voice rows encoded above, start points 0
            PocketTTSConfigEx ex;
            std::memset(&ex, 0, sizeof ex);
            ex.models_dir = cfg.modelsDir.c_str(); ex.tokenizer_path = cfg.tokenizerPath.c_str(); ex.precision = cfg.precision.c_str();
            ex.temperature = cfg.temperature; ex.lsd_steps = cfg.lsdSteps; ex.max_frames = cfg.maxFrames; ex.frames_after_eos = -1;
            PocketTTSHandle h = pocket_tts_create_ex(&ex);
            if (!h) { std::fprintf(stderr, "ref_dump --latents: %s\n", pocket_tts_get_last_error()); return 1; }
            int dims[8] = {0};
            pocket_tts_dims(h, dims);
            latent_dim = dims[0];
            std::vector<long long> ids(4096);
            const int n_ids = pocket_tts_tokenize(h, text.c_str(), ids.data(), static_cast<int>(ids.size()));
            if (n_ids <= 0 || n_ids > static_cast<int>(ids.size())) { std::fprintf(stderr, "ref_dump --latents: tokenize failed\n"); return 1; }
            std::vector<float> lat(static_cast<size_t>(dims[7]) * latent_dim), eos(static_cast<size_t>(dims[7]));
            const long long* idp = ids.data();
            const float* vp = emb.data();
            const int vf = static_cast<int>(emb.size() / d_model);
            float* lp = lat.data();
            float* ep = eos.data();
            AudioResult res;
            std::memset(&res, 0, sizeof res);
            if (pocket_tts_generate_ids(h, 1, &idp, &n_ids, &vp, &vf, nullptr, &res, &lp, &ep, &latent_frames, &eos_count) != 0) {
                std::fprintf(stderr, "ref_dump --latents: %s\n", pocket_tts_get_last_error());
                return 1;
            }
            lat.resize(static_cast<size_t>(latent_frames) * latent_dim);
            eos.resize(static_cast<size_t>(eos_count));
            write_f32(out + "/latents.f32", lat);
            write_f32(out + "/eos.f32", eos);
            pocket_tts_free_audio(&res);
            pocket_tts_destroy(h);
        }
#else
        (void)want_latents;
#endif
        auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
        std::ofstream m(out + "/meta.json");
        m << "{\n  \"text\": \"" << json_escape(text) << "\",\n  \"precision\": \"" << json_escape(cfg.precision) << "\",\n"
          << "  \"temperature\": " << cfg.temperature << ",\n  \"lsd_steps\": " << cfg.lsdSteps << ",\n  \"max_frames\": " << cfg.maxFrames << ",\n"
          << "  \"samples\": " << pcm.size() << ",\n  \"frames\": " << pcm.size() / pocket_tts::PocketTTS::SAMPLES_PER_FRAME << ",\n"
          << "  \"latent_frames\": " << latent_frames << ",\n  \"latent_dim\": " << latent_dim << ",\n  \"eos_values\": " << eos_count << ",\n"
          << "  \"voice_values\": " << emb.size() << ",\n  \"stream_total\": " << total << ",\n  \"chunk_frames\": " << chunk << ",\n  \"chunks\": [";
        for (size_t i = 0; i < chunk_n.size(); ++i) m << (i ? ", " : "") << "[" << chunk_n[i] << ", " << chunk_final[i] << "]";
        m << "],\n  \"load_s\": " << secs(t0, t1) << ",\n  \"encode_s\": " << secs(t1, t2) << ",\n  \"generate_s\": " << secs(t2, t3) << "\n}\n";
        std::printf("%s: %zu samples (%.2f s of audio) in %.3f s, %zu stream chunks\n", out.c_str(), pcm.size(),
                    pcm.size() / static_cast<double>(pocket_tts::PocketTTS::SAMPLE_RATE), secs(t2, t3), chunk_n.size());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ref_dump: %s\n", e.what());
        return 1;
    }
    return 0;
}
