// pocket_tts::PocketTTS on top of the HIP engine: text pre-step, voice clip preparation, path-keyed
// voice cache and the mapping of the reference's option structs onto engine options.
// Interface contract: /root/reference/include/pocket_tts/pocket_tts.hpp:78-163; the behaviour of each
// method follows /root/reference/src/pocket_tts.cpp (encodeVoice :241-297, generate :548-657,
// generateWithEmbeddings :676-684, generateStreaming :686-979, cancelStreaming :981-983).
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <iostream>
#include <map>
#include <mutex>
#include <set>
#include <sstream>
#include <thread>

#include "engine.h"
#include "pocket_tts/audio_utils.hpp"
#include "pocket_tts/pocket_tts.hpp"
#include "pocket_tts/tokenizer.hpp"

namespace pocket_tts {

struct PocketTTS::Impl {
    PocketTTSConfig config;
    std::unique_ptr<ptts::Engine> engine;
    // POCKET_TTS_DEVICES=0,1,...: one more engine (weights, slot table, worker thread) per further entry behind this one
    // handle (an ordinal may repeat: several engines on one GPU); single utterances and streams go to the engine with the fewest utterances pending, queued batches are dealt
    // out over all of them.  `engine` is the first listed device and serves everything else (operators, voice encoding).
    std::vector<std::unique_ptr<ptts::Engine>> more;
    std::unique_ptr<Tokenizer> tokenizer;
    std::mutex cache_mu;
    std::map<std::string, std::vector<float>> voice_cache;  // keyed by path, like the reference (:75, :243-246)
    // One flag per running stream.  cancelStreaming() raises the flags of the streams that are running when it is called: a
    // stream that starts afterwards is not affected and cannot un-cancel the others (the reference has one stream per handle
    // and one flag that every stream start clears, src/pocket_tts.cpp:698, :981-983).
    std::mutex cancel_mu;
    std::set<std::atomic<bool>*> live_streams;

    explicit Impl(const PocketTTSConfig& cfg) : config(cfg) {
        ptts::EngineOptions o;
        o.pack_path = cfg.packPath.empty() ? cfg.modelsDir + "/pocket_tts.ptts" : cfg.packPath;
        if (cfg.precision == "fp32") o.precision = ptts::Precision::FP32;
        else if (cfg.precision == "bf16") o.precision = ptts::Precision::BF16;
        else if (cfg.precision == "int8") o.precision = ptts::Precision::INT8;
        else throw std::runtime_error("unknown precision '" + cfg.precision + "' (expected int8, bf16 or fp32)");
        o.device = cfg.device;
        o.lsd_steps = cfg.lsdSteps;
        o.max_frames = cfg.maxFrames;
        o.frames_after_eos = cfg.framesAfterEos;
        o.temperature = cfg.temperature;
        // POCKET_TTS_TEMPERATURE overrides the configured value: the reference's C API applies the field only when > 0
        // (src/pocket_tts_c.cpp:33-40), so this is how a caller of the unchanged bindings gets temperature 0 (deterministic runs)
        if (const char* e = std::getenv("POCKET_TTS_TEMPERATURE")) {
            char* end = nullptr;
            const float t = std::strtof(e, &end);
            if (end == e || !(t >= 0.0f)) throw std::runtime_error(std::string("POCKET_TTS_TEMPERATURE is not a non-negative number: ") + e);
            o.temperature = t;
        }
        o.max_text_tokens = cfg.maxTextTokens;
        o.seed = cfg.seed;
        o.has_seed = cfg.hasSeed;
        if (cfg.verbose) std::cout << "Loading weight pack " << o.pack_path << " (precision: " << cfg.precision << ")..." << std::endl;
        std::vector<int> devices;
        if (const char* e = std::getenv("POCKET_TTS_DEVICES")) {
            std::stringstream ss(e);
            for (std::string tok; std::getline(ss, tok, ',');)
                if (!tok.empty()) devices.push_back(std::stoi(tok));
        }
        // POCKET_TTS_ENGINES_PER_GPU=n (without a device list): n engines on the configured device, each with its own slot
        // table and streams.  A frame step is bound by the latency of its dependent launches, not by the chip, so two
        // tables side by side serve about a quarter more audio per second than one (the first holds the persistent flow
        // head, the others run the head as a launch chain).  At most 3: measured on MI355X, 512 utterances of 125 frames,
        // int8 5060 / 6550 / 7520 / 7080 RTFx with 1 / 2 / 3 / 4 engines, bf16 5620 / 7080 / 7240; beyond three the streams of
        // the engines no longer get a hardware queue each (GPU_MAX_HW_QUEUES is 4 by default).
        if (devices.empty())
            if (const char* e = std::getenv("POCKET_TTS_ENGINES_PER_GPU"))
                devices.assign(static_cast<size_t>(std::max(1, std::min(std::atoi(e), 3))), cfg.device);
        if (!devices.empty()) o.device = devices[0];
        {
            // Every engine's frame loop needs a hardware queue of its own (two streams on one queue run their work one after
            // the other: measured 7550 -> 4180 RTFx with two engines), and the device's side stream takes one more.  A process
            // gets GPU_MAX_HW_QUEUES of them per GPU (4 unless the variable says otherwise); CU-masked decoder streams do not count.
            int hwq = 4, lanes = 1;
            if (const char* e = std::getenv("GPU_MAX_HW_QUEUES")) hwq = std::max(1, std::atoi(e));
            if (const char* e = std::getenv("POCKET_TTS_LANES")) lanes = std::max(1, std::min(std::atoi(e), 8));
            std::map<int, int> per_dev;
            for (int d : devices) per_dev[d] += lanes;
            for (auto& kv : per_dev)
                if (kv.second > 1 && kv.second + 1 > hwq)
                    throw std::runtime_error("GPU " + std::to_string(kv.first) + ": " + std::to_string(kv.second) + " frame-loop streams and the side stream need " +
                                             std::to_string(kv.second + 1) + " hardware queues, the process has " + std::to_string(hwq) +
                                             " (GPU_MAX_HW_QUEUES); list fewer engines for that GPU");
        }
        engine = std::make_unique<ptts::Engine>(o);
        for (size_t i = 1; i < devices.size(); ++i) {
            ptts::EngineOptions oi = o;
            oi.device = devices[i];
            if (oi.has_seed) oi.seed += i;
            more.push_back(std::make_unique<ptts::Engine>(oi));
        }
        if (cfg.verbose && !more.empty()) std::cout << devices.size() << " engines behind this handle." << std::endl;
        if (cfg.loadTokenizer) {
            tokenizer = std::make_unique<Tokenizer>(cfg.tokenizerPath);
            if (cfg.verbose) std::cout << "Tokenizer loaded (vocab size: " << tokenizer->vocabSize() << ")." << std::endl;
        }
    }

    ptts::Engine& least_loaded() {
        ptts::Engine* best = engine.get();
        for (auto& e : more)
            if (e->pending() < best->pending()) best = e.get();
        return *best;
    }

    std::vector<int64_t> tokenize(const std::string& text) const {
        if (!tokenizer) throw std::runtime_error("tokenizer not loaded");
        return tokenizer->encode(text);
    }

    // 24 kHz mono clip -> conditioning rows.  Only the first 5 s are used (:254-259).
    std::vector<float> encode_clip(std::vector<float> audio) {
        if (!config.loadVoiceEncoder) throw std::runtime_error("Voice encoder is disabled (loadVoiceEncoder=false).");
        const size_t max_samples = static_cast<size_t>(AudioUtils::TARGET_SAMPLE_RATE) * 5;
        if (audio.size() > max_samples) audio.resize(max_samples);
        return engine->encode_voice(audio.data(), static_cast<int>(audio.size()));
    }

    std::vector<float> encode_voice_path(const std::string& path) {
        {
            std::lock_guard<std::mutex> lk(cache_mu);
            auto it = voice_cache.find(path);
            if (it != voice_cache.end()) return it->second;
        }
        auto audio = AudioUtils::loadWav(path, AudioUtils::TARGET_SAMPLE_RATE);  // decode, resample, attenuate (:253)
        auto emb = encode_clip(std::move(audio));
        std::lock_guard<std::mutex> lk(cache_mu);
        voice_cache[path] = emb;
        return emb;
    }

    // Many voice files at once (the voice-clone workload: a thousand reference clips): host threads decode, resample,
    // attenuate and cut the files (src/audio_utils.cpp:45-159, 205-259; src/pocket_tts.cpp:253-259) while the GPU encodes the
    // window of clips that is already prepared; results land in the path-keyed cache like single calls (:243-246, :294).
    std::vector<std::vector<float>> encode_voice_paths(const std::vector<std::string>& paths) {
        if (!config.loadVoiceEncoder) throw std::runtime_error("Voice encoder is disabled (loadVoiceEncoder=false).");
        const size_t n = paths.size(), max_samples = static_cast<size_t>(AudioUtils::TARGET_SAMPLE_RATE) * 5;
        std::vector<std::vector<float>> out(n), clips(n);
        std::vector<char> cached(n, 0), ready(n, 0);
        {
            std::lock_guard<std::mutex> lk(cache_mu);
            for (size_t i = 0; i < n; ++i) {
                auto it = voice_cache.find(paths[i]);
                if (it != voice_cache.end()) { out[i] = it->second; cached[i] = 1; }
            }
        }
        std::mutex mu;
        std::condition_variable cv;
        std::atomic<size_t> next{0};
        std::exception_ptr failed;
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const size_t workers = std::min<size_t>({n, 16, hw});
        std::vector<std::thread> pool;
        for (size_t w = 0; w < workers; ++w)
            pool.emplace_back([&] {
                for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
                    std::exception_ptr err;
                    if (!cached[i]) {
                        try {
                            clips[i] = AudioUtils::loadWav(paths[i], AudioUtils::TARGET_SAMPLE_RATE);
                            if (clips[i].size() > max_samples) clips[i].resize(max_samples);
                            if (clips[i].empty()) throw std::runtime_error("voice clip is empty: " + paths[i]);
                        } catch (...) { err = std::current_exception(); }
                    }
                    std::lock_guard<std::mutex> lk(mu);
                    if (err && !failed) failed = err;
                    ready[i] = 1;
                    cv.notify_all();
                }
            });
        const size_t window = 32;   // clips per encoder call; the engine stacks them 8 at a time over two buffer sets
        std::exception_ptr enc_failed;
        for (size_t first = 0; first < n && !enc_failed; first += window) {
            const size_t last = std::min(n, first + window);
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { for (size_t i = first; i < last; ++i) if (!ready[i]) return false; return true; });
                if (failed) break;
            }
            std::vector<const float*> ptrs;
            std::vector<int> lens;
            std::vector<size_t> idx;
            for (size_t i = first; i < last; ++i)
                if (!cached[i]) { ptrs.push_back(clips[i].data()); lens.push_back(static_cast<int>(clips[i].size())); idx.push_back(i); }
            if (ptrs.empty()) continue;
            try {
                std::vector<std::vector<float>> emb;
                engine->encode_voice_batch(ptrs, lens, emb);
                for (size_t k = 0; k < idx.size(); ++k) { out[idx[k]] = std::move(emb[k]); std::vector<float>().swap(clips[idx[k]]); }
            } catch (...) { enc_failed = std::current_exception(); }
        }
        for (auto& t : pool) t.join();
        if (failed) std::rethrow_exception(failed);
        if (enc_failed) std::rethrow_exception(enc_failed);
        std::lock_guard<std::mutex> lk(cache_mu);
        for (size_t i = 0; i < n; ++i)
            if (!cached[i]) voice_cache[paths[i]] = out[i];
        return out;
    }

    int frames_of(const std::vector<float>& emb, const std::vector<int64_t>& shape) const {
        const int D = engine->spec().d_model;
        if (shape.size() >= 2) {
            int64_t rows = shape[shape.size() - 2];
            if (rows >= 0 && static_cast<size_t>(rows) * D <= emb.size()) return static_cast<int>(rows);
        }
        return static_cast<int>(emb.size() / D);
    }
};

PocketTTS::PocketTTS(const PocketTTSConfig& config) : impl_(std::make_unique<Impl>(config)) {}
PocketTTS::~PocketTTS() = default;
PocketTTS::PocketTTS(PocketTTS&&) noexcept = default;
PocketTTS& PocketTTS::operator=(PocketTTS&&) noexcept = default;

ptts::Engine& PocketTTS::engine() { return *impl_->engine; }
ptts::Engine& PocketTTS::pickEngine() { return impl_->least_loaded(); }
int PocketTTS::engineCount() const { return 1 + static_cast<int>(impl_->more.size()); }
ptts::Engine& PocketTTS::engineAt(int i) { return i <= 0 ? *impl_->engine : *impl_->more.at(static_cast<size_t>(i - 1)); }

// Queued utterances over every engine of the handle: dealt out in turn starting with the least loaded engine, each engine's
// share runs through its own slot table, all of them at once.
void PocketTTS::generateQueue(const std::vector<ptts::Utterance>& utts, std::vector<ptts::UtteranceResult>& out, bool wantLatents) {
    if (impl_->more.empty()) { impl_->engine->generate_queue(utts, out, wantLatents); return; }
    std::vector<ptts::Engine*> engines{impl_->engine.get()};
    for (auto& e : impl_->more) engines.push_back(e.get());
    std::stable_sort(engines.begin(), engines.end(), [](ptts::Engine* a, ptts::Engine* b) { return a->pending() < b->pending(); });
    const size_t E = engines.size();
    std::vector<std::vector<ptts::Utterance>> part(E);
    std::vector<std::vector<size_t>> index(E);
    for (size_t i = 0; i < utts.size(); ++i) { part[i % E].push_back(utts[i]); index[i % E].push_back(i); }
    std::vector<std::vector<ptts::UtteranceResult>> res(E);
    std::vector<std::exception_ptr> err(E);
    std::vector<std::thread> th;
    for (size_t e = 0; e < E; ++e)
        th.emplace_back([&, e] {
            try { if (!part[e].empty()) engines[e]->generate_queue(part[e], res[e], wantLatents); }
            catch (...) { err[e] = std::current_exception(); }
        });
    for (auto& t : th) t.join();
    for (auto& x : err) if (x) std::rethrow_exception(x);
    out.assign(utts.size(), ptts::UtteranceResult());
    for (size_t e = 0; e < E; ++e)
        for (size_t k = 0; k < index[e].size(); ++k) out[index[e][k]] = std::move(res[e][k]);
}
const PocketTTSConfig& PocketTTS::config() const { return impl_->config; }
std::vector<int64_t> PocketTTS::tokenize(const std::string& text) const { return impl_->tokenize(text); }

std::vector<float> PocketTTS::encodeVoice(const std::string& audioPath) { return impl_->encode_voice_path(audioPath); }
std::vector<std::vector<float>> PocketTTS::encodeVoices(const std::vector<std::string>& audioPaths) { return impl_->encode_voice_paths(audioPaths); }

std::vector<float> PocketTTS::encodeVoiceFromSamples(const float* samples, int count, int sampleRate) {
    std::vector<float> audio(samples, samples + count);
    if (sampleRate != AudioUtils::TARGET_SAMPLE_RATE) audio = AudioUtils::resample(audio, sampleRate, AudioUtils::TARGET_SAMPLE_RATE);
    // the reference round-trips through a float32 WAV and loadWav, whose only further effect is the
    // peak attenuation (src/pocket_tts_c.cpp:96-107 -> src/audio_utils.cpp:156)
    audio = AudioUtils::normalize(audio);
    return impl_->encode_clip(std::move(audio));
}

std::vector<float> PocketTTS::generate(const std::string& text, const std::string& voicePath) {
    auto t0 = std::chrono::high_resolution_clock::now();
    auto emb = impl_->encode_voice_path(voicePath);
    auto audio = generateWithEmbeddings(text, emb, {1, static_cast<int64_t>(emb.size() / impl_->engine->spec().d_model), impl_->engine->spec().d_model});
    if (impl_->config.verbose) {
        auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count();
        float dur = static_cast<float>(audio.size()) / impl_->engine->spec().sample_rate;
        std::cout << "Generated " << dur << "s audio in " << (ms / 1000.0f) << "s (RTFx: " << dur / (ms / 1000.0f) << "x)" << std::endl;
    }
    return audio;
}

std::vector<float> PocketTTS::generateWithEmbeddings(const std::string& text, const std::vector<float>& voiceEmbeddings,
                                                     const std::vector<int64_t>& voiceEmbeddingShape) {
    ptts::Utterance u;
    u.token_ids = impl_->tokenize(text);
    u.voice = voiceEmbeddings.data();
    u.voice_frames = impl_->frames_of(voiceEmbeddings, voiceEmbeddingShape);
    ptts::UtteranceResult res;
    impl_->least_loaded().generate_one(u, res, false);   // a slot of that engine's table, beside concurrent callers' utterances
    return std::move(res.audio);
}

int PocketTTS::generateStreaming(const std::string& text, const std::vector<float>& voiceEmbeddings,
                                 const std::vector<int64_t>& voiceEmbeddingShape, AudioChunkCallback callback,
                                 const StreamingConfig& streamConfig) {
    if (!callback) throw std::invalid_argument("Callback must be provided");
    std::atomic<bool> my_cancel{false};   // starts clear, as the reference's flag does at every stream start (:698)
    struct Registered {
        Impl& im; std::atomic<bool>* f;
        Registered(Impl& i, std::atomic<bool>* flag) : im(i), f(flag) { std::lock_guard<std::mutex> lk(im.cancel_mu); im.live_streams.insert(f); }
        ~Registered() { std::lock_guard<std::mutex> lk(im.cancel_mu); im.live_streams.erase(f); }
    } registered(*impl_, &my_cancel);
    ptts::Utterance u;
    u.token_ids = impl_->tokenize(text);
    u.voice = voiceEmbeddings.data();
    u.voice_frames = impl_->frames_of(voiceEmbeddings, voiceEmbeddingShape);
    // the tail flush is suppressed by a pending cancel request even when per-frame polling is off (:897)
    std::function<void(int)> progress;
    if (streamConfig.onProgress) progress = [&](int frame) { streamConfig.onProgress(frame, 0); };
    auto t0 = std::chrono::high_resolution_clock::now();
    int total = impl_->least_loaded().generate_streaming(u, streamConfig.chunkSizeFrames, callback, &my_cancel,
                                                  streamConfig.enableCancellation, progress, nullptr);
    if (impl_->config.verbose) {
        auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count();
        float dur = static_cast<float>(total) / impl_->engine->spec().sample_rate;
        std::cout << "Streamed " << dur << "s audio in " << (ms / 1000.0f) << "s (RTFx: " << dur / (ms / 1000.0f) << "x)" << std::endl;
    }
    return total;
}

void PocketTTS::cancelStreaming() {
    std::lock_guard<std::mutex> lk(impl_->cancel_mu);
    for (std::atomic<bool>* f : impl_->live_streams) f->store(true);
}

}  // namespace pocket_tts
