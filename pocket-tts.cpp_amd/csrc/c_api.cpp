// extern "C" surface of libpocket_tts.so: the eleven drop-in entry points of pocket_tts_c.h and the
// additive ones of pocket_tts_amd.h.  Conventions follow /root/reference/src/pocket_tts_c.cpp:
// exceptions never cross the boundary, failures set a thread-local message (:11, :23-25, :159-161).
#include <algorithm>
#include <cstring>
#include <exception>
#include <string>
#include <thread>
#include <vector>

#include "onnx_reader.h"
#include "engine.h"
#include "pocket_tts/audio_utils.hpp"
#include "pocket_tts/pocket_tts.hpp"
#include "pocket_tts/pocket_tts_amd.h"
#include "pocket_tts/tokenizer.hpp"

namespace {

thread_local std::string g_last_error;

struct VoiceData {
    std::vector<float> embeddings;  // [frames][d_model]
    std::vector<int64_t> shape;     // {1, frames, d_model}
};

void set_error(const std::string& m) { g_last_error = m; }

pocket_tts::PocketTTS* tts_of(PocketTTSHandle h) { return static_cast<pocket_tts::PocketTTS*>(h); }

VoiceHandle make_voice(std::vector<float>&& emb, int d_model) {
    auto* v = new VoiceData();
    v->embeddings = std::move(emb);
    v->shape = {1, static_cast<int64_t>(v->embeddings.size()) / d_model, d_model};
    return v;
}

void fill_result(AudioResult* r, const std::vector<float>& audio, int sample_rate) {
    r->data = new float[audio.size() ? audio.size() : 1];
    if (!audio.empty()) std::memcpy(r->data, audio.data(), audio.size() * sizeof(float));
    r->sample_count = static_cast<int>(audio.size());
    r->sample_rate = sample_rate;
}

template <typename F>
int guarded(const char* what, F&& fn) {
    try {
        return fn();
    } catch (const std::exception& e) {
        set_error(std::string(what) + e.what());
        return -1;
    }
}

}  // namespace

extern "C" {

POCKET_TTS_API PocketTTSHandle pocket_tts_create(const PocketTTSConfig* config) {
    try {
        pocket_tts::PocketTTSConfig cfg;
        if (config) {
            if (config->models_dir) cfg.modelsDir = config->models_dir;
            if (config->tokenizer_path) cfg.tokenizerPath = config->tokenizer_path;
            if (config->precision) cfg.precision = config->precision;
            if (config->temperature > 0) cfg.temperature = config->temperature;
            if (config->lsd_steps > 0) cfg.lsdSteps = config->lsd_steps;
            if (config->max_frames > 0) cfg.maxFrames = config->max_frames;
        }
        cfg.verbose = false;
        return new pocket_tts::PocketTTS(cfg);
    } catch (const std::exception& e) {
        set_error(std::string("Failed to create PocketTTS: ") + e.what());
        return nullptr;
    }
}

POCKET_TTS_API PocketTTSHandle pocket_tts_create_ex(const PocketTTSConfigEx* c) {
    try {
        pocket_tts::PocketTTSConfig cfg;
        cfg.verbose = false;
        if (c) {
            if (c->models_dir) cfg.modelsDir = c->models_dir;
            if (c->tokenizer_path) {
                std::string t = c->tokenizer_path;
                if (t.empty() || t == "none") cfg.loadTokenizer = false;
                else cfg.tokenizerPath = t;
            }
            if (c->precision) cfg.precision = c->precision;
            if (c->pack_path) cfg.packPath = c->pack_path;
            cfg.temperature = c->temperature;
            if (c->lsd_steps > 0) cfg.lsdSteps = c->lsd_steps;
            if (c->max_frames > 0) cfg.maxFrames = c->max_frames;
            if (c->frames_after_eos >= 0) cfg.framesAfterEos = c->frames_after_eos;
            cfg.device = c->device;
            if (c->max_text_tokens > 0) cfg.maxTextTokens = c->max_text_tokens;
            cfg.hasSeed = c->has_seed != 0;
            cfg.seed = c->seed;
        }
        return new pocket_tts::PocketTTS(cfg);
    } catch (const std::exception& e) {
        set_error(std::string("Failed to create PocketTTS: ") + e.what());
        return nullptr;
    }
}

POCKET_TTS_API void pocket_tts_destroy(PocketTTSHandle handle) {
    if (handle) delete tts_of(handle);
}

POCKET_TTS_API VoiceHandle pocket_tts_encode_voice(PocketTTSHandle handle, const char* audio_path) {
    if (!handle || !audio_path) {
        set_error("Invalid handle or audio path");
        return nullptr;
    }
    try {
        auto* tts = tts_of(handle);
        return make_voice(tts->encodeVoice(audio_path), tts->engine().spec().d_model);
    } catch (const std::exception& e) {
        set_error(std::string("Failed to encode voice: ") + e.what());
        return nullptr;
    }
}

POCKET_TTS_API int pocket_tts_encode_voices(PocketTTSHandle handle, int n, const char* const* audio_paths, VoiceHandle* voices_out) {
    if (!handle || n < 0 || (n > 0 && (!audio_paths || !voices_out))) { set_error("Invalid parameters"); return -1; }
    for (int i = 0; i < n; ++i) {
        voices_out[i] = nullptr;
        if (!audio_paths[i]) { set_error("Invalid audio path"); return -1; }
    }
    return guarded("Failed to encode voices: ", [&] {
        auto* tts = tts_of(handle);
        std::vector<std::string> paths(audio_paths, audio_paths + n);
        auto emb = tts->encodeVoices(paths);
        for (int i = 0; i < n; ++i) voices_out[i] = make_voice(std::move(emb[i]), tts->engine().spec().d_model);
        return 0;
    });
}

POCKET_TTS_API VoiceHandle pocket_tts_encode_voice_from_samples(PocketTTSHandle handle, const float* audio_data, int sample_count,
                                                                int sample_rate) {
    if (!handle || !audio_data || sample_count <= 0) {
        set_error("Invalid parameters");
        return nullptr;
    }
    try {
        auto* tts = tts_of(handle);
        return make_voice(tts->encodeVoiceFromSamples(audio_data, sample_count, sample_rate), tts->engine().spec().d_model);
    } catch (const std::exception& e) {
        set_error(std::string("Failed to encode voice from samples: ") + e.what());
        return nullptr;
    }
}

POCKET_TTS_API void pocket_tts_free_voice(VoiceHandle voice) {
    if (voice) delete static_cast<VoiceData*>(voice);
}

POCKET_TTS_API int pocket_tts_generate(PocketTTSHandle handle, const char* text, VoiceHandle voice, AudioResult* result) {
    if (!handle || !text || !voice || !result) {
        set_error("Invalid parameters");
        return -1;
    }
    return guarded("Failed to generate: ", [&] {
        auto* tts = tts_of(handle);
        auto* v = static_cast<VoiceData*>(voice);
        auto audio = tts->generateWithEmbeddings(text, v->embeddings, v->shape);
        fill_result(result, audio, tts->engine().spec().sample_rate);
        return 0;
    });
}

POCKET_TTS_API void pocket_tts_free_audio(AudioResult* result) {
    if (result && result->data) {
        delete[] result->data;
        result->data = nullptr;
        result->sample_count = 0;
    }
}

POCKET_TTS_API const char* pocket_tts_get_last_error(void) { return g_last_error.c_str(); }

POCKET_TTS_API int pocket_tts_generate_streaming(PocketTTSHandle handle, const char* text, VoiceHandle voice,
                                                 AudioChunkCallbackC callback, const StreamingConfig* config) {
    if (!handle || !text || !voice || !callback) {
        set_error("Invalid parameters");
        return -1;
    }
    return guarded("Streaming failed: ", [&] {
        auto* tts = tts_of(handle);
        auto* v = static_cast<VoiceData*>(voice);
        void* user = config ? config->user_data : nullptr;
        pocket_tts::StreamingConfig sc;
        if (config && config->chunk_size_frames > 0) sc.chunkSizeFrames = config->chunk_size_frames;
        sc.enableCancellation = true;
        return tts->generateStreaming(
            text, v->embeddings, v->shape,
            [callback, user](const float* s, int n, bool fin) { callback(s, n, fin ? 1 : 0, user); }, sc);
    });
}

POCKET_TTS_API void pocket_tts_cancel_streaming(PocketTTSHandle handle) {
    if (!handle) return;
    try {
        tts_of(handle)->cancelStreaming();
    } catch (...) {
    }
}

POCKET_TTS_API const char* pocket_tts_version(void) { return "1.0.0"; }

// ------------------------------------------------------------------ additive entry points

POCKET_TTS_API int pocket_tts_dims(PocketTTSHandle handle, int* dims) {
    if (!handle || !dims) { set_error("Invalid parameters"); return -1; }
    return guarded("dims: ", [&] {
        auto& e = tts_of(handle)->engine();
        const auto& s = e.spec();
        dims[0] = s.latent; dims[1] = s.d_model; dims[2] = s.frame_size(); dims[3] = s.vocab; dims[4] = s.mimi_dim;
        dims[5] = s.sample_rate; dims[6] = e.options().lsd_steps; dims[7] = e.options().max_frames;
        return 0;
    });
}

POCKET_TTS_API VoiceHandle pocket_tts_voice_from_embeddings(const float* embeddings, int frames, int d_model) {
    if (!embeddings || frames < 0 || d_model <= 0) { set_error("Invalid parameters"); return nullptr; }
    return make_voice(std::vector<float>(embeddings, embeddings + static_cast<size_t>(frames) * d_model), d_model);
}

POCKET_TTS_API int pocket_tts_voice_frames(VoiceHandle voice) {
    return voice ? static_cast<int>(static_cast<VoiceData*>(voice)->shape[1]) : -1;
}

POCKET_TTS_API const float* pocket_tts_voice_data(VoiceHandle voice) {
    return voice ? static_cast<VoiceData*>(voice)->embeddings.data() : nullptr;
}

POCKET_TTS_API int pocket_tts_tokenize(PocketTTSHandle handle, const char* text, long long* ids, int cap) {
    if (!handle || !text) { set_error("Invalid parameters"); return -1; }
    return guarded("tokenize: ", [&] {
        auto v = tts_of(handle)->tokenize(text);
        for (int i = 0; i < cap && i < static_cast<int>(v.size()); ++i) ids[i] = v[i];
        return static_cast<int>(v.size());
    });
}

POCKET_TTS_API int pocket_tts_generate_batch(PocketTTSHandle handle, int n, const char* const* texts, const VoiceHandle* voices,
                                             AudioResult* results) {
    if (!handle || n < 0 || (n > 0 && (!texts || !voices || !results))) { set_error("Invalid parameters"); return -1; }
    return guarded("Failed to generate: ", [&] {
        auto* tts = tts_of(handle);
        std::vector<ptts::Utterance> utts(n);
        for (int i = 0; i < n; ++i) {
            if (!texts[i] || !voices[i]) throw std::runtime_error("null text or voice in batch");
            auto* v = static_cast<VoiceData*>(voices[i]);
            utts[i].token_ids = tts->tokenize(texts[i]);
            utts[i].voice = v->embeddings.data();
            utts[i].voice_frames = static_cast<int>(v->shape[1]);
        }
        std::vector<ptts::UtteranceResult> res;
        tts->generateQueue(utts, res, false);   // slot table(s): any n, finished slots are refilled from the rest
        for (int i = 0; i < n; ++i) fill_result(&results[i], res[i].audio, tts->engine().spec().sample_rate);
        return 0;
    });
}

static int generate_ids_impl(PocketTTSHandle handle, int n, const long long* const* ids, const int* n_ids, const float* const* voices,
                             const int* voice_frames, const float* const* noise, AudioResult* results, float* const* latents_out,
                             float* const* eos_out, int* frames_out, int* eos_count_out, bool queued) {
    if (!handle || n < 0 || (n > 0 && (!ids || !n_ids || !voices || !voice_frames))) { set_error("Invalid parameters"); return -1; }
    return guarded("Failed to generate: ", [&] {
        auto* tts = tts_of(handle);
        std::vector<ptts::Utterance> utts(n);
        bool want = latents_out || eos_out;
        for (int i = 0; i < n; ++i) {
            utts[i].token_ids.assign(ids[i], ids[i] + n_ids[i]);
            utts[i].voice = voices[i];
            utts[i].voice_frames = voice_frames[i];
            utts[i].noise = noise ? noise[i] : nullptr;
        }
        std::vector<ptts::UtteranceResult> res;
        if (queued) {
            tts->generateQueue(utts, res, want);
        } else if (n == 1) {   // single utterances share the slot table with concurrent callers, like pocket_tts_generate
            res.resize(1);
            tts->pickEngine().generate_one(utts[0], res[0], want);
        } else {
            tts->engine().generate(utts, res, want);
        }
        // the caller owns a copy of every utterance's samples (pocket_tts_free_audio): about 1 MB per 10 s of audio, copied on
        // several threads when there are many
        const int sr = tts->engine().spec().sample_rate;
        const int copiers = results ? std::min({n / 8, 8, static_cast<int>(std::max(1u, std::thread::hardware_concurrency()))}) : 0;
        if (copiers > 1) {
            std::vector<std::thread> pool;
            std::vector<std::exception_ptr> failed(copiers);
            for (int i = 0; i < n; ++i) results[i].data = nullptr;
            for (int w = 0; w < copiers; ++w)
                pool.emplace_back([&, w] {
                    try { for (int i = w; i < n; i += copiers) fill_result(&results[i], res[i].audio, sr); }
                    catch (...) { failed[w] = std::current_exception(); }
                });
            for (auto& t : pool) t.join();
            for (auto& f : failed)
                if (f) {
                    for (int i = 0; i < n; ++i) { delete[] results[i].data; results[i].data = nullptr; results[i].sample_count = 0; }
                    std::rethrow_exception(f);
                }
        }
        for (int i = 0; i < n; ++i) {
            if (results && copiers <= 1) fill_result(&results[i], res[i].audio, sr);
            if (frames_out) frames_out[i] = res[i].frames;
            if (latents_out && latents_out[i] && !res[i].latents.empty())
                std::memcpy(latents_out[i], res[i].latents.data(), res[i].latents.size() * 4);
            if (eos_out && eos_out[i] && !res[i].eos.empty()) std::memcpy(eos_out[i], res[i].eos.data(), res[i].eos.size() * 4);
            if (eos_count_out) eos_count_out[i] = static_cast<int>(res[i].eos.size());
        }
        return 0;
    });
}

POCKET_TTS_API int pocket_tts_generate_ids(PocketTTSHandle handle, int n, const long long* const* ids, const int* n_ids,
                                           const float* const* voices, const int* voice_frames, const float* const* noise,
                                           AudioResult* results, float* const* latents_out, float* const* eos_out, int* frames_out,
                                           int* eos_count_out) {
    return generate_ids_impl(handle, n, ids, n_ids, voices, voice_frames, noise, results, latents_out, eos_out, frames_out, eos_count_out, false);
}

POCKET_TTS_API int pocket_tts_generate_ids_queued(PocketTTSHandle handle, int n, const long long* const* ids, const int* n_ids,
                                                  const float* const* voices, const int* voice_frames, const float* const* noise,
                                                  AudioResult* results, float* const* latents_out, float* const* eos_out, int* frames_out,
                                                  int* eos_count_out) {
    return generate_ids_impl(handle, n, ids, n_ids, voices, voice_frames, noise, results, latents_out, eos_out, frames_out, eos_count_out, true);
}

POCKET_TTS_API void* pocket_tts_stage_ids(PocketTTSHandle handle, int n, const long long* const* ids, const int* n_ids,
                                          const float* const* voices, const int* voice_frames, const float* const* noise) {
    if (!handle || n <= 0 || !ids || !n_ids || !voices || !voice_frames) { set_error("Invalid parameters"); return nullptr; }
    try {
        std::vector<ptts::Utterance> utts(n);
        for (int i = 0; i < n; ++i) {
            utts[i].token_ids.assign(ids[i], ids[i] + n_ids[i]);
            utts[i].voice = voices[i];
            utts[i].voice_frames = voice_frames[i];
            utts[i].noise = noise ? noise[i] : nullptr;
        }
        return tts_of(handle)->engine().stage(utts);
    } catch (const std::exception& e) {
        set_error(std::string("stage: ") + e.what());
        return nullptr;
    }
}

POCKET_TTS_API int pocket_tts_run_staged(PocketTTSHandle handle, void* staged, int* frames_out) {
    if (!handle || !staged || !frames_out) { set_error("Invalid parameters"); return -1; }
    return guarded("run_staged: ", [&] {
        auto* b = static_cast<ptts::Job*>(staged);
        std::vector<int> frames;
        tts_of(handle)->engine().run_staged(*b, frames, nullptr);
        for (size_t i = 0; i < frames.size(); ++i) frames_out[i] = frames[i];
        return 0;
    });
}

POCKET_TTS_API int pocket_tts_fetch_staged(PocketTTSHandle handle, void* staged, const int* frames, AudioResult* results,
                                           float* const* latents_out) {
    if (!handle || !staged || !frames) { set_error("Invalid parameters"); return -1; }
    return guarded("fetch_staged: ", [&] {
        auto* b = static_cast<ptts::Job*>(staged);
        auto& e = tts_of(handle)->engine();
        std::vector<int> fr(frames, frames + b->B);
        for (int f : fr)
            if (f < 0 || f > e.options().max_frames) throw std::runtime_error("frame count out of range");
        std::vector<ptts::UtteranceResult> res;
        e.fetch_staged(*b, fr, res, latents_out != nullptr);
        for (int i = 0; i < b->B; ++i) {
            if (results) fill_result(&results[i], res[i].audio, e.spec().sample_rate);
            if (latents_out && latents_out[i] && !res[i].latents.empty())
                std::memcpy(latents_out[i], res[i].latents.data(), res[i].latents.size() * 4);
        }
        return 0;
    });
}

POCKET_TTS_API int pocket_tts_streaming_ids(PocketTTSHandle handle, const long long* ids, int n_ids, const float* voice,
                                            int voice_frames, const float* noise, int chunk_size_frames,
                                            AudioChunkCallbackC callback, void* user_data) {
    if (!handle || !ids || !voice || !callback) { set_error("Invalid parameters"); return -1; }
    return guarded("Streaming failed: ", [&] {
        ptts::Utterance u;
        u.token_ids.assign(ids, ids + n_ids);
        u.voice = voice;
        u.voice_frames = voice_frames;
        u.noise = noise;
        return tts_of(handle)->pickEngine().generate_streaming(
            u, chunk_size_frames > 0 ? chunk_size_frames : 5,
            [callback, user_data](const float* s, int n, bool fin) { callback(s, n, fin ? 1 : 0, user_data); }, nullptr, false, nullptr,
            nullptr);
    });
}

POCKET_TTS_API int ptts_text_conditioner(PocketTTSHandle handle, const long long* ids, int T, float* out) {
    if (!handle || !ids || !out) { set_error("Invalid parameters"); return -1; }
    return guarded("text_conditioner: ", [&] {
        tts_of(handle)->engine().op_text_conditioner(reinterpret_cast<const int64_t*>(ids), T, out);
        return 0;
    });
}

POCKET_TTS_API void* ptts_state_new(PocketTTSHandle handle, int batch) {
    if (!handle || batch <= 0) { set_error("Invalid parameters"); return nullptr; }
    try {
        return tts_of(handle)->engine().batch_new(batch);
    } catch (const std::exception& e) {
        set_error(std::string("state_new: ") + e.what());
        return nullptr;
    }
}

POCKET_TTS_API void ptts_state_free(PocketTTSHandle handle, void* state) {
    if (handle && state) tts_of(handle)->engine().batch_free(static_cast<ptts::Batch*>(state));
}

POCKET_TTS_API int ptts_state_reset(PocketTTSHandle handle, void* state) {
    if (!handle || !state) { set_error("Invalid parameters"); return -1; }
    return guarded("state_reset: ", [&] {
        tts_of(handle)->engine().batch_reset(static_cast<ptts::Batch*>(state));
        return 0;
    });
}

POCKET_TTS_API int ptts_lm_prefill(PocketTTSHandle handle, void* state, int slot, const float* emb, int rows) {
    if (!handle || !state || !emb || rows < 0) { set_error("Invalid parameters"); return -1; }
    return guarded("lm_prefill: ", [&] {
        tts_of(handle)->engine().op_lm_prefill(static_cast<ptts::Batch*>(state), slot, emb, rows);
        return 0;
    });
}

POCKET_TTS_API int ptts_lm_step(PocketTTSHandle handle, void* state, const float* sequence, float* cond, float* eos) {
    if (!handle || !state || !sequence || !cond || !eos) { set_error("Invalid parameters"); return -1; }
    return guarded("lm_step: ", [&] {
        tts_of(handle)->engine().op_lm_step(static_cast<ptts::Batch*>(state), sequence, cond, eos);
        return 0;
    });
}

POCKET_TTS_API int ptts_flow_euler(PocketTTSHandle handle, const float* cond, const float* x0, int rows, float* x) {
    if (!handle || !cond || !x0 || !x || rows <= 0) { set_error("Invalid parameters"); return -1; }
    return guarded("flow_euler: ", [&] {
        tts_of(handle)->engine().op_flow_euler(cond, x0, rows, x);
        return 0;
    });
}

POCKET_TTS_API int ptts_flow_euler_noalloc(PocketTTSHandle handle, const float* cond_dev, const float* x0_dev, int rows, float* x_dev) {
    if (!handle || !cond_dev || !x0_dev || !x_dev || rows <= 0) { set_error("Invalid parameters"); return -1; }
    return guarded("flow_euler_noalloc: ", [&] {
        tts_of(handle)->engine().op_flow_euler_static(cond_dev, x0_dev, rows, x_dev);
        return 0;
    });
}

POCKET_TTS_API unsigned long long ptts_flow_euler_noalloc_capacity(void) { return ptts::Engine::static_scratch_capacity(); }

POCKET_TTS_API int ptts_mimi_decode(PocketTTSHandle handle, void* state, const float* latents, int n, float* audio) {
    if (!handle || !state || !latents || !audio) { set_error("Invalid parameters"); return -1; }
    return guarded("mimi_decode: ", [&] {
        tts_of(handle)->engine().op_mimi_decode(static_cast<ptts::Batch*>(state), latents, n, audio);
        return 0;
    });
}

POCKET_TTS_API int ptts_mimi_encode(PocketTTSHandle handle, const float* audio, int samples, float* out, int cap_frames) {
    if (!handle || !audio || !out || samples <= 0 || cap_frames < 0) { set_error("Invalid parameters"); return -1; }
    return guarded("mimi_encode: ", [&] {
        auto& e = tts_of(handle)->engine();
        auto emb = e.encode_voice(audio, samples);
        const int D = e.spec().d_model, frames = static_cast<int>(emb.size()) / D;
        std::memcpy(out, emb.data(), static_cast<size_t>(std::min(frames, cap_frames)) * D * 4);
        return frames;
    });
}

POCKET_TTS_API int ptts_mimi_encode_batch(PocketTTSHandle handle, const float* const* clips, int n_clips, int samples, float* const* outs,
                                          int cap_frames) {
    if (!handle || !clips || !outs || n_clips <= 0 || samples <= 0 || cap_frames < 0) { set_error("Invalid parameters"); return -1; }
    return guarded("mimi_encode_batch: ", [&] {
        auto& e = tts_of(handle)->engine();
        std::vector<const float*> ptrs(clips, clips + n_clips);
        std::vector<int> lens(static_cast<size_t>(n_clips), samples);
        std::vector<std::vector<float>> emb;
        e.encode_voice_batch(ptrs, lens, emb);
        const int D = e.spec().d_model;
        int frames = 0;
        for (int i = 0; i < n_clips; ++i) {
            frames = static_cast<int>(emb[i].size()) / D;
            std::memcpy(outs[i], emb[i].data(), static_cast<size_t>(std::min(frames, cap_frames)) * D * 4);
        }
        return frames;
    });
}

POCKET_TTS_API void* ptts_tokenizer_new(const char* model_path) {
    if (!model_path) { set_error("Invalid parameters"); return nullptr; }
    try {
        return new pocket_tts::Tokenizer(model_path);
    } catch (const std::exception& e) {
        set_error(e.what());
        return nullptr;
    }
}

POCKET_TTS_API void ptts_tokenizer_free(void* tokenizer) { delete static_cast<pocket_tts::Tokenizer*>(tokenizer); }

POCKET_TTS_API int ptts_tokenizer_vocab(void* tokenizer) {
    return tokenizer ? static_cast<pocket_tts::Tokenizer*>(tokenizer)->vocabSize() : -1;
}

POCKET_TTS_API int ptts_tokenizer_encode(void* tokenizer, const char* text, int raw, long long* ids, int cap) {
    if (!tokenizer || !text) { set_error("Invalid parameters"); return -1; }
    return guarded("tokenize: ", [&] {
        auto* t = static_cast<pocket_tts::Tokenizer*>(tokenizer);
        auto v = raw ? t->encodeRaw(text) : t->encode(text);
        for (int i = 0; i < cap && i < static_cast<int>(v.size()); ++i) ids[i] = v[i];
        return static_cast<int>(v.size());
    });
}

static int copy_out(const std::vector<float>& v, float* out, int cap) {
    const int n = static_cast<int>(v.size());
    if (out && cap > 0) std::memcpy(out, v.data(), sizeof(float) * static_cast<size_t>(std::min(n, cap)));
    return n;
}

POCKET_TTS_API int ptts_audio_load_wav(const char* path, int target_rate, float* out, int cap) {
    if (!path) { set_error("Invalid parameters"); return -1; }
    return guarded("", [&] { return copy_out(pocket_tts::AudioUtils::loadWav(path, target_rate), out, cap); });
}

POCKET_TTS_API int ptts_audio_save_wav(const char* path, const float* samples, int n, int rate) {
    if (!path || (!samples && n > 0) || n < 0) { set_error("Invalid parameters"); return -1; }
    return guarded("", [&] {
        pocket_tts::AudioUtils::saveWav(path, std::vector<float>(samples, samples + n), rate);
        return 0;
    });
}

POCKET_TTS_API int ptts_audio_resample(const float* in, int n, int in_rate, int out_rate, float* out, int cap) {
    if ((!in && n > 0) || n < 0) { set_error("Invalid parameters"); return -1; }
    return guarded("", [&] {
        return copy_out(pocket_tts::AudioUtils::resample(std::vector<float>(in, in + n), in_rate, out_rate), out, cap);
    });
}

POCKET_TTS_API int ptts_audio_normalize(const float* in, int n, float* out) {
    if ((!in && n > 0) || n < 0) { set_error("Invalid parameters"); return -1; }
    return guarded("", [&] { return copy_out(pocket_tts::AudioUtils::normalize(std::vector<float>(in, in + n)), out, n); });
}

POCKET_TTS_API int ptts_phase_timing(PocketTTSHandle handle, int enable) {
    if (!handle) { set_error("Invalid parameters"); return -1; }
    return guarded("phase_timing: ", [&] {
        tts_of(handle)->engine().enable_phase_timing(enable != 0);
        return 0;
    });
}

POCKET_TTS_API int ptts_phase_times(PocketTTSHandle handle, double* times_ms, long* calls) {
    if (!handle || !times_ms || !calls) { set_error("Invalid parameters"); return -1; }
    return guarded("phase_times: ", [&] {
        auto t = tts_of(handle)->engine().phase_times();
        times_ms[0] = t.lm_ms; times_ms[1] = t.flow_ms; times_ms[2] = t.dec_ms; times_ms[3] = t.head_ms;
        calls[0] = t.lm_calls; calls[1] = t.flow_calls; calls[2] = t.dec_calls; calls[3] = t.head_calls;
        return 0;
    });
}

POCKET_TTS_API int ptts_q8_linear(PocketTTSHandle handle, const float* A, int M, int K, const float* W, int N, float* out) {
    if (!handle || !A || !W || !out || M <= 0 || N <= 0 || K <= 0) { set_error("Invalid parameters"); return -1; }
    return guarded("q8_linear: ", [&] {
        tts_of(handle)->engine().q8_linear(A, M, K, W, N, out);
        return 0;
    });
}

POCKET_TTS_API int ptts_q8_linear_dyn(PocketTTSHandle handle, const float* A, int M, int K, const float* W, int N, float* out) {
    if (!handle || !A || !W || !out || M <= 0 || N <= 0 || K <= 0) { set_error("Invalid parameters"); return -1; }
    return guarded("q8_linear_dyn: ", [&] {
        tts_of(handle)->engine().q8_linear(A, M, K, W, N, out, nullptr, nullptr, 1);
        return 0;
    });
}

POCKET_TTS_API int ptts_q8_linear_q(PocketTTSHandle handle, const float* A, int M, int K, const signed char* q, const float* scale, int N,
                                    float* out) {
    if (!handle || !A || !q || !scale || !out || M <= 0 || N <= 0 || K <= 0) { set_error("Invalid parameters"); return -1; }
    return guarded("q8_linear_q: ", [&] {
        tts_of(handle)->engine().q8_linear(A, M, K, nullptr, N, out, q, scale);
        return 0;
    });
}

POCKET_TTS_API int ptts_bf16_linear(PocketTTSHandle handle, const float* A, int M, int K, const float* W, int N, const float* bias,
                                    const float* colscale, const float* R, int act, int flags, float* out) {
    if (!handle || !A || !W || !out || M <= 0 || N <= 0 || K <= 0) { set_error("Invalid parameters"); return -1; }
    return guarded("bf16_linear: ", [&] {
        tts_of(handle)->engine().bf16_linear(A, M, K, W, N, bias, colscale, R, act, flags, out);
        return 0;
    });
}

POCKET_TTS_API int ptts_scheduler_stats(PocketTTSHandle handle, long* stats) {
    if (!handle || !stats) { set_error("Invalid parameters"); return -1; }
    return guarded("scheduler_stats: ", [&] {
        auto st = tts_of(handle)->engine().scheduler_stats();
        stats[0] = st.batches; stats[1] = st.utterances; stats[2] = st.largest;
        return 0;
    });
}

POCKET_TTS_API int ptts_scheduler_stats_ex(PocketTTSHandle handle, long* stats) {
    if (!handle || !stats) { set_error("Invalid parameters"); return -1; }
    return guarded("scheduler_stats: ", [&] {
        for (int k = 0; k < 8; ++k) stats[k] = 0;
        auto* tts = tts_of(handle);
        for (int e = 0; e < tts->engineCount(); ++e) {   // summed over the handle's engines (largest: the maximum)
            auto st = tts->engineAt(e).scheduler_stats();
            stats[0] += st.batches; stats[1] += st.utterances; stats[2] = std::max(stats[2], st.largest); stats[3] += st.frame_steps;
            stats[4] += st.slot_steps; stats[5] += st.frames; stats[6] += st.decode_calls; stats[7] += st.admissions;
        }
        return 0;
    });
}

/* Engines behind the handle (POCKET_TTS_DEVICES) and, per engine, the utterances it has served: served[i] for i < cap. */
POCKET_TTS_API int ptts_engine_count(PocketTTSHandle handle, long* served, int cap) {
    if (!handle) { set_error("Invalid parameters"); return -1; }
    return guarded("engine_count: ", [&] {
        auto* tts = tts_of(handle);
        for (int e = 0; e < tts->engineCount() && served && e < cap; ++e) served[e] = tts->engineAt(e).scheduler_stats().utterances;
        return tts->engineCount();
    });
}

POCKET_TTS_API int ptts_weight_bytes(PocketTTSHandle handle, unsigned long long* bytes) {
    if (!handle || !bytes) { set_error("Invalid parameters"); return -1; }
    return guarded("weight_bytes: ", [&] {
        const auto& m = tts_of(handle)->engine().model();
        bytes[0] = m.gemm_weight_bytes("lm.");
        bytes[1] = m.gemm_weight_bytes("flow.");
        bytes[2] = m.gemm_weight_bytes("dec.");
        bytes[3] = m.gemm_weight_bytes("enc.");
        return 0;
    });
}

}  // extern "C"

// ------------------------------------------------------------------ weights inside an ONNX file (host only)

POCKET_TTS_API void* ptts_onnx_open(const char* path) {
    if (!path) { set_error("Invalid parameters"); return nullptr; }
    try {
        return new ptts::OnnxFile(path);
    } catch (const std::exception& e) {
        set_error(std::string("Failed to read ONNX file: ") + e.what());
        return nullptr;
    }
}

POCKET_TTS_API void ptts_onnx_close(void* onnx) { delete static_cast<ptts::OnnxFile*>(onnx); }

POCKET_TTS_API int ptts_onnx_tensor_count(void* onnx) {
    return onnx ? static_cast<int>(static_cast<ptts::OnnxFile*>(onnx)->tensors().size()) : -1;
}

POCKET_TTS_API int ptts_onnx_tensor_info(void* onnx, int index, const char** name, int* dtype, int* rank, long long* dims8) {
    if (!onnx) { set_error("Invalid parameters"); return -1; }
    const auto& ts = static_cast<ptts::OnnxFile*>(onnx)->tensors();
    if (index < 0 || index >= static_cast<int>(ts.size())) { set_error("ONNX tensor index out of range"); return -1; }
    const auto& t = ts[index];
    if (t.dims.size() > 8) { set_error("ONNX tensor rank above 8"); return -1; }
    if (name) *name = t.name.c_str();
    if (dtype) *dtype = t.dtype;
    if (rank) *rank = static_cast<int>(t.dims.size());
    if (dims8) for (size_t d = 0; d < t.dims.size(); ++d) dims8[d] = t.dims[d];
    return 0;
}

POCKET_TTS_API int ptts_onnx_find(void* onnx, const char* name) {
    if (!onnx || !name) return -1;
    return static_cast<ptts::OnnxFile*>(onnx)->find(name);
}

POCKET_TTS_API long long ptts_onnx_read_f32(void* onnx, int index, float* out, long long cap) {
    if (!onnx || !out) { set_error("Invalid parameters"); return -1; }
    long long n = -1;
    guarded("Failed to read ONNX tensor: ", [&] {
        const auto v = static_cast<ptts::OnnxFile*>(onnx)->read_f32(index);
        if (static_cast<long long>(v.size()) > cap) throw std::runtime_error("output buffer too small");
        std::memcpy(out, v.data(), v.size() * 4);
        n = static_cast<long long>(v.size());
        return 0;
    });
    return n;
}

POCKET_TTS_API long long ptts_onnx_dequantize(void* onnx, const char* base, int axis, float* out, long long cap, int* rank, long long* dims8) {
    if (!onnx || !base || !out) { set_error("Invalid parameters"); return -1; }
    long long n = -1;
    guarded("Failed to dequantise ONNX tensor: ", [&] {
        std::vector<int64_t> dims;
        const auto v = static_cast<ptts::OnnxFile*>(onnx)->dequantize(base, axis, &dims);
        if (static_cast<long long>(v.size()) > cap) throw std::runtime_error("output buffer too small");
        if (dims.size() > 8) throw std::runtime_error("rank above 8");
        std::memcpy(out, v.data(), v.size() * 4);
        if (rank) *rank = static_cast<int>(dims.size());
        if (dims8) for (size_t d = 0; d < dims.size(); ++d) dims8[d] = dims[d];
        n = static_cast<long long>(v.size());
        return 0;
    });
    return n;
}

POCKET_TTS_API int ptts_onnx_node_count(void* onnx) {
    return onnx ? static_cast<int>(static_cast<ptts::OnnxFile*>(onnx)->nodes().size()) : -1;
}

// op type and the comma-joined input / output names of node `index`, written into caller buffers (truncated to cap - 1)
POCKET_TTS_API int ptts_onnx_node_info(void* onnx, int index, char* op_type, char* inputs, char* outputs, int cap) {
    if (!onnx || cap <= 0) { set_error("Invalid parameters"); return -1; }
    const auto& ns = static_cast<ptts::OnnxFile*>(onnx)->nodes();
    if (index < 0 || index >= static_cast<int>(ns.size())) { set_error("ONNX node index out of range"); return -1; }
    auto put = [&](char* dst, const std::string& v) {
        if (!dst) return;
        const size_t n = std::min(v.size(), static_cast<size_t>(cap - 1));
        std::memcpy(dst, v.data(), n);
        dst[n] = 0;
    };
    auto join = [](const std::vector<std::string>& v) {
        std::string o;
        for (size_t i = 0; i < v.size(); ++i) { if (i) o += ','; o += v[i]; }
        return o;
    };
    put(op_type, ns[index].op_type);
    put(inputs, join(ns[index].inputs));
    put(outputs, join(ns[index].outputs));
    return 0;
}
