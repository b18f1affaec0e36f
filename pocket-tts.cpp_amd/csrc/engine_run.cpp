// Orchestration on top of the forward passes: prefill, the autoregressive frame loop, offline and
// streaming decode schedules, the voice encoder and the graph-level entry points.
// The loop bodies restate /root/reference/src/pocket_tts.cpp:548-657 (offline) and :686-979 (streaming).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "engine.h"

namespace ptts {

static constexpr int kDecChunk = 15;        // src/pocket_tts.cpp:461
static constexpr float kEosThreshold = -4.0f;  // src/pocket_tts.cpp:596

void Engine::op_text_conditioner(const int64_t* ids, int T, float* out) {
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    const Spec& s = spec();
    if (T <= 0) return;
    DevBuf dids(static_cast<size_t>(T) * 8), dout(static_cast<size_t>(T) * s.d_model * 4);
    PTTS_HIP(hipMemcpyAsync(dids.p, ids, static_cast<size_t>(T) * 8, hipMemcpyHostToDevice, stream_));
    launch_embed_rows(dids.as<long long>(), T, hd_.text_embed, s.vocab + 1, s.d_model, dout.f(), stream_);
    PTTS_HIP(hipMemcpyAsync(out, dout.p, dout.bytes, hipMemcpyDeviceToHost, stream_));
    PTTS_HIP(hipStreamSynchronize(stream_));
}

// Voice rows first, then text rows, as two passes like the reference's two prefill Runs (:572, :575).
void Engine::prefill_utterance(Batch& b, int slot, const Utterance& u) {
    const Spec& s = spec();
    const int D = s.d_model;
    if (u.token_ids.empty()) throw std::runtime_error("Text cannot be empty");
    if (static_cast<int>(u.token_ids.size()) > opt_.max_text_tokens) throw std::runtime_error("text has more tokens than the engine was sized for");
    const int chunk = 256;
    for (int r0 = 0; r0 < u.voice_frames; r0 += chunk) {
        int n = std::min(chunk, u.voice_frames - r0);
        PTTS_HIP(hipMemcpyAsync(b.x.p, u.voice + static_cast<size_t>(r0) * D, static_cast<size_t>(n) * D * 4, hipMemcpyHostToDevice, stream_));
        lm_rows(b, slot, 1, n, b.x.f());
    }
    const int T = static_cast<int>(u.token_ids.size());
    PTTS_HIP(hipMemcpyAsync(b.ids.p, u.token_ids.data(), static_cast<size_t>(T) * 8, hipMemcpyHostToDevice, stream_));
    for (int r0 = 0; r0 < T; r0 += chunk) {
        int n = std::min(chunk, T - r0);
        launch_embed_rows(b.ids.as<long long>() + r0, n, hd_.text_embed, s.vocab + 1, D, b.x.f(), stream_);
        lm_rows(b, slot, 1, n, b.x.f());
    }
}

// Start point of the Euler integration for this frame: zeros at temperature 0, else N(0, sqrt(T)) per
// element from a fresh distribution object each frame (src/pocket_tts.cpp:606-613).
void Engine::fill_noise(std::vector<std::mt19937>& rngs, const std::vector<Utterance>& utts, int step, float* host_x0) {
    const int L = spec().latent;
    for (size_t i = 0; i < utts.size(); ++i) {
        float* x = host_x0 + i * L;
        if (utts[i].noise) {
            std::memcpy(x, utts[i].noise + static_cast<size_t>(step) * L, static_cast<size_t>(L) * 4);
        } else if (opt_.temperature > 0) {
            std::normal_distribution<float> dist(0.0f, std::sqrt(opt_.temperature));
            for (int k = 0; k < L; ++k) x[k] = dist(rngs[i]);
        } else {
            std::memset(x, 0, static_cast<size_t>(L) * 4);
        }
    }
}

// Upload everything a batch needs (conditioning rows, token ids, start points) so that run_staged()
// touches host memory only for the per-step EOS logits.
Batch* Engine::stage(const std::vector<Utterance>& utts) {
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    const Spec& s = spec();
    const int B = static_cast<int>(utts.size()), L = s.latent, D = s.d_model;
    if (B == 0) throw std::runtime_error("empty batch");
    Batch& b = *acquire_batch(B);
    int vmax = 1;
    for (auto& u : utts) {
        if (u.token_ids.empty()) throw std::runtime_error("Text cannot be empty");
        if (static_cast<int>(u.token_ids.size()) > opt_.max_text_tokens) throw std::runtime_error("text has more tokens than the engine was sized for");
        vmax = std::max(vmax, u.voice_frames);
    }
    b.st_vmax = vmax;
    if (b.voice_stage.bytes < static_cast<size_t>(B) * vmax * D * 4) b.voice_stage.alloc(static_cast<size_t>(B) * vmax * D * 4);
    if (b.ids_stage.bytes < static_cast<size_t>(B) * opt_.max_text_tokens * 8) b.ids_stage.alloc(static_cast<size_t>(B) * opt_.max_text_tokens * 8);
    b.st_voice_frames.assign(B, 0);
    b.st_ntok.assign(B, 0);
    bool any_noise = opt_.temperature > 0;
    for (auto& u : utts) any_noise = any_noise || u.noise;
    b.st_has_noise = any_noise;
    for (int i = 0; i < B; ++i) {
        const Utterance& u = utts[i];
        b.st_voice_frames[i] = u.voice_frames;
        b.st_ntok[i] = static_cast<int>(u.token_ids.size());
        if (u.voice_frames)
            PTTS_HIP(hipMemcpyAsync(b.voice_stage.f() + static_cast<size_t>(i) * vmax * D, u.voice, static_cast<size_t>(u.voice_frames) * D * 4,
                                    hipMemcpyHostToDevice, stream_));
        PTTS_HIP(hipMemcpyAsync(b.ids_stage.as<long long>() + static_cast<size_t>(i) * opt_.max_text_tokens, u.token_ids.data(),
                                u.token_ids.size() * 8, hipMemcpyHostToDevice, stream_));
    }
    if (any_noise) {
        // Start points for every frame: explicit ones, else N(0, sqrt(T)) from a fresh distribution object per
        // frame and utterance (src/pocket_tts.cpp:606-613), drawn up front so the frame loop stays on the device.
        const size_t per = static_cast<size_t>(opt_.max_frames) * L;
        if (b.noise_all.bytes < static_cast<size_t>(B) * per * 4) b.noise_all.alloc(static_cast<size_t>(B) * per * 4);
        std::vector<float> host(static_cast<size_t>(B) * per, 0.f);
        for (int i = 0; i < B; ++i) {
            float* dst = host.data() + static_cast<size_t>(i) * per;
            if (utts[i].noise) {
                std::memcpy(dst, utts[i].noise, per * 4);
            } else if (opt_.temperature > 0) {
                std::mt19937 gen(rng_());
                for (int f = 0; f < opt_.max_frames; ++f) {
                    std::normal_distribution<float> dist(0.0f, std::sqrt(opt_.temperature));
                    for (int k = 0; k < L; ++k) dst[static_cast<size_t>(f) * L + k] = dist(gen);
                }
            }
        }
        PTTS_HIP(hipMemcpyAsync(b.noise_all.p, host.data(), host.size() * 4, hipMemcpyHostToDevice, stream_));
        PTTS_HIP(hipStreamSynchronize(stream_));
    }
    const size_t audio_need = static_cast<size_t>(B) * opt_.max_frames * s.frame_size() * 4;
    if (b.audio_all.bytes < audio_need) b.audio_all.alloc(audio_need);
    PTTS_HIP(hipStreamSynchronize(stream_));
    return &b;
}

// Prefill, frame loop and chunked decode for a staged batch; audio stays in HBM (b.audio_all).
// Offline schedule of /root/reference/src/pocket_tts.cpp:548-657 applied to every utterance.
void Engine::run_staged(Batch& b, std::vector<int>& frames, std::vector<std::vector<float>>* eos_log) {
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    const Spec& s = spec();
    const int B = b.B, L = s.latent, D = s.d_model, fs = s.frame_size();
    b.reset(stream_);
    // Prefill.  The reference feeds the voice rows and then the text rows of an utterance as two causal passes
    // (:572, :575); one causal pass over [voice; text] writes the same KV cache.  All utterances go through
    // together, padded to the longest; padding rows land beyond each utterance's real length in its cache,
    // are never attended to by real rows, and are overwritten by the frames that follow.
    {
        int rpb = 1;
        std::vector<int> len(B);
        for (int i = 0; i < B; ++i) { len[i] = b.st_voice_frames[i] + b.st_ntok[i]; rpb = std::max(rpb, len[i]); }
        PTTS_HIP(hipMemcpyAsync(b.st_nv.p, b.st_voice_frames.data(), static_cast<size_t>(B) * 4, hipMemcpyHostToDevice, stream_));
        PTTS_HIP(hipMemcpyAsync(b.st_nt.p, b.st_ntok.data(), static_cast<size_t>(B) * 4, hipMemcpyHostToDevice, stream_));
        PTTS_HIP(hipMemcpyAsync(b.st_len.p, len.data(), static_cast<size_t>(B) * 4, hipMemcpyHostToDevice, stream_));
        PTTS_HIP(hipStreamSynchronize(stream_));  // len[] is a local
        b.ensure_rows(static_cast<size_t>(B) * rpb, D, s.lm_ffn);
        launch_assemble_prefill(b.voice_stage.f(), static_cast<long long>(b.st_vmax) * D, b.ids_stage.as<long long>(), opt_.max_text_tokens,
                                b.st_nv.as<int>(), b.st_nt.as<int>(), hd_.text_embed, s.vocab + 1, D, rpb, B, b.x.f(), stream_);
        lm_rows(b, 0, B, rpb, b.x.f(), b.st_len.as<int>(), len.data());
    }
    // BOS marker: NaN latent (:579)
    for (int i = 0; i < B * L; ++i) b.h_x0[i] = std::numeric_limits<float>::quiet_NaN();
    PTTS_HIP(hipMemcpyAsync(b.xcur.p, b.h_x0, static_cast<size_t>(B) * L * 4, hipMemcpyHostToDevice, stream_));

    frames.assign(B, 0);
    if (eos_log) eos_log->assign(B, std::vector<float>());
    std::vector<int> eos_step(B, -1);
    std::vector<char> done(B, 0);
    int n_done = 0, decoded = 0;
    for (int step = 0; step < opt_.max_frames; ++step) {
        lm_step_dev(b, b.xcur.f(), b.cond.f(), b.eos.f());                                  // :591
        PTTS_HIP(hipMemcpyAsync(b.h_eos, b.eos.p, static_cast<size_t>(B) * 4, hipMemcpyDeviceToHost, stream_));
        PTTS_HIP(hipStreamSynchronize(stream_));
        for (int i = 0; i < B; ++i) {
            if (done[i]) continue;
            const float e = b.h_eos[i];
            if (eos_log) (*eos_log)[i].push_back(e);
            if (e > kEosThreshold && eos_step[i] < 0) eos_step[i] = step;                   // :596
            if (eos_step[i] >= 0 && step >= eos_step[i] + opt_.frames_after_eos) { done[i] = 1; ++n_done; }  // :601
        }
        if (n_done == B) break;
        const float* x0 = b.st_has_noise ? b.noise_all.f() + static_cast<size_t>(step) * L : b.x0.f();
        const long long x0_stride = b.st_has_noise ? static_cast<long long>(opt_.max_frames) * L : L;
        flow_dev(b, b.cond.f(), x0, x0_stride, b.xcur.f(), B);                              // :606-622
        PTTS_HIP(hipMemcpy2DAsync(b.latents.f() + static_cast<size_t>(step) * L, static_cast<size_t>(opt_.max_frames) * L * 4, b.xcur.p,
                                  static_cast<size_t>(L) * 4, static_cast<size_t>(L) * 4, B, hipMemcpyDeviceToDevice, stream_));
        for (int i = 0; i < B; ++i)
            if (!done[i]) frames[i]++;
        // Every full 15-frame chunk goes to the decoder stream as soon as its latents exist: the decoder
        // never feeds back into the frame loop, so it runs beside the following frames (the reference decodes
        // after the loop, :640; the chunk boundaries and the carried state are the same).
        if ((step + 1) % kDecChunk == 0) {
            PTTS_HIP(hipEventRecord(latents_ready_, stream_));
            PTTS_HIP(hipStreamWaitEvent(dec_stream_, latents_ready_, 0));
            const int f0 = step + 1 - kDecChunk;
            mimi_decode_dev(b, b.latents.f() + static_cast<size_t>(f0) * L, static_cast<long long>(opt_.max_frames) * L, kDecChunk,
                            b.audio_all.f() + static_cast<size_t>(f0) * fs, static_cast<long long>(opt_.max_frames) * fs, dec_stream_);
            decoded = step + 1;
        }
    }
    int fmax = 0;
    for (int i = 0; i < B; ++i) fmax = std::max(fmax, frames[i]);
    // remaining frames (:454-546): chunks of <= 15 with the same carried state
    PTTS_HIP(hipEventRecord(latents_ready_, stream_));
    PTTS_HIP(hipStreamWaitEvent(dec_stream_, latents_ready_, 0));
    for (int f0 = decoded; f0 < fmax; f0 += kDecChunk) {
        const int n = std::min(kDecChunk, fmax - f0);
        mimi_decode_dev(b, b.latents.f() + static_cast<size_t>(f0) * L, static_cast<long long>(opt_.max_frames) * L, n,
                        b.audio_all.f() + static_cast<size_t>(f0) * fs, static_cast<long long>(opt_.max_frames) * fs, dec_stream_);
    }
    PTTS_HIP(hipStreamSynchronize(stream_));
    PTTS_HIP(hipStreamSynchronize(dec_stream_));
}

void Engine::fetch_staged(Batch& b, const std::vector<int>& frames, std::vector<UtteranceResult>& out, bool want_latents) {
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    const Spec& s = spec();
    const int L = s.latent, fs = s.frame_size();
    out.resize(b.B);
    for (int i = 0; i < b.B; ++i) {
        out[i].frames = frames[i];
        out[i].audio.resize(static_cast<size_t>(frames[i]) * fs);
        if (frames[i])
            PTTS_HIP(hipMemcpyAsync(out[i].audio.data(), b.audio_all.f() + static_cast<size_t>(i) * opt_.max_frames * fs,
                                    out[i].audio.size() * 4, hipMemcpyDeviceToHost, stream_));
        if (want_latents) {
            out[i].latents.resize(static_cast<size_t>(frames[i]) * L);
            if (frames[i])
                PTTS_HIP(hipMemcpyAsync(out[i].latents.data(), b.latents.f() + static_cast<size_t>(i) * opt_.max_frames * L,
                                        out[i].latents.size() * 4, hipMemcpyDeviceToHost, stream_));
        }
    }
    PTTS_HIP(hipStreamSynchronize(stream_));
}

void Engine::generate(const std::vector<Utterance>& utts, std::vector<UtteranceResult>& out, bool want_latents) {
    out.assign(utts.size(), UtteranceResult());
    if (utts.empty()) return;
    std::lock_guard<std::recursive_mutex> call(call_mu_);  // stage/run/fetch of one call must not interleave with another call's
    Batch* b = stage(utts);
    std::vector<int> frames;
    std::vector<std::vector<float>> eos;
    run_staged(*b, frames, want_latents ? &eos : nullptr);
    fetch_staged(*b, frames, out, want_latents);
    if (want_latents)
        for (size_t i = 0; i < out.size(); ++i) out[i].eos = std::move(eos[i]);
}

int Engine::generate_streaming(const Utterance& utt, int chunk_frames, const ChunkCallback& cb,
                               const std::atomic<bool>* cancel_flag, bool poll_cancel, const std::function<void(int)>& progress,
                               UtteranceResult* capture) {
    if (!cb) throw std::invalid_argument("Callback must be provided");  // :693-695
    if (chunk_frames <= 0) chunk_frames = 1;  // size() >= 0 holds after every frame (:790)
    std::lock_guard<std::recursive_mutex> call(call_mu_);
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    const Spec& s = spec();
    const int L = s.latent, fs = s.frame_size();
    Batch& b = *acquire_batch(1);
    prefill_utterance(b, 0, utt);
    std::vector<Utterance> one{utt};
    std::vector<std::mt19937> rngs;
    rngs.emplace_back(rng_());
    const bool any_noise = opt_.temperature > 0 || utt.noise;
    for (int i = 0; i < L; ++i) b.h_x0[i] = std::numeric_limits<float>::quiet_NaN();
    PTTS_HIP(hipMemcpyAsync(b.xcur.p, b.h_x0, static_cast<size_t>(L) * 4, hipMemcpyHostToDevice, stream_));
    PTTS_HIP(hipStreamSynchronize(stream_));

    std::vector<float> chunk_audio;
    auto flush = [&](int first, int count, bool is_final) {
        chunk_audio.resize(static_cast<size_t>(count) * fs);
        for (int f0 = 0; f0 < count; f0 += kDecChunk) {  // sub-chunks of <= 15 frames (:796)
            const int n = std::min(kDecChunk, count - f0);
            mimi_decode_dev(b, b.latents.f() + static_cast<size_t>(first + f0) * L, static_cast<long long>(opt_.max_frames) * L, n,
                            b.audio.f(), static_cast<long long>(kDecChunk) * fs, stream_);
            PTTS_HIP(hipMemcpyAsync(chunk_audio.data() + static_cast<size_t>(f0) * fs, b.audio.p, static_cast<size_t>(n) * fs * 4,
                                    hipMemcpyDeviceToHost, stream_));
            PTTS_HIP(hipStreamSynchronize(stream_));
        }
        if (capture) capture->audio.insert(capture->audio.end(), chunk_audio.begin(), chunk_audio.end());
        cb(chunk_audio.data(), static_cast<int>(chunk_audio.size()), is_final);
    };

    int eos_step = -1, frames = 0, pending = 0, total = 0;
    for (int step = 0; step < opt_.max_frames; ++step) {
        if (poll_cancel && cancel_flag && cancel_flag->load()) break;                      // :745-750
        lm_step_dev(b, b.xcur.f(), b.cond.f(), b.eos.f());                                 // :753
        PTTS_HIP(hipMemcpyAsync(b.h_eos, b.eos.p, 4, hipMemcpyDeviceToHost, stream_));
        PTTS_HIP(hipStreamSynchronize(stream_));
        const float e = b.h_eos[0];
        if (capture) capture->eos.push_back(e);
        if (e > kEosThreshold && eos_step < 0) eos_step = step;                            // :758
        if (eos_step >= 0 && step >= eos_step + opt_.frames_after_eos) break;              // :763
        if (any_noise) {
            fill_noise(rngs, one, step, b.h_x0);
            PTTS_HIP(hipMemcpyAsync(b.x0.p, b.h_x0, static_cast<size_t>(L) * 4, hipMemcpyHostToDevice, stream_));
        }
        flow_dev(b, b.cond.f(), b.x0.f(), L, b.xcur.f(), 1);                               // :777-784
        PTTS_HIP(hipMemcpyAsync(b.latents.f() + static_cast<size_t>(step) * L, b.xcur.p, static_cast<size_t>(L) * 4, hipMemcpyDeviceToDevice, stream_));
        if (any_noise) PTTS_HIP(hipStreamSynchronize(stream_));
        ++frames; ++pending;
        if (pending >= chunk_frames) {                                                     // :790 (the second clause cannot hold here)
            flush(frames - pending, pending, false);
            total += pending * fs;
            pending = 0;
        }
        if (progress) progress(step + 1);                                                  // :891-893
    }
    // Tail flush, the only place is_final = 1 is delivered; skipped after a cancel (:897-965).
    const bool cancel_now = cancel_flag && cancel_flag->load();
    if (pending > 0 && !cancel_now) {
        flush(frames - pending, pending, true);
        total += pending * fs;
        pending = 0;
    }
    PTTS_HIP(hipStreamSynchronize(stream_));
    if (capture) {
        capture->frames = frames - pending;
        capture->latents.resize(static_cast<size_t>(frames) * L);
        if (frames) PTTS_HIP(hipMemcpy(capture->latents.data(), b.latents.p, capture->latents.size() * 4, hipMemcpyDeviceToHost));
    }
    return total;
}

// ------------------------------------------------------------------------------------------ voice encoder

std::vector<float> Engine::encode_voice(const float* audio, int S) {
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    const Spec& s = spec();
    if (S <= 0) throw std::runtime_error("voice clip is empty");
    const int M = s.mimi_dim, U = s.upsample, D = s.d_model, rk = s.res_kernel, lk = s.last_kernel;
    auto zeros = [&](size_t floats) {
        DevBuf d(floats * 4);
        PTTS_HIP(hipMemsetAsync(d.p, 0, d.bytes, stream_));
        return d;
    };
    DevBuf dau(static_cast<size_t>(S) * 4);
    PTTS_HIP(hipMemcpyAsync(dau.p, audio, static_cast<size_t>(S) * 4, hipMemcpyHostToDevice, stream_));
    int c = s.n_filters, T = S;
    DevBuf xraw(static_cast<size_t>(T) * c * 4);
    DevBuf xe = zeros(static_cast<size_t>(rk - 1 + T) * c);
    launch_conv_from_mono(dau.f(), 0, nullptr, T, s.kernel, c, hd_.enc_conv0_w, hd_.enc_conv0_b, xraw.f(), 0,
                          xe.f() + static_cast<size_t>(rk - 1) * c, 0, 1, stream_);
    for (int i = 0; i < 3; ++i) {
        const int r = s.ratios[2 - i], To = (T + r - 1) / r, Tpad = To * r;
        DevBuf h1(static_cast<size_t>(T) * (c / 2) * 4);
        GemmArgs c1;
        c1.A = xe.f(); c1.lda = c; c1.W = hd_.enc_c1[i].p; c1.wtype = hd_.enc_c1[i].wtype; c1.bias = hd_.enc_c1_b[i];
        c1.M = T; c1.N = c / 2; c1.K = rk * c; c1.C = h1.f(); c1.ldc = c / 2; c1.c_act = ACT_ELU;
        launch_gemm(c1, stream_);
        DevBuf ye = zeros(static_cast<size_t>(r + Tpad) * c);
        GemmArgs c2;
        c2.A = h1.f(); c2.lda = c / 2; c2.W = hd_.enc_c2[i].p; c2.wtype = hd_.enc_c2[i].wtype; c2.bias = hd_.enc_c2_b[i];
        c2.M = T; c2.N = c; c2.K = c / 2; c2.R = xraw.f(); c2.ldr = c; c2.C = ye.f() + static_cast<size_t>(r) * c; c2.ldc = c; c2.c_act = ACT_ELU;
        launch_gemm(c2, stream_);
        const int pn = i < 2 ? rk - 1 : lk - 1;
        DevBuf xraw2(static_cast<size_t>(To) * 2 * c * 4);
        DevBuf xe2 = zeros(static_cast<size_t>(pn + To) * 2 * c);
        GemmArgs dn;
        dn.A = ye.f(); dn.lda = r * c; dn.W = hd_.enc_down[i].p; dn.wtype = hd_.enc_down[i].wtype; dn.bias = hd_.enc_down_b[i];
        dn.M = To; dn.N = 2 * c; dn.K = 2 * r * c; dn.C = xraw2.f(); dn.ldc = 2 * c;
        dn.C2 = xe2.f() + static_cast<size_t>(pn) * 2 * c; dn.ldc2 = 2 * c; dn.c2_act = ACT_ELU;
        launch_gemm(dn, stream_);
        PTTS_HIP(hipStreamSynchronize(stream_));  // buffers of this stage are released below
        xraw = std::move(xraw2);
        xe = std::move(xe2);
        T = To; c *= 2;
    }
    DevBuf y(static_cast<size_t>(T) * M * 4);
    GemmArgs fin;
    fin.A = xe.f(); fin.lda = c; fin.W = hd_.enc_final.p; fin.wtype = hd_.enc_final.wtype; fin.bias = hd_.enc_final_b;
    fin.M = T; fin.N = M; fin.K = lk * c; fin.C = y.f(); fin.ldc = M;
    launch_gemm(fin, stream_);
    // transformer over the whole clip from position 0
    if (T > 16000) throw std::runtime_error("voice clip too long for the encoder attention buffer");
    const int Nv = (T + U - 1) / U, Tpad = Nv * U;
    DevBuf kc(static_cast<size_t>(s.mimi_layers) * T * M * 4), vc(static_cast<size_t>(s.mimi_layers) * T * M * 4);
    DevBuf pos = zeros(1);
    DevBuf sh(static_cast<size_t>(T) * M * 4), sqkv(static_cast<size_t>(T) * 3 * M * 4), sq(static_cast<size_t>(T) * M * 4),
        satt(static_cast<size_t>(T) * M * 4), sff(static_cast<size_t>(T) * s.mimi_ffn * 4);
    TfScratch sc{sh.f(), sqkv.f(), sq.f(), satt.f(), sff.f()};
    DevBuf yd = zeros(static_cast<size_t>(U + Tpad) * M);
    if (T > model_->mimi_tab_len) throw std::runtime_error("voice clip exceeds the rotary table");
    transformer(hd_.enc, kc.f(), vc.f(), static_cast<long long>(T) * M, T, pos.as<int>(), 0, 1, T, y.f(), M, s.mimi_heads, s.mimi_ffn,
                s.mimi_eps, s.mimi_context, model_->mimi_cos, model_->mimi_sin, model_->mimi_tab_len, sc,
                yd.f() + static_cast<size_t>(U) * M, 0, M, stream_);
    DevBuf d(static_cast<size_t>(Nv) * M * 4), o(static_cast<size_t>(Nv) * D * 4);
    GemmArgs ds;
    ds.A = yd.f(); ds.lda = U * M; ds.W = hd_.enc_ds.p; ds.wtype = hd_.enc_ds.wtype; ds.M = Nv; ds.N = M; ds.K = 2 * U * M; ds.C = d.f(); ds.ldc = M;
    launch_gemm(ds, stream_);
    GemmArgs sp;
    sp.A = d.f(); sp.lda = M; sp.W = hd_.enc_spk.p; sp.wtype = hd_.enc_spk.wtype; sp.M = Nv; sp.N = D; sp.K = M; sp.C = o.f(); sp.ldc = D;
    launch_gemm(sp, stream_);
    std::vector<float> out(static_cast<size_t>(Nv) * D);
    PTTS_HIP(hipMemcpyAsync(out.data(), o.p, out.size() * 4, hipMemcpyDeviceToHost, stream_));
    PTTS_HIP(hipStreamSynchronize(stream_));
    return out;
}

void Engine::encode_voice_batch(const std::vector<const float*>& clips, const std::vector<int>& lens, std::vector<std::vector<float>>& out) {
    out.clear();
    for (size_t i = 0; i < clips.size(); ++i) out.push_back(encode_voice(clips[i], lens[i]));
}

// ------------------------------------------------------------------------------------------ graph-level entry points

void Engine::op_lm_prefill(Batch* b, int slot, const float* emb, int rows) {
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    if (slot < 0 || slot >= b->B) throw std::runtime_error("lm_prefill: slot out of range");
    const int D = spec().d_model, chunk = 256;
    for (int r0 = 0; r0 < rows; r0 += chunk) {
        int n = std::min(chunk, rows - r0);
        PTTS_HIP(hipMemcpyAsync(b->x.p, emb + static_cast<size_t>(r0) * D, static_cast<size_t>(n) * D * 4, hipMemcpyHostToDevice, stream_));
        lm_rows(*b, slot, 1, n, b->x.f());
    }
    PTTS_HIP(hipStreamSynchronize(stream_));
}

void Engine::op_lm_step(Batch* b, const float* seq, float* cond, float* eos) {
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    const Spec& s = spec();
    PTTS_HIP(hipMemcpyAsync(b->xcur.p, seq, static_cast<size_t>(b->B) * s.latent * 4, hipMemcpyHostToDevice, stream_));
    lm_step_dev(*b, b->xcur.f(), b->cond.f(), b->eos.f());
    PTTS_HIP(hipMemcpyAsync(cond, b->cond.p, static_cast<size_t>(b->B) * s.d_model * 4, hipMemcpyDeviceToHost, stream_));
    PTTS_HIP(hipMemcpyAsync(eos, b->eos.p, static_cast<size_t>(b->B) * 4, hipMemcpyDeviceToHost, stream_));
    PTTS_HIP(hipStreamSynchronize(stream_));
}

void Engine::op_flow_euler(const float* cond, const float* x0, int B, float* x_out) {
    std::lock_guard<std::recursive_mutex> call(call_mu_);
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    const Spec& s = spec();
    Batch& b = *acquire_batch(B);
    PTTS_HIP(hipMemcpyAsync(b.cond.p, cond, static_cast<size_t>(B) * s.d_model * 4, hipMemcpyHostToDevice, stream_));
    PTTS_HIP(hipMemcpyAsync(b.x0.p, x0, static_cast<size_t>(B) * s.latent * 4, hipMemcpyHostToDevice, stream_));
    flow_dev(b, b.cond.f(), b.x0.f(), s.latent, b.xcur.f(), B);
    PTTS_HIP(hipMemcpyAsync(x_out, b.xcur.p, static_cast<size_t>(B) * s.latent * 4, hipMemcpyDeviceToHost, stream_));
    PTTS_HIP(hipStreamSynchronize(stream_));
}

void Engine::op_mimi_decode(Batch* b, const float* latents, int n, float* audio) {
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    const Spec& s = spec();
    const int L = s.latent, fs = s.frame_size();
    if (n <= 0 || n > kDecChunk) throw std::runtime_error("decoder chunk must hold 1..15 frames");
    if (n > opt_.max_frames) throw std::runtime_error("decoder chunk larger than max_frames");
    // stage the chunk at the front of each utterance's latent buffer
    PTTS_HIP(hipMemcpy2DAsync(b->latents.p, static_cast<size_t>(opt_.max_frames) * L * 4, latents, static_cast<size_t>(n) * L * 4,
                              static_cast<size_t>(n) * L * 4, b->B, hipMemcpyHostToDevice, stream_));
    mimi_decode_dev(*b, b->latents.f(), static_cast<long long>(opt_.max_frames) * L, n, b->audio.f(), static_cast<long long>(kDecChunk) * fs, stream_);
    PTTS_HIP(hipMemcpy2DAsync(audio, static_cast<size_t>(n) * fs * 4, b->audio.p, static_cast<size_t>(kDecChunk) * fs * 4,
                              static_cast<size_t>(n) * fs * 4, b->B, hipMemcpyDeviceToHost, stream_));
    PTTS_HIP(hipStreamSynchronize(stream_));
}

}  // namespace ptts
