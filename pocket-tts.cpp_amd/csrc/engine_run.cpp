// Orchestration on top of the forward passes: prefill, the autoregressive frame loop, offline and
// streaming decode schedules, the voice encoder and the graph-level entry points.
// The loop bodies restate /root/reference/src/pocket_tts.cpp:548-657 (offline) and :686-979 (streaming).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>

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
    const int T_all = static_cast<int>(u.token_ids.size());
    if (u.voice_frames + T_all <= chunk && !std::getenv("POCKET_TTS_TWO_PASS_PREFILL")) {
        // One causal pass over [voice; text] writes the same KV cache as the two passes (the batched loop does the
        // same) and halves the number of launches before the first frame.
        if (u.voice_frames)
            PTTS_HIP(hipMemcpyAsync(b.x.p, u.voice, static_cast<size_t>(u.voice_frames) * D * 4, hipMemcpyHostToDevice, stream_));
        PTTS_HIP(hipMemcpyAsync(b.ids.p, u.token_ids.data(), static_cast<size_t>(T_all) * 8, hipMemcpyHostToDevice, stream_));
        launch_embed_rows(b.ids.as<long long>(), T_all, hd_.text_embed, s.vocab + 1, D, b.x.f() + static_cast<size_t>(u.voice_frames) * D, stream_);
        lm_rows(b, slot, 1, u.voice_frames + T_all, b.x.f());
        return;
    }
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
// touches host memory only for the per-step EOS logits.  The batch is cut into up to `lanes` contiguous parts
// of at least 16 utterances (one MFMA row group) each.
Job* Engine::stage(const std::vector<Utterance>& utts) {
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    const int B = static_cast<int>(utts.size());
    if (B == 0) throw std::runtime_error("empty batch");
    for (auto& u : utts) {
        if (u.token_ids.empty()) throw std::runtime_error("Text cannot be empty");
        if (static_cast<int>(u.token_ids.size()) > opt_.max_text_tokens) throw std::runtime_error("text has more tokens than the engine was sized for");
    }
    int parts = std::min(lanes_, std::max(1, B / 16));
    const int per = ((B + parts - 1) / parts + 15) / 16 * 16;   // whole row groups per part
    parts = (B + per - 1) / per;
    job_.parts.clear(); job_.first.clear(); job_.B = B;
    for (int p = 0; p < parts; ++p) {
        const int first = p * per, n = std::min(per, B - first);
        Batch* b = acquire_batch(n, p);
        if (parts > 1 && !lane_shared_decoder_) b->dec_st = b->st;   // decoder chunks in line: other lanes fill the gap
        stage_part(*b, utts.data() + first, n);
        job_.parts.push_back(b);
        job_.first.push_back(first);
    }
    for (Batch* b : job_.parts) PTTS_HIP(hipStreamSynchronize(b->st));
    return &job_;
}

void Engine::stage_part(Batch& b, const Utterance* utts, int B) {
    const Spec& s = spec();
    const int L = s.latent, D = s.d_model;
    int vmax = 1;
    for (int i = 0; i < B; ++i) vmax = std::max(vmax, utts[i].voice_frames);
    b.st_vmax = vmax;
    if (b.voice_stage.bytes < static_cast<size_t>(B) * vmax * D * 4) b.voice_stage.alloc(static_cast<size_t>(B) * vmax * D * 4);
    if (b.ids_stage.bytes < static_cast<size_t>(B) * opt_.max_text_tokens * 8) b.ids_stage.alloc(static_cast<size_t>(B) * opt_.max_text_tokens * 8);
    b.st_voice_frames.assign(B, 0);
    b.st_ntok.assign(B, 0);
    bool any_noise = opt_.temperature > 0;
    for (int i = 0; i < B; ++i) any_noise = any_noise || utts[i].noise;
    b.st_has_noise = any_noise;
    for (int i = 0; i < B; ++i) {
        const Utterance& u = utts[i];
        b.st_voice_frames[i] = u.voice_frames;
        b.st_ntok[i] = static_cast<int>(u.token_ids.size());
        if (u.voice_frames)
            PTTS_HIP(hipMemcpyAsync(b.voice_stage.f() + static_cast<size_t>(i) * vmax * D, u.voice, static_cast<size_t>(u.voice_frames) * D * 4,
                                    hipMemcpyHostToDevice, b.st));
        PTTS_HIP(hipMemcpyAsync(b.ids_stage.as<long long>() + static_cast<size_t>(i) * opt_.max_text_tokens, u.token_ids.data(),
                                u.token_ids.size() * 8, hipMemcpyHostToDevice, b.st));
    }
    if (any_noise) {
        // Start points for every frame: explicit ones, else N(0, sqrt(T)) from a fresh distribution object per
        // frame and utterance (src/pocket_tts.cpp:606-613), drawn up front so the frame loop stays on the device.
        const size_t per = static_cast<size_t>(opt_.max_frames) * L;
        if (b.noise_all.bytes < static_cast<size_t>(B) * per * 4) b.noise_all.alloc(static_cast<size_t>(B) * per * 4);
        std::vector<float> host(static_cast<size_t>(B) * per, 0.f);
        std::vector<uint32_t> seeds(static_cast<size_t>(B), 0u);
        {
            std::lock_guard<std::mutex> lk(rng_mu_);
            for (int i = 0; i < B; ++i)
                if (!utts[i].noise && opt_.temperature > 0) seeds[i] = static_cast<uint32_t>(rng_());   // in utterance order
        }
        auto fill = [&](int i) {
            float* dst = host.data() + static_cast<size_t>(i) * per;
            if (utts[i].noise) {
                std::memcpy(dst, utts[i].noise, per * 4);
            } else if (opt_.temperature > 0) {
                std::mt19937 gen(seeds[i]);
                for (int f = 0; f < opt_.max_frames; ++f) {
                    std::normal_distribution<float> dist(0.0f, std::sqrt(opt_.temperature));
                    for (int k = 0; k < L; ++k) dst[static_cast<size_t>(f) * L + k] = dist(gen);
                }
            }
        };
        const int workers = std::min({B / 4, 8, static_cast<int>(std::max(1u, std::thread::hardware_concurrency()))});
        if (workers > 1) {   // 0.1 ms of one core per utterance at 125 frames
            std::vector<std::thread> pool;
            for (int w = 0; w < workers; ++w)
                pool.emplace_back([&, w] { for (int i = w; i < B; i += workers) fill(i); });
            for (auto& t : pool) t.join();
        } else {
            for (int i = 0; i < B; ++i) fill(i);
        }
        PTTS_HIP(hipMemcpyAsync(b.noise_all.p, host.data(), host.size() * 4, hipMemcpyHostToDevice, b.st));
        PTTS_HIP(hipStreamSynchronize(b.st));
    }
    const size_t audio_need = static_cast<size_t>(B) * opt_.max_frames * s.frame_size() * 4;
    if (b.audio_all.bytes < audio_need) b.audio_all.alloc(audio_need);
    PTTS_HIP(hipStreamSynchronize(b.st));
}

// One frame, free of host-side frame state.  With timing on, the events of this parity are recorded around the
// phases (as graph nodes when the body is being captured).
void Engine::frame_body(Batch& b, int parity) {
    hipStream_t st = b.st;
    hipEvent_t* ev = timing_ ? b.tev[parity] : nullptr;
    if (ev) PTTS_HIP(hipEventRecord(ev[0], st));
    lm_step_dev(b, b.xcur.f(), b.cond.f(), b.eos.f(), true);                                // :591
    if (ev) { PTTS_HIP(hipEventRecord(ev[1], st)); PTTS_HIP(hipEventRecord(ev[2], st)); }
    // flow phase (:606-622); its second launch also publishes the EOS logits and advances the frame counter, its last
    // launch files the new latent under the frame index
    flow_dev(b, b.cond.f(), b.x0_src, b.x0_stride, b.xcur.f(), b.Bw, b.step_dev.as<int>(), b.x0_step_stride, ev ? ev[3] : nullptr,
             ev ? ev[4] : nullptr, true);
    if (ev) {
        PTTS_HIP(hipEventRecord(ev[5], st));
        PTTS_HIP(hipEventRecord(ev[6], st));   // nothing between these two: what a bracketing pair costs by itself
        PTTS_HIP(hipEventRecord(ev[7], st));
    }
}

// The frame graphs of a batch at its current launch width, (re)captured when the start-point source, the ring mode or the
// timing mode changed.  Returns the two executables (one per frame parity).
hipGraphExec_t* Engine::ensure_frame_graph(Batch& b) {
    Batch::FrameGraphs& fg = b.frame_graphs[b.Bw];
    if (fg.exec[0] && fg.x0 == b.x0_src && fg.x0_stride == b.x0_stride && fg.timed == timing_ && fg.ring == b.step_ring) return fg.exec;
    for (auto& ex : fg.exec)
        if (ex) { PTTS_HIP(hipGraphExecDestroy(ex)); ex = nullptr; }
    for (int parity = 0; parity < 2; ++parity) {
        hipGraph_t g = nullptr;
        PTTS_HIP(hipStreamBeginCapture(b.st, hipStreamCaptureModeThreadLocal));
        try {
            frame_body(b, parity);
        } catch (...) {
            (void)hipStreamEndCapture(b.st, &g);
            if (g) (void)hipGraphDestroy(g);
            throw;
        }
        PTTS_HIP(hipStreamEndCapture(b.st, &g));
        PTTS_HIP(hipGraphInstantiate(&fg.exec[parity], g, nullptr, nullptr, 0));
        PTTS_HIP(hipGraphDestroy(g));
    }
    fg.x0 = b.x0_src; fg.x0_stride = b.x0_stride; fg.timed = timing_; fg.ring = b.step_ring;
    return fg.exec;
}

// Prefill, frame loop and chunked decode for a staged batch; audio stays in HBM (audio_all of every part).
// Offline schedule of /root/reference/src/pocket_tts.cpp:548-657 applied to every utterance.  All parts are
// driven from this one thread: a frame is one graph launch, the EOS logits of step k reach the host while the
// flow head of step k is still running, and step k+1 is enqueued behind it at once, so neither the host nor
// the other parts ever wait for a whole frame.
void Engine::run_staged(Job& job, std::vector<int>& frames_out, std::vector<std::vector<float>>* eos_log) {
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    const Spec& s = spec();
    const int L = s.latent, D = s.d_model, fs = s.frame_size();
    frames_out.assign(job.B, 0);
    if (eos_log) eos_log->assign(job.B, std::vector<float>());

    struct PartRun {
        Batch* b; int first; std::vector<int> len, eos_step; std::vector<char> done;
        int n_done = 0, decoded = 0, step = 0, executed = 0, timed = 0; bool launched = false, finished = false;
        hipGraphExec_t* exec = nullptr;
    };
    std::vector<PartRun> runs(job.parts.size());
    for (size_t p = 0; p < runs.size(); ++p) {
        PartRun& r = runs[p];
        Batch& b = *job.parts[p];
        r.b = &b; r.first = job.first[p];
        const int B = b.B;
        hipStream_t st = b.st;
        b.reset(st);
        b.Bw = B; b.step_ring = 0; b.eos_ring = 0;
        PTTS_HIP(hipMemsetAsync(b.step_dev.p, 0, 4, st));
        *b.h_flag = 0;
        // Prefill.  The reference feeds the voice rows and then the text rows of an utterance as two causal passes
        // (:572, :575); one causal pass over [voice; text] writes the same KV cache.  All utterances of a part go
        // through together, padded to the longest; padding rows land beyond each utterance's real length in its
        // cache, are never attended to by real rows, and are overwritten by the frames that follow.
        int rpb = 1;
        r.len.resize(B);
        for (int i = 0; i < B; ++i) { r.len[i] = b.st_voice_frames[i] + b.st_ntok[i]; rpb = std::max(rpb, r.len[i]); }
        PTTS_HIP(hipMemcpyAsync(b.st_nv.p, b.st_voice_frames.data(), static_cast<size_t>(B) * 4, hipMemcpyHostToDevice, st));
        PTTS_HIP(hipMemcpyAsync(b.st_nt.p, b.st_ntok.data(), static_cast<size_t>(B) * 4, hipMemcpyHostToDevice, st));
        PTTS_HIP(hipMemcpyAsync(b.st_len.p, r.len.data(), static_cast<size_t>(B) * 4, hipMemcpyHostToDevice, st));
        b.ensure_rows(static_cast<size_t>(B) * rpb, D, s.lm_ffn);
        launch_assemble_prefill(b.voice_stage.f(), static_cast<long long>(b.st_vmax) * D, b.ids_stage.as<long long>(), opt_.max_text_tokens,
                                b.st_nv.as<int>(), b.st_nt.as<int>(), hd_.text_embed, s.vocab + 1, D, rpb, B, b.x.f(), st);
        lm_rows(b, 0, B, rpb, b.x.f(), b.st_len.as<int>(), r.len.data());
        // BOS marker: NaN latent (:579), replaced here by the BOS embedding it stands for so that the frame body
        // needs no preparation launch
        for (int i = 0; i < B * L; ++i) b.h_x0[i] = std::numeric_limits<float>::quiet_NaN();
        PTTS_HIP(hipMemcpyAsync(b.seq_in.p, b.h_x0, static_cast<size_t>(B) * L * 4, hipMemcpyHostToDevice, st));
        launch_prep_latent(b.seq_in.f(), b.xcur.f(), B, L, 0, hd_.bos, nullptr, st);
        b.x0_src = b.st_has_noise ? b.noise_all.f() : b.x0.f();
        b.x0_stride = b.st_has_noise ? static_cast<long long>(opt_.max_frames) * L : L;
        b.x0_step_stride = b.st_has_noise ? L : 0;
        r.eos_step.assign(B, -1);
        r.done.assign(B, 0);
    }
    // Events recorded by graph nodes cannot be read back with hipEventElapsedTime on this runtime, so a timed run
    // launches the same frame body eagerly.
    const bool graph = use_graph_ && !timing_;
    if (graph)
        for (auto& r : runs) {
            const std::vector<int> keep = r.b->lm_pos_host;
            r.exec = ensure_frame_graph(*r.b);
            r.b->lm_pos_host = keep;
        }

    auto decode_chunk = [&](PartRun& r, int f0, int n) {
        Batch& b = *r.b;
        if (b.dec_st != b.st) {
            PTTS_HIP(hipEventRecord(b.ready, b.st));
            PTTS_HIP(hipStreamWaitEvent(b.dec_st, b.ready, 0));
        }
        mimi_decode_dev(b, b.latents.f() + static_cast<size_t>(f0) * L, static_cast<long long>(b.lat_frames) * L, n,
                        b.audio_all.f() + static_cast<size_t>(f0) * fs, static_cast<long long>(opt_.max_frames) * fs, b.dec_st);
    };
    auto collect_timing = [&](Batch& b, int parity) {
        float lm = 0.f, fl = 0.f, hd = 0.f;
        if (hipEventElapsedTime(&lm, b.tev[parity][0], b.tev[parity][1]) != hipSuccess) return;
        if (hipEventElapsedTime(&fl, b.tev[parity][2], b.tev[parity][5]) != hipSuccess) return;
        if (hipEventElapsedTime(&hd, b.tev[parity][3], b.tev[parity][4]) != hipSuccess) return;
        float empty = 0.f;
        if (hipEventElapsedTime(&empty, b.tev[parity][6], b.tev[parity][7]) != hipSuccess) return;
        hd = std::max(hd - empty, 0.f);   // take the bracketing pair's own cost out
        std::lock_guard<std::mutex> g(ev_mu_);
        times_.lm_ms += lm; times_.lm_calls++;
        times_.flow_ms += fl; times_.flow_calls++;
        FlowHeadArgs fa;
        fa.C = spec().flow_dim; fa.L = spec().latent; fa.B = b.Bw; fa.steps = opt_.lsd_steps; fa.blocks = spec().flow_blocks;
        fa.wtype = model_->flow_wtype; fa.block_wtype = model_->flow_block_wtype; fa.ring_split = model_->flow_ring_split;
        fa.ring_ctl = owns_ring_ ? b.flow_ctl.as<unsigned>() : nullptr; fa.ring_wg_budget = 224 / std::max(lanes_, 1);
        times_.head_ms += hd; times_.head_calls += flow_head_launch_count(fa);
    };

    // frames per decoder call of the batch (at most 15, src/pocket_tts.cpp:461; the decoder is streaming, so any chunking gives
    // the same samples)
    static const int dec_every = [] { const char* e = std::getenv("POCKET_TTS_DEC_CHUNK"); const int v = e ? std::atoi(e) : kDecChunk; return v >= 1 && v <= kDecChunk ? v : kDecChunk; }();
    size_t n_finished = 0;
    FlagWait flag_wait;
    auto last_progress = std::chrono::steady_clock::now();   // several parts are polled in turn: none may stall for longer than the wait limit
    static const double stall_ms = [] { const char* e = std::getenv("POCKET_TTS_WAIT_LIMIT_MS"); return static_cast<double>(e ? std::max(1, std::atoi(e)) : 2000); }();
    while (n_finished < runs.size()) {
        if (runs.size() > 1 && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - last_progress).count() > stall_ms)
            throw std::runtime_error("no part of the staged batch published a frame within " + std::to_string(static_cast<long>(stall_ms)) + " ms");
        for (auto& r : runs) {
            if (r.finished || r.launched) continue;
            Batch& b = *r.b;
            lm_bookkeeping(b);
            if (graph) PTTS_HIP(hipGraphLaunch(r.exec[r.step & 1], b.st));
            else frame_body(b, r.step & 1);
            r.launched = true;
        }
        if (runs.size() == 1 && runs[0].launched && !runs[0].finished) wait_flag(runs[0].b->h_flag, runs[0].step + 1, runs[0].b->st, flag_wait);
        for (auto& r : runs) {
            if (r.finished || !r.launched) continue;
            Batch& b = *r.b;
            if (__atomic_load_n(b.h_flag, __ATOMIC_ACQUIRE) < r.step + 1) { __builtin_ia32_pause(); continue; }   // EOS logits of this step not here yet
            const int B = b.B, step = r.step;
            last_progress = std::chrono::steady_clock::now();
            r.executed = step + 1;
            if (timing_ && step > 0) { collect_timing(b, (step - 1) & 1); r.timed = step; }   // frame step-1 has fully retired
            const float* e_now = b.h_eos_all + static_cast<size_t>(step) * B;
            for (int i = 0; i < B; ++i) {
                if (r.done[i]) continue;
                const float e = e_now[i];
                if (eos_log) (*eos_log)[r.first + i].push_back(e);
                if (e > kEosThreshold && r.eos_step[i] < 0) r.eos_step[i] = step;                     // :596
                if (r.eos_step[i] >= 0 && step >= r.eos_step[i] + opt_.frames_after_eos) { r.done[i] = 1; ++r.n_done; }  // :601
            }
            r.launched = false;
            bool stop = r.n_done == B;   // the flow head of this step is already queued; its output is not used
            if (!stop) {
                for (int i = 0; i < B; ++i)
                    if (!r.done[i]) frames_out[r.first + i]++;
                // Every full 15-frame chunk goes to the decoder as soon as its latents are queued: the decoder never
                // feeds back into the frame loop (the reference decodes after the loop, :640; the chunk boundaries
                // and the carried state are the same).
                if ((step + 1) % dec_every == 0) {
                    decode_chunk(r, step + 1 - dec_every, dec_every);
                    r.decoded = step + 1;
                }
                r.step = step + 1;
                stop = r.step >= opt_.max_frames;
            }
            if (stop) {
                int fmax = 0;
                for (int i = 0; i < B; ++i) fmax = std::max(fmax, frames_out[r.first + i]);
                // The frames left over (:454-546) are decoded on the frame-loop stream, which has the whole chip now
                // that the loop has ended, behind whatever the (CU-masked) decoder stream still has queued.
                if (b.dec_st != b.st && r.decoded < fmax) {
                    PTTS_HIP(hipEventRecord(dec_done_, b.dec_st));
                    PTTS_HIP(hipStreamWaitEvent(b.st, dec_done_, 0));
                }
                for (int f0 = r.decoded; f0 < fmax; f0 += kDecChunk) {
                    const int n = std::min(kDecChunk, fmax - f0);
                    mimi_decode_dev(b, b.latents.f() + static_cast<size_t>(f0) * L, static_cast<long long>(b.lat_frames) * L, n,
                                    b.audio_all.f() + static_cast<size_t>(f0) * fs, static_cast<long long>(opt_.max_frames) * fs, b.st);
                }
                r.finished = true;
                ++n_finished;
            }
        }
    }
    for (auto& r : runs) {
        PTTS_HIP(hipStreamSynchronize(r.b->st));
        if (timing_ && r.timed < r.executed) collect_timing(*r.b, (r.executed - 1) & 1);
    }
    bool dec_separate = false;
    for (auto& r : runs) dec_separate = dec_separate || r.b->dec_st != r.b->st;
    if (dec_separate) PTTS_HIP(hipStreamSynchronize(dec_stream_));
    for (auto& r : runs) check_flow_ring(r.b->flow_ctl.as<unsigned>(), r.b->B);
}

void Engine::fetch_staged(Job& job, const std::vector<int>& frames, std::vector<UtteranceResult>& out, bool want_latents) {
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    const Spec& s = spec();
    const int L = s.latent, fs = s.frame_size();
    out.resize(job.B);
    for (size_t p = 0; p < job.parts.size(); ++p) {
        Batch& b = *job.parts[p];
        for (int j = 0; j < b.B; ++j) {
            const int i = job.first[p] + j;
            out[i].frames = frames[i];
            out[i].audio.resize(static_cast<size_t>(frames[i]) * fs);
            if (frames[i])
                PTTS_HIP(hipMemcpyAsync(out[i].audio.data(), b.audio_all.f() + static_cast<size_t>(j) * opt_.max_frames * fs,
                                        out[i].audio.size() * 4, hipMemcpyDeviceToHost, b.st));
            if (want_latents) {
                out[i].latents.resize(static_cast<size_t>(frames[i]) * L);
                if (frames[i])
                    PTTS_HIP(hipMemcpyAsync(out[i].latents.data(), b.latents.f() + static_cast<size_t>(j) * b.lat_frames * L,
                                            out[i].latents.size() * 4, hipMemcpyDeviceToHost, b.st));
            }
        }
    }
    for (Batch* b : job.parts) PTTS_HIP(hipStreamSynchronize(b->st));
}

void Engine::generate(const std::vector<Utterance>& utts, std::vector<UtteranceResult>& out, bool want_latents) {
    out.assign(utts.size(), UtteranceResult());
    if (utts.empty()) return;
    std::lock_guard<std::recursive_mutex> call(call_mu_);  // stage/run/fetch of one call must not interleave with another call's
    Job* job = stage(utts);
    std::vector<int> frames;
    std::vector<std::vector<float>> eos;
    run_staged(*job, frames, want_latents ? &eos : nullptr);
    fetch_staged(*job, frames, out, want_latents);
    if (want_latents)
        for (size_t i = 0; i < out.size(); ++i) out[i].eos = std::move(eos[i]);
}

int Engine::generate_streaming_inline(const Utterance& utt, int chunk_frames, const ChunkCallback& cb,
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
    b.Bw = 1; b.step_ring = 0; b.eos_ring = 0;
    prefill_utterance(b, 0, utt);
    std::vector<Utterance> one{utt};
    std::vector<std::mt19937> rngs;
    {
        std::lock_guard<std::mutex> lk(rng_mu_);
        rngs.emplace_back(rng_());
    }
    const bool any_noise = opt_.temperature > 0 || utt.noise;
    // Same frame body as the batched loop (one graph launch per frame): the EOS logit reaches mapped host memory from
    // inside the flow phase, the new latent is filed under its frame index, and nothing else leaves the device.
    hipStream_t st = b.st;
    PTTS_HIP(hipMemsetAsync(b.step_dev.p, 0, 4, st));
    *b.h_flag = 0;
    for (int i = 0; i < L; ++i) b.h_x0[i] = std::numeric_limits<float>::quiet_NaN();   // BOS marker (:721)
    PTTS_HIP(hipMemcpyAsync(b.seq_in.p, b.h_x0, static_cast<size_t>(L) * 4, hipMemcpyHostToDevice, st));
    launch_prep_latent(b.seq_in.f(), b.xcur.f(), 1, L, 0, hd_.bos, nullptr, st);
    b.x0_src = b.x0.f(); b.x0_stride = L; b.x0_step_stride = 0;   // start points arrive frame by frame
    if (!any_noise) PTTS_HIP(hipMemsetAsync(b.x0.p, 0, static_cast<size_t>(L) * 4, st));
    const bool graph = use_graph_ && !timing_;
    FlagWait flag_wait;
    hipGraphExec_t* exec = nullptr;
    if (graph) {
        const std::vector<int> keep = b.lm_pos_host;
        exec = ensure_frame_graph(b);
        b.lm_pos_host = keep;
    }

    std::vector<float> chunk_audio;
    auto flush = [&](int first, int count, bool is_final) {
        chunk_audio.resize(static_cast<size_t>(count) * fs);
        for (int f0 = 0; f0 < count; f0 += kDecChunk) {  // sub-chunks of <= 15 frames (:796)
            const int n = std::min(kDecChunk, count - f0);
            mimi_decode_dev(b, b.latents.f() + static_cast<size_t>(first + f0) * L, static_cast<long long>(b.lat_frames) * L, n,
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
        lm_bookkeeping(b);
        if (any_noise) {
            // two pinned sets in turn, behind the BOS rows: the copy of frame k - 2 has run by now (its frame's EOS was seen)
            float* hx = b.h_x0 + static_cast<size_t>(1 + (step & 1)) * L;
            fill_noise(rngs, one, step, hx);
            PTTS_HIP(hipMemcpyAsync(b.x0.p, hx, static_cast<size_t>(L) * 4, hipMemcpyHostToDevice, st));
        }
        if (graph) PTTS_HIP(hipGraphLaunch(exec[step & 1], st));                     // :753, :777-784
        else frame_body(b, step & 1);
        wait_flag(b.h_flag, step + 1, st, flag_wait);
        const float e = b.h_eos_all[step];
        if (capture) capture->eos.push_back(e);
        if (e > kEosThreshold && eos_step < 0) eos_step = step;                            // :758
        if (eos_step >= 0 && step >= eos_step + opt_.frames_after_eos) break;              // :763 (this step's flow output is not used)
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
    check_flow_ring(b.flow_ctl.as<unsigned>(), 1);
    if (capture) {
        capture->frames = frames - pending;
        capture->latents.resize(static_cast<size_t>(frames) * L);
        if (frames) sync_copy(capture->latents.data(), b.latents.p, capture->latents.size() * 4, hipMemcpyDeviceToHost);
    }
    return total;
}

// ------------------------------------------------------------------------------------------ voice encoder

std::vector<float> Engine::encode_voice(const float* audio, int S) {
    std::vector<std::vector<float>> out;
    encode_clips(&audio, 1, S, out);
    return std::move(out[0]);
}

void Engine::encode_clips(const float* const* clips, int B, int S, std::vector<std::vector<float>>& out) {
    std::lock_guard<std::mutex> order(enc_mu_);
    encode_enqueue(clips, B, S, 0);
    encode_collect(0, out);
}

// The results of the pass enqueued on `slot`, once its copy to pinned memory has landed.
void Engine::encode_collect(int slot, std::vector<std::vector<float>>& out) {
    EncSlot& E = enc_slot_[slot];
    PTTS_HIP(hipEventSynchronize(E.done));
    out.assign(static_cast<size_t>(E.clips), std::vector<float>(E.row_floats));
    for (int b = 0; b < E.clips; ++b) std::memcpy(out[b].data(), E.pin_out + static_cast<size_t>(b) * E.row_floats, E.row_floats * 4);
}

// B clips of S samples each -> B embeddings [ceil(frames)][d_model] (src/pocket_tts.cpp:241-297 for one clip), enqueued on
// the engine stream with the buffers of `slot` (two slots: the host side of one pass overlaps the GPU side of the other).
// The clips are stacked along the row dimension of every product: the SEANet stages and the transformer's linear layers
// see B times the rows (at one clip the stages behind the second down-sampling run a few hundred workgroups or fewer),
// attention runs per clip.
void Engine::encode_enqueue(const float* const* clips, int B, int S, int slot) {
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    EncSlot& E = enc_slot_[slot];
    DevBuf& enc_arena_ = E.arena;
    float*& enc_pin_ = E.pin_in;
    size_t& enc_pin_bytes_ = E.pin_in_bytes;
    if (!E.done) PTTS_HIP(hipEventCreateWithFlags(&E.done, hipEventDisableTiming));
    const Spec& s = spec();
    if (S <= 0 || B <= 0) throw std::runtime_error("voice clip is empty");
    const int M = s.mimi_dim, U = s.upsample, D = s.d_model, rk = s.res_kernel, lk = s.last_kernel;
    const int ds_splits = 8;
    const size_t Bz = static_cast<size_t>(B);
    // Every buffer of the pass is a slice of one arena that is kept between calls (clips are encoded without a
    // device allocation or a stream synchronisation before the result is copied out).  First the sizes, in the
    // order the slices are taken below.
    auto rounded = [](size_t floats) { return (floats * 4 + 255) & ~static_cast<size_t>(255); };
    size_t need = rounded(Bz * S);
    {
        int c = s.n_filters, T = S;
        need += rounded(Bz * T * c) + rounded(Bz * (rk - 1 + T) * c);
        for (int i = 0; i < 3; ++i) {
            const int r = s.ratios[2 - i], To = (T + r - 1) / r, Tpad = To * r, pn = i < 2 ? rk - 1 : lk - 1;
            need += rounded(Bz * T * (c / 2)) + rounded(Bz * (r + Tpad) * c) +
                    rounded(Bz * To * 2 * c) + rounded(Bz * (pn + To) * 2 * c);
            T = To; c *= 2;
        }
        if (T > 16000) throw std::runtime_error("voice clip too long for the encoder attention buffer");
        if (T > model_->mimi_tab_len) throw std::runtime_error("voice clip exceeds the rotary table");
        const int Nv = (T + U - 1) / U, Tpad = Nv * U;
        const size_t TM = Bz * T * M;
        need += rounded(TM) + 2 * rounded(static_cast<size_t>(s.mimi_layers) * TM) + rounded(Bz) + 3 * rounded(TM) + rounded(3 * TM) +
                rounded(Bz * T * s.mimi_ffn) + rounded(Bz * (U + Tpad) * M) +
                2 * rounded(Bz * (1 + Nv) * M) + rounded(Bz * (1 + Nv) * D) + rounded(static_cast<size_t>(ds_splits) * Bz * (1 + Nv) * M);
    }
    if (enc_arena_.bytes < need) {
        PTTS_HIP(hipStreamSynchronize(stream_));
        enc_arena_.alloc(need);
    }
    if (enc_pin_bytes_ < Bz * S * 4) {
        PTTS_HIP(hipStreamSynchronize(stream_));
        if (enc_pin_) PTTS_HIP(hipHostFree(enc_pin_));
        enc_pin_ = nullptr; enc_pin_bytes_ = 0;
        PTTS_HIP(hipHostMalloc(reinterpret_cast<void**>(&enc_pin_), Bz * S * 4));
        enc_pin_bytes_ = Bz * S * 4;
    }
    size_t top = 0;
    auto take = [&](size_t floats, bool zero = false) -> float* {
        const size_t n = rounded(floats);
        if (top + n > enc_arena_.bytes) throw std::runtime_error("voice encoder arena sized too small");
        float* p = reinterpret_cast<float*>(static_cast<unsigned char*>(enc_arena_.p) + top);
        top += n;
        if (zero) PTTS_HIP(hipMemsetAsync(p, 0, floats * 4, stream_));
        return p;
    };
    // rows [first, first + count) of every clip's segment (seg rows of width floats each) set to zero
    auto zero_rows = [&](float* base, size_t seg, size_t first, size_t count, size_t width) {
        if (count == 0) return;
        PTTS_HIP(hipMemset2DAsync(base + first * width, seg * width * 4, 0, count * width * 4, Bz, stream_));
    };
    float* dau = take(Bz * S);
    for (int b = 0; b < B; ++b) std::memcpy(enc_pin_ + static_cast<size_t>(b) * S, clips[b], static_cast<size_t>(S) * 4);
    PTTS_HIP(hipMemcpyAsync(dau, enc_pin_, Bz * S * 4, hipMemcpyHostToDevice, stream_));
    int c = s.n_filters, T = S;
    float* xraw = take(Bz * T * c);
    float* xe = take(Bz * (rk - 1 + T) * c);
    size_t xe_seg = static_cast<size_t>(rk - 1 + T);            // rows per clip in xe; its first rows are the causal left padding
    int xe_pad = rk - 1;
    // the first convolution and the residual unit behind it in one launch where the shapes allow (64 / 32 channels)
    EncStage0Args e0;
    e0.audio = dau; e0.a_bs = S; e0.T = T; e0.B = B; e0.k = s.kernel; e0.c0 = c; e0.c1 = c / 2; e0.rk = rk;
    e0.w0 = hd_.enc_conv0_w; e0.b0 = hd_.enc_conv0_b;
    e0.w1_hi = hd_.enc_c1[0].lo ? hd_.enc_c1[0].p : nullptr; e0.w1_lo = hd_.enc_c1[0].lo; e0.b1 = hd_.enc_c1_b[0];
    e0.w2_hi = hd_.enc_c2[0].lo ? hd_.enc_c2[0].p : nullptr; e0.w2_lo = hd_.enc_c2[0].lo; e0.b2 = hd_.enc_c2_b[0];
    e0.out = reinterpret_cast<float*>(16); e0.out_bs = 0;       // placeholder for the shape check; set per stage below
    const bool fused0 = !std::getenv("POCKET_TTS_NO_FUSED_ENCODER_STAGE") && enc_stage0_supported(e0);
    if (!fused0) {
        zero_rows(xe, xe_seg, 0, xe_pad, c);
        launch_conv_from_mono(dau, S, nullptr, T, s.kernel, c, hd_.enc_conv0_w, hd_.enc_conv0_b, xraw, static_cast<long long>(T) * c,
                              xe + static_cast<size_t>(xe_pad) * c, static_cast<long long>(xe_seg) * c, B, stream_);
    }
    for (int i = 0; i < 3; ++i) {
        const int r = s.ratios[2 - i], To = (T + r - 1) / r, Tpad = To * r;
        float* h1 = take(Bz * T * (c / 2));
        GemmArgs c1;
        c1.A = xe; c1.lda = c; c1.a_bs = static_cast<long long>(xe_seg) * c; c1.rpb = T;
         c1.bias = hd_.enc_c1_b[i];
        c1.M = B * T; c1.N = c / 2; c1.K = rk * c; c1.C = h1; c1.ldc = c / 2; c1.c_bs = static_cast<long long>(T) * (c / 2); c1.c_act = ACT_ELU;
        hd_.enc_c1[i].bind(c1, c1.M);
        if (!(fused0 && i == 0)) launch_gemm(c1, stream_);
        // rows [r, r + T) of a clip's segment are written by the product below; the padding before and after them stays zero
        const size_t ye_seg = static_cast<size_t>(r + Tpad);
        float* ye = take(Bz * ye_seg * c);
        zero_rows(ye, ye_seg, 0, r, c);
        zero_rows(ye, ye_seg, static_cast<size_t>(r + T), static_cast<size_t>(Tpad - T), c);
        GemmArgs c2;
        c2.A = h1; c2.lda = c / 2; c2.a_bs = static_cast<long long>(T) * (c / 2); c2.rpb = T;
         c2.bias = hd_.enc_c2_b[i];
        c2.M = B * T; c2.N = c; c2.K = c / 2; c2.R = xraw; c2.ldr = c; c2.r_bs = static_cast<long long>(T) * c;
        c2.C = ye + static_cast<size_t>(r) * c; c2.ldc = c; c2.c_bs = static_cast<long long>(ye_seg) * c; c2.c_act = ACT_ELU;
        hd_.enc_c2[i].bind(c2, c2.M);
        if (fused0 && i == 0) {
            e0.out = ye + static_cast<size_t>(r) * c; e0.out_bs = static_cast<long long>(ye_seg) * c;
            launch_enc_stage0(e0, stream_);
        } else {
            launch_gemm(c2, stream_);
        }
        const int pn = i < 2 ? rk - 1 : lk - 1;
        float* xraw2 = take(Bz * To * 2 * c);
        const size_t xe2_seg = static_cast<size_t>(pn + To);
        float* xe2 = take(Bz * xe2_seg * 2 * c);
        zero_rows(xe2, xe2_seg, 0, pn, 2 * c);
        GemmArgs dn;
        dn.A = ye; dn.lda = r * c; dn.a_bs = static_cast<long long>(ye_seg) * c; dn.rpb = To;
         dn.bias = hd_.enc_down_b[i];
        dn.M = B * To; dn.N = 2 * c; dn.K = 2 * r * c; dn.C = xraw2; dn.ldc = 2 * c; dn.c_bs = static_cast<long long>(To) * 2 * c;
        dn.C2 = xe2 + static_cast<size_t>(pn) * 2 * c; dn.ldc2 = 2 * c; dn.c2_bs = static_cast<long long>(xe2_seg) * 2 * c; dn.c2_act = ACT_ELU;
        hd_.enc_down[i].bind(dn, dn.M);
        launch_gemm(dn, stream_);
        xraw = xraw2;
        xe = xe2; xe_seg = xe2_seg; xe_pad = pn;
        T = To; c *= 2;
    }
    float* y = take(Bz * T * M);
    GemmArgs fin;
    fin.A = xe; fin.lda = c; fin.a_bs = static_cast<long long>(xe_seg) * c; fin.rpb = T;
     fin.bias = hd_.enc_final_b;
    fin.M = B * T; fin.N = M; fin.K = lk * c; fin.C = y; fin.ldc = M; fin.c_bs = static_cast<long long>(T) * M;
    hd_.enc_final.bind(fin, fin.M);
    launch_gemm(fin, stream_);
    // transformer over every clip from position 0 (attention per clip)
    const int Nv = (T + U - 1) / U, Tpad = Nv * U;
    const size_t TM = Bz * T * M;
    float* kc = take(static_cast<size_t>(s.mimi_layers) * TM);
    float* vc = take(static_cast<size_t>(s.mimi_layers) * TM);
    int* pos = reinterpret_cast<int*>(take(Bz, true));
    float* sh = take(TM); float* sq = take(TM); float* satt = take(TM);
    float* sqkv = take(3 * TM);
    float* sff = take(Bz * T * s.mimi_ffn);
    TfScratch sc{sh, sqkv, sq, satt, sff};
    const size_t yd_seg = static_cast<size_t>(U + Tpad);
    float* yd = take(Bz * yd_seg * M);
    zero_rows(yd, yd_seg, 0, U, M);
    zero_rows(yd, yd_seg, static_cast<size_t>(U + T), static_cast<size_t>(Tpad - T), M);
    transformer(hd_.enc, kc, vc, 0, static_cast<long long>(TM), T, pos, 0, B, T, y, M, s.mimi_heads, s.mimi_ffn,
                s.mimi_eps, s.mimi_context, model_->mimi_cos, model_->mimi_sin, model_->mimi_tab_len, sc,
                yd + static_cast<size_t>(U) * M, static_cast<long long>(yd_seg) * M, M, stream_);
    // Down-sampling by U with a kernel of 2 U: output row j of a clip reads 2 U rows from row j U of its segment, i.e. two
    // consecutive "rows" of U M floats.  A clip's segment is 1 + Nv such rows, so over the stacked segments row b (1 + Nv) + j
    // is output j of clip b for j < Nv, and row j = Nv (which straddles two clips) is computed and never read.
    const int seg = 1 + Nv, rows = B * seg - 1;
    float* d2 = take(Bz * seg * M);
    float* o = take(Bz * seg * D);
    float* slabs = take(static_cast<size_t>(ds_splits) * Bz * seg * M);
    const int dsK = 2 * U * M;
    if (model_->enc_ds_packed && M <= 1024 && dsK % (ds_splits * 16) == 0 && !std::getenv("POCKET_TTS_NO_SKINNY_ENCODER")) {
        // few rows against a K of 2 * U * M: eight workgroups as a tiled GEMM; as K-slices on the skinny kernel the
        // product fills the chip, and the add & norm kernel sums the slices
        SkinnyArgs k;
        k.A = yd; k.lda = U * M; k.Wp = model_->enc_ds_packed; k.wtype = W_F32; k.M = rows; k.N = M; k.K = dsK; k.splits = ds_splits; k.P = slabs;
        launch_skinny_gemm(k, stream_);
        PTTS_HIP(hipMemsetAsync(d2, 0, static_cast<size_t>(rows) * M * 4, stream_));
        AddNormArgs an;
        an.X = d2; an.ldx = M; an.M = rows; an.D = M; an.P = slabs; an.splits = ds_splits;
        launch_add_norm(an, stream_);
    } else {
        GemmArgs ds;
        ds.A = yd; ds.lda = U * M; ds.W = hd_.enc_ds.p; ds.wtype = hd_.enc_ds.wtype; ds.M = rows; ds.N = M; ds.K = dsK; ds.C = d2; ds.ldc = M;
        launch_gemm(ds, stream_);
    }
    GemmArgs sp;
    sp.A = d2; sp.lda = M;  sp.M = rows; sp.N = D; sp.K = M; sp.C = o; sp.ldc = D;
    hd_.enc_spk.bind(sp, sp.M);
    launch_gemm(sp, stream_);
    check_launches("the voice encoder");
    const size_t row_floats = static_cast<size_t>(Nv) * D;
    if (E.pin_out_bytes < Bz * row_floats * 4) {
        if (E.pin_out) PTTS_HIP(hipHostFree(E.pin_out));
        E.pin_out = nullptr; E.pin_out_bytes = 0;
        PTTS_HIP(hipHostMalloc(reinterpret_cast<void**>(&E.pin_out), Bz * row_floats * 4));
        E.pin_out_bytes = Bz * row_floats * 4;
    }
    PTTS_HIP(hipMemcpy2DAsync(E.pin_out, row_floats * 4, o, static_cast<size_t>(seg) * D * 4, row_floats * 4, Bz, hipMemcpyDeviceToHost, stream_));
    PTTS_HIP(hipEventRecord(E.done, stream_));
    E.clips = B; E.row_floats = row_floats;
}

// Clips of equal length go through the encoder together, up to POCKET_TTS_ENCODE_BATCH (default 8) at a time; while one
// group runs on the GPU the next one is staged and enqueued.
void Engine::encode_voice_batch(const std::vector<const float*>& clips, const std::vector<int>& lens, std::vector<std::vector<float>>& out) {
    static const int max_b = [] { const char* e = std::getenv("POCKET_TTS_ENCODE_BATCH"); return e ? std::max(1, std::min(64, std::atoi(e))) : 8; }();
    out.assign(clips.size(), std::vector<float>());
    std::vector<char> done(clips.size(), 0);
    std::vector<std::vector<size_t>> groups;
    for (size_t i = 0; i < clips.size(); ++i) {
        if (done[i]) continue;
        std::vector<size_t> group{i};
        done[i] = 1;
        for (size_t j = i + 1; j < clips.size() && group.size() < static_cast<size_t>(max_b); ++j)
            if (!done[j] && lens[j] == lens[i]) { group.push_back(j); done[j] = 1; }
        groups.push_back(std::move(group));
    }
    std::lock_guard<std::mutex> order(enc_mu_);   // one caller's groups at a time own the two slots
    auto collect = [&](size_t g) {
        std::vector<std::vector<float>> part;
        encode_collect(static_cast<int>(g & 1), part);
        for (size_t k = 0; k < groups[g].size(); ++k) out[groups[g][k]] = std::move(part[k]);
    };
    for (size_t g = 0; g < groups.size(); ++g) {
        std::vector<const float*> ptrs;
        for (size_t idx : groups[g]) ptrs.push_back(clips[idx]);
        if (g >= 2) collect(g - 2);   // the slot about to be reused
        encode_enqueue(ptrs.data(), static_cast<int>(ptrs.size()), lens[groups[g][0]], static_cast<int>(g & 1));
    }
    for (size_t g = groups.size() >= 2 ? groups.size() - 2 : 0; g < groups.size(); ++g) collect(g);
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
    lm_bookkeeping(*b);
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
    check_flow_ring(b.flow_ctl.as<unsigned>(), B);
}

// W given: quantised here per output row; else q / qscale are the int8 values [N][K] and their per-row scales as they are.
void Engine::q8_linear(const float* A, int M, int K, const float* W, int N, float* out, const signed char* q, const float* qscale, int scheme) {
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    std::vector<float> scale;
    std::vector<signed char> qv;
    if (W) qv = quantize_q8_rows(W, N, K, scale);
    else { qv.assign(q, q + static_cast<size_t>(N) * K); scale.assign(qscale, qscale + N); }
    const std::vector<unsigned char> wq = pack_q8_layout(qv.data(), N, K);
    const std::vector<int> sums = q8_row_sums(qv.data(), N, K, 1);
    DevBuf dA(static_cast<size_t>(M) * K * 4), dW(wq.size()), dS(scale.size() * 4), dC(static_cast<size_t>(M) * N * 4), dU(sums.size() * 4);
    PTTS_HIP(hipMemcpyAsync(dA.p, A, dA.bytes, hipMemcpyHostToDevice, stream_));
    PTTS_HIP(hipMemcpyAsync(dW.p, wq.data(), wq.size(), hipMemcpyHostToDevice, stream_));
    PTTS_HIP(hipMemcpyAsync(dS.p, scale.data(), scale.size() * 4, hipMemcpyHostToDevice, stream_));
    PTTS_HIP(hipMemcpyAsync(dU.p, sums.data(), sums.size() * 4, hipMemcpyHostToDevice, stream_));
    SkinnyArgs g;
    g.A = dA.f(); g.lda = K; g.Wp = dW.p; g.wtype = W_I8; g.wscale = dS.f(); g.M = M; g.N = N; g.K = K; g.C = dC.f(); g.ldc = N;
    if (scheme == 1) { g.q8_act = 1; g.wsum = dU.as<int>(); }
    launch_skinny_gemm(g, stream_);
    PTTS_HIP(hipMemcpyAsync(out, dC.p, dC.bytes, hipMemcpyDeviceToHost, stream_));
    PTTS_HIP(hipStreamSynchronize(stream_));
}

// The bf16-weight GEMM dispatch on its own (test support): C = act(A . W^T + bias) * colscale + R with the options
// that choose between the kernels; flags bit 0: two-term A, bit 1: A rows bf16 in memory, bit 2: C rows bf16.
void Engine::bf16_linear(const float* A, int M, int K, const float* W, int N, const float* bias, const float* colscale, const float* R,
                         int act, int flags, float* out) {
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    const bool abf = flags & 2, cbf = flags & 4;
    std::vector<uint16_t> wq(static_cast<size_t>(N) * K), aq;
    for (size_t i = 0; i < wq.size(); ++i) wq[i] = f32_to_bf16(W[i]);
    DevBuf dA(static_cast<size_t>(M) * K * 4 + 64), dW(wq.size() * 2 + 64), dC(static_cast<size_t>(M) * N * 4 + 64), dB(static_cast<size_t>(N) * 4 + 64),
        dS(static_cast<size_t>(N) * 4 + 64), dR(static_cast<size_t>(M) * N * 4 + 64);
    if (abf) {
        aq.resize(static_cast<size_t>(M) * K);
        for (size_t i = 0; i < aq.size(); ++i) aq[i] = f32_to_bf16(A[i]);
        PTTS_HIP(hipMemcpyAsync(dA.p, aq.data(), aq.size() * 2, hipMemcpyHostToDevice, stream_));
    } else {
        PTTS_HIP(hipMemcpyAsync(dA.p, A, static_cast<size_t>(M) * K * 4, hipMemcpyHostToDevice, stream_));
    }
    PTTS_HIP(hipMemcpyAsync(dW.p, wq.data(), wq.size() * 2, hipMemcpyHostToDevice, stream_));
    if (bias) PTTS_HIP(hipMemcpyAsync(dB.p, bias, static_cast<size_t>(N) * 4, hipMemcpyHostToDevice, stream_));
    if (colscale) PTTS_HIP(hipMemcpyAsync(dS.p, colscale, static_cast<size_t>(N) * 4, hipMemcpyHostToDevice, stream_));
    if (R) PTTS_HIP(hipMemcpyAsync(dR.p, R, static_cast<size_t>(M) * N * 4, hipMemcpyHostToDevice, stream_));
    GemmArgs g;
    g.A = dA.f(); g.lda = K; g.W = dW.p; g.wtype = W_BF16; g.M = M; g.N = N; g.K = K; g.C = dC.f(); g.ldc = N;
    g.bias = bias ? dB.f() : nullptr; g.colscale = colscale ? dS.f() : nullptr;
    if (R) { g.R = dR.f(); g.ldr = N; }
    g.c_act = act; g.split_a = flags & 1; g.a_bf16 = abf; g.c_bf16 = cbf;
    launch_gemm(g, stream_);
    if (cbf) {
        std::vector<uint16_t> cq(static_cast<size_t>(M) * N);
        PTTS_HIP(hipMemcpyAsync(cq.data(), dC.p, cq.size() * 2, hipMemcpyDeviceToHost, stream_));
        PTTS_HIP(hipStreamSynchronize(stream_));
        for (size_t i = 0; i < cq.size(); ++i) {
            const uint32_t u = static_cast<uint32_t>(cq[i]) << 16;
            std::memcpy(out + i, &u, 4);
        }
    } else {
        PTTS_HIP(hipMemcpyAsync(out, dC.p, static_cast<size_t>(M) * N * 4, hipMemcpyDeviceToHost, stream_));
        PTTS_HIP(hipStreamSynchronize(stream_));
    }
}

void Engine::op_mimi_decode(Batch* b, const float* latents, int n, float* audio) {
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    const Spec& s = spec();
    const int L = s.latent, fs = s.frame_size();
    if (n <= 0 || n > kDecChunk) throw std::runtime_error("decoder chunk must hold 1..15 frames");
    if (n > opt_.max_frames) throw std::runtime_error("decoder chunk larger than max_frames");
    // stage the chunk at the front of each utterance's latent buffer
    PTTS_HIP(hipMemcpy2DAsync(b->latents.p, static_cast<size_t>(b->lat_frames) * L * 4, latents, static_cast<size_t>(n) * L * 4,
                              static_cast<size_t>(n) * L * 4, b->B, hipMemcpyHostToDevice, stream_));
    mimi_decode_dev(*b, b->latents.f(), static_cast<long long>(b->lat_frames) * L, n, b->audio.f(), static_cast<long long>(kDecChunk) * fs, stream_);
    PTTS_HIP(hipMemcpy2DAsync(audio, static_cast<size_t>(n) * fs * 4, b->audio.p, static_cast<size_t>(kDecChunk) * fs * 4,
                              static_cast<size_t>(n) * fs * 4, b->B, hipMemcpyDeviceToHost, stream_));
    PTTS_HIP(hipStreamSynchronize(stream_));
}

void Engine::wait_flag(const int* flag, int target, hipStream_t st, FlagWait& w) {
    static const int sleep_pct = [] { const char* e = std::getenv("POCKET_TTS_WAIT_SLEEP_PCT"); return e ? std::max(0, std::min(90, std::atoi(e))) : 0; }();
    static const double floor_us = [] { const char* e = std::getenv("POCKET_TTS_WAIT_LIMIT_MS"); return 1e3 * (e ? std::max(1, std::atoi(e)) : 2000); }();
    const auto t0 = std::chrono::steady_clock::now();
    if (sleep_pct > 0 && w.avg_us > 150.0) std::this_thread::sleep_for(std::chrono::microseconds(static_cast<long>(w.avg_us * sleep_pct / 100.0)));
    const double limit_us = std::max(floor_us, 100.0 * w.avg_us);
    for (unsigned spins = 1; __atomic_load_n(flag, __ATOMIC_ACQUIRE) < target; ++spins) {
        __builtin_ia32_pause();
        if ((spins & 0x3fff) == 0) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipErrorNotReady) (void)hipGetLastError();
            else if (q != hipSuccess) PTTS_HIP(q);   // a real failure (launch error, lost device): not something to wait out
            else if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) < target) throw std::runtime_error("frame finished without publishing its EOS logits");
            const double waited = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (waited > limit_us)
                throw std::runtime_error("frame " + std::to_string(target - 1) + " did not publish within " + std::to_string(static_cast<long>(waited / 1e3)) +
                                         " ms (frames published: " + std::to_string(__atomic_load_n(flag, __ATOMIC_ACQUIRE)) + ", stream still busy, flow-head error word " +
                                         std::to_string(static_cast<unsigned>(__atomic_load_n(flag + 1, __ATOMIC_ACQUIRE))) + ")");
        }
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    w.avg_us = w.avg_us > 0 ? 0.9 * w.avg_us + 0.1 * us : us;
}

}  // namespace ptts
