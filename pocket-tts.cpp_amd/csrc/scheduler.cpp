// Host scheduler: a slot table with refill (continuous batching) behind the single-utterance entry points.
//
// The reference serves one utterance per call on one thread (src/pocket_tts.cpp:548-657 offline, :686-979 streaming) and
// is not thread-safe.  Utterances are independent of each other (every call creates its own state, :563, :710, :737), so
// here any number of threads may call on one handle and their utterances share the GPU:
//
//   * one worker thread per engine steps a table of kSlots utterance slots, one frame of every occupied slot per step
//     (the same captured frame body as the lock-step batch, launched over the occupied prefix of the table);
//   * an utterance is prefilled into a staging batch, its KV rows are copied into a free slot and it joins at the next
//     step.  Offline utterances join at multiples of 15 steps of the table's clock, so that their 15-frame decoder chunks
//     (:461) fall on the same steps and one decoder call serves all of them; streaming utterances join at multiples of
//     their chunk size for the same reason once a quarter of the table is taken, and at once below that;
//   * when an utterance's stop rule fires (:596-603) or it reaches max_frames, its remaining frames are decoded, its slot is
//     free again, and the next queued utterance takes it: nobody waits for the longest member of a batch;
//   * decoder chunks are issued per step for the slots that are due, one call per chunk length over the slot range that
//     holds them; slots inside the range that are not due are computed along and keep their streaming state (due mask);
//   * PCM leaves through pinned slabs; a second thread per engine waits for each slab's copy and hands the samples to
//     their owners, so the stepping thread never copies audio; a caller of the streaming API gets its chunks through a
//     queue and runs its callback on its own thread, as in the reference (src/pocket_tts_c.cpp:183-185).
//
// Start points, latents and the EOS rows live in rings indexed by the table's step counter (kept on the device), so the
// captured frame graph never changes while utterances come and go.
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>

#include "engine.h"

namespace ptts {

namespace {
constexpr int kDecChunk = 15;            // src/pocket_tts.cpp:461
constexpr float kEosThreshold = -4.0f;   // src/pocket_tts.cpp:596
constexpr int kSlabs = 8;                // pinned PCM slabs in flight
constexpr int kWidths[] = {1, 2, 4, 8, 16, 32, 48, 64};   // launch widths of the frame body (occupied prefix, rounded up)

// A hand-off of the persistent flow head gave up during the last frame steps: the utterances in the table have stale
// latents and fail; utterances still queued did nothing wrong and are served by the next activation (launch chain).
struct RingTimeout : std::runtime_error { using std::runtime_error::runtime_error; };
}  // namespace

struct Engine::Request {
    Utterance utt;                       // token ids copied; voice / noise memory stays the caller's (it blocks until done)
    bool want_latents = false;
    bool streaming = false;
    int chunk_frames = 0;
    const std::atomic<bool>* cancel = nullptr;
    bool poll_cancel = false, want_progress = false;
    UtteranceResult* out = nullptr;
    std::vector<float> noise;            // start points drawn at submission (temperature > 0 and none given)
    // ---- worker side
    // decoder items not yet handed over: counted by the stepping thread when an item becomes due (before `closed` is
    // set), counted down by the delivery thread; whoever brings it to zero on a closed request finishes the request
    std::atomic<int> outstanding{0};
    std::atomic<bool> closed{false};     // no further decode will be issued
    std::atomic<bool> abandon{false};    // the caller's callback threw: stop at the next step, deliver nothing more
    std::vector<float> chunk_acc;        // streaming: sub-chunks of one callback's audio (delivery thread only)
    // ---- shared with the caller (mu)
    struct Event { int kind = 0; std::vector<float> samples; bool is_final = false; int frame = 0; };   // kind 0: chunk, 1: progress
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Event> events;
    bool done = false;
    std::exception_ptr err;
};

struct Engine::Table {
    struct Slot {
        Request* req = nullptr;
        int g0 = 0, frames = 0, eos_step = -1, decoded = 0;
    };
    struct Item { Request* req; int slot, f0, n; bool last_sub, is_final; };
    struct Delivery { hipEvent_t ev = nullptr; int slab = -1; int slot0 = 0, n = 0; bool latents = false; std::vector<Item> items; };

    std::unique_ptr<Batch> b, pb;        // the table's state, and the staging batch prefill passes run in
    // Utterances whose prefill is in the staging batch (row i of it) and that wait for a slot or for their step of the clock.
    // While other slots are stepping, the pass runs on the prefill stream beside the frame loop instead of in front of it.
    std::vector<Request*> staged;
    std::vector<char> staged_in;         // row already moved into a slot
    std::vector<int> staged_len;
    std::vector<unsigned char> staged_noise;
    bool staged_side = false, admit_pending = false;
    hipEvent_t staged_done = nullptr, admit_done = nullptr;   // end of the pass on the prefill stream; last move out of the staging batch
    std::array<Slot, kSlots> slots;
    int occupied = 0;
    bool graphs_ready = false;
    float* slabs[kSlabs] = {};           // pinned [kSlots][15 * frame] PCM
    float* lat_slabs[kSlabs] = {};       // pinned [kSlots][15][L]: the chunk's latent rows, for callers that asked for them
    // dec_mark[m % 4] closes the decoder stream's work of steps <= 15 m + 14: the frame stream waits for the mark of two
    // periods ago before it starts a period, so it never writes a row of the latent ring that a queued decoder call still reads
    hipEvent_t dec_mark[4] = {};
    int* h_small = nullptr;                 // pinned [3][kSlots]: voice rows, tokens, lengths of the utterances being admitted
    // pinned staging of an admission's voice rows, token ids and start points (one upload each instead of one per utterance)
    float* h_voice = nullptr; size_t h_voice_rows = 0;
    long long* h_ids = nullptr;
    float* h_noise = nullptr;
    DevBuf noise_stage;
    // ---- shared between the stepping thread and the delivery thread (d_mu)
    std::mutex d_mu;
    std::condition_variable d_cv;
    std::deque<Delivery> deliveries;
    int in_flight = 0;                   // deliveries popped by the delivery thread and not yet handed over
    bool slab_busy[kSlabs] = {};
    std::vector<hipEvent_t> free_events;
    bool d_stop = false;
    std::exception_ptr d_err;            // a HIP failure seen by the delivery thread
    // the requests of this step's decoder items, in issue order, and how many of them have reached `deliveries`
    std::vector<Request*> due_reqs;
    size_t due_pushed = 0;
    std::thread d_thread;
    int fs = 0, R = 0, L = 0, F = 0, device = 0;
    double wait_limit_ms = 20000.0;      // ten times the frame wait's floor (POCKET_TTS_WAIT_LIMIT_MS)

    void deliver_main();
    static void finish(Request* r, std::exception_ptr err = nullptr) {
        std::lock_guard<std::mutex> lk(r->mu);
        if (err && !r->err) r->err = err;
        r->done = true;
        r->cv.notify_all();
    }
    ~Table() {
        {
            std::lock_guard<std::mutex> lk(d_mu);
            d_stop = true;
            d_cv.notify_all();
        }
        if (d_thread.joinable()) d_thread.join();
        for (auto& s : slabs) if (s) (void)hipHostFree(s);
        for (auto& p : lat_slabs) if (p) (void)hipHostFree(p);
        for (auto& e : dec_mark) if (e) (void)hipEventDestroy(e);
        if (h_small) (void)hipHostFree(h_small);
        if (h_voice) (void)hipHostFree(h_voice);
        if (h_ids) (void)hipHostFree(h_ids);
        if (h_noise) (void)hipHostFree(h_noise);
        if (staged_done) (void)hipEventDestroy(staged_done);
        if (admit_done) (void)hipEventDestroy(admit_done);
        for (auto& d : deliveries) if (d.ev) (void)hipEventDestroy(d.ev);
        for (auto& e : free_events) (void)hipEventDestroy(e);
    }
};

// Completed deliveries, oldest first: PCM from the pinned slab to its owner (result vector or streaming chunk).
void Engine::Table::deliver_main() {
    (void)hipSetDevice(device);
    for (;;) {
        Delivery d;
        {
            std::unique_lock<std::mutex> lk(d_mu);
            d_cv.wait(lk, [&] { return d_stop || !deliveries.empty(); });
            if (deliveries.empty()) return;   // stopping, nothing left
            d = std::move(deliveries.front());
            deliveries.pop_front();
            ++in_flight;
        }
        std::exception_ptr err;
        // bounded wait: a decoder chunk takes milliseconds; one that has not landed after wait_limit_ms fails its owners
        hipError_t e = hipEventQuery(d.ev);
        for (auto t0 = std::chrono::steady_clock::now(); e == hipErrorNotReady; e = hipEventQuery(d.ev)) {
            (void)hipGetLastError();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms > wait_limit_ms) break;
            if (ms > 0.2) std::this_thread::sleep_for(std::chrono::microseconds(50));
            else __builtin_ia32_pause();
        }
        if (e == hipErrorNotReady) err = std::make_exception_ptr(std::runtime_error("a decoder chunk did not finish within " + std::to_string(static_cast<long>(wait_limit_ms)) + " ms"));
        else if (e != hipSuccess) err = std::make_exception_ptr(std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e) + " while waiting for a decoder chunk"));
        PTTS_TRACEF("delivery: %d frames, %d items", d.n, static_cast<int>(d.items.size()));
        for (auto& it : d.items) {
            Request* r = it.req;
            if (!err && it.n > 0) {
                const float* src = slabs[d.slab] + static_cast<size_t>(it.slot - d.slot0) * d.n * fs;
                if (!r->streaming) {
                    std::vector<float>& a = r->out->audio;
                    if (a.capacity() < static_cast<size_t>(F) * fs) a.reserve(static_cast<size_t>(F) * fs);   // one allocation per utterance
                    if (a.size() < static_cast<size_t>(it.f0 + it.n) * fs) a.resize(static_cast<size_t>(it.f0 + it.n) * fs);
                    std::memcpy(a.data() + static_cast<size_t>(it.f0) * fs, src, static_cast<size_t>(it.n) * fs * 4);
                } else {
                    r->chunk_acc.insert(r->chunk_acc.end(), src, src + static_cast<size_t>(it.n) * fs);
                    if (it.last_sub) {
                        Request::Event ev;
                        ev.samples.swap(r->chunk_acc);
                        ev.is_final = it.is_final;
                        std::lock_guard<std::mutex> lk(r->mu);
                        r->events.push_back(std::move(ev));
                        r->cv.notify_all();
                    }
                }
            }
            if (!err && it.n > 0 && d.latents && r->want_latents) {
                std::vector<float>& z = r->out->latents;
                if (z.size() < static_cast<size_t>(it.f0 + it.n) * L) z.resize(static_cast<size_t>(it.f0 + it.n) * L);
                std::memcpy(z.data() + static_cast<size_t>(it.f0) * L, lat_slabs[d.slab] + static_cast<size_t>(it.slot - d.slot0) * d.n * L,
                            static_cast<size_t>(it.n) * L * 4);
            }
            if (err) {
                // the request fails once, at its last item; its earlier items are dropped
                std::lock_guard<std::mutex> lk(r->mu);
                if (!r->err) r->err = err;
            }
            if (r->outstanding.fetch_sub(1) == 1 && r->closed.load()) finish(r);
        }
        std::lock_guard<std::mutex> lk(d_mu);
        if (err && !d_err) d_err = err;
        if (d.slab >= 0) slab_busy[d.slab] = false;
        free_events.push_back(d.ev);
        --in_flight;
        d_cv.notify_all();
    }
}

void Engine::validate_request(const Utterance& utt) const {
    if (utt.token_ids.empty()) throw std::runtime_error("Text cannot be empty");
    if (static_cast<int>(utt.token_ids.size()) > opt_.max_text_tokens) throw std::runtime_error("text has more tokens than the engine was sized for");
    if (utt.voice_frames < 0 || (utt.voice_frames > 0 && !utt.voice)) throw std::runtime_error("voice rows missing");
    // the same capacity every Batch is built with (voice + text + frames, and room for a finished slot's idle steps)
    int cap = 64 + opt_.max_text_tokens + opt_.max_frames + 8;
    cap = (cap + 63) / 64 * 64;
    const long long need = static_cast<long long>(utt.voice_frames) + static_cast<long long>(utt.token_ids.size());
    if (need + opt_.max_frames > cap || need > cap)
        throw std::runtime_error("LM context exceeds the KV-cache capacity (voice + text + frames too long)");
}

void Engine::submit(Request& r) {
    Request* one = &r;
    submit_all(&one, 1);
}

// Validates every request, then prepares and queues them in waves of one table's worth: a burst of up to 64 joins the
// table in one admission, and the start points of a longer queue are drawn while the table already serves the first wave
// (4000 normal deviates per utterance at 125 frames: 0.1 ms each on one core, drawn on up to 8 threads).
void Engine::submit_all(Request* const* reqs, size_t n) {
    for (size_t i = 0; i < n; ++i) validate_request(reqs[i]->utt);   // fail before anything is queued
    PTTS_TRACEF("submit: %d (%s)", static_cast<int>(n), n && reqs[0]->streaming ? "stream" : "offline");
    const int L = spec().latent;
    const size_t per = static_cast<size_t>(opt_.max_frames) * L;
    const float sd = std::sqrt(std::max(opt_.temperature, 0.0f));
    auto draw = [&](Request& r, uint32_t seed) {
        // N(0, sqrt(T)) from a fresh distribution object per frame (src/pocket_tts.cpp:606-613), drawn up front
        r.noise.resize(per);
        std::mt19937 gen(seed);
        for (int f = 0; f < opt_.max_frames; ++f) {
            std::normal_distribution<float> dist(0.0f, sd);
            for (int k = 0; k < L; ++k) r.noise[static_cast<size_t>(f) * L + k] = dist(gen);
        }
    };
    for (size_t first = 0; first < n; first += kSlots) {
        const size_t count = std::min(n - first, static_cast<size_t>(kSlots));
        std::vector<Request*> need;
        std::vector<uint32_t> seeds;
        if (opt_.temperature > 0) {
            std::lock_guard<std::mutex> lk(rng_mu_);
            for (size_t i = 0; i < count; ++i)
                if (!reqs[first + i]->utt.noise) { need.push_back(reqs[first + i]); seeds.push_back(static_cast<uint32_t>(rng_())); }   // seeds in request order
        }
        const size_t workers = std::min<size_t>({need.size() / 4, 8, std::max(1u, std::thread::hardware_concurrency())});
        if (workers > 1) {
            std::vector<std::thread> pool;
            for (size_t w = 0; w < workers; ++w)
                pool.emplace_back([&, w] { for (size_t i = w; i < need.size(); i += workers) draw(*need[i], seeds[i]); });
            for (auto& t : pool) t.join();
        } else {
            for (size_t i = 0; i < need.size(); ++i) draw(*need[i], seeds[i]);
        }
        std::unique_lock<std::mutex> lk(q_mu_);
        if (worker_stop_) {
            if (first == 0) throw std::runtime_error("engine is shutting down");
            // earlier waves are already queued and will be answered by the worker: the rest is answered here
            lk.unlock();
            for (size_t i = first; i < n; ++i) Table::finish(reqs[i], std::make_exception_ptr(std::runtime_error("engine is shutting down")));
            return;
        }
        if (!worker_started_) {
            worker_started_ = true;
            worker_ = std::thread([this] { worker_main(); });
        }
        for (size_t i = 0; i < count; ++i) queue_.push_back(reqs[first + i]);
        q_cv_.notify_all();
    }
}

namespace {
struct PendingGuard {
    std::atomic<int>& c; int n;
    PendingGuard(std::atomic<int>& counter, int k) : c(counter), n(k) { c.fetch_add(n); }
    ~PendingGuard() { c.fetch_sub(n); }
};
}  // namespace

void Engine::generate_one(const Utterance& utt, UtteranceResult& out, bool want_latents) {
    PendingGuard pend(pending_, 1);
    Request r;
    r.utt = utt;
    r.want_latents = want_latents;
    r.out = &out;
    out = UtteranceResult();
    submit(r);
    std::unique_lock<std::mutex> lk(r.mu);
    r.cv.wait(lk, [&] { return r.done; });
    if (r.err) std::rethrow_exception(r.err);
}

void Engine::generate_queue(const std::vector<Utterance>& utts, std::vector<UtteranceResult>& out, bool want_latents) {
    PendingGuard pend(pending_, static_cast<int>(utts.size()));
    out.assign(utts.size(), UtteranceResult());
    std::vector<std::unique_ptr<Request>> reqs;
    std::vector<Request*> ptrs;
    for (size_t i = 0; i < utts.size(); ++i) {
        reqs.emplace_back(new Request());
        reqs.back()->utt = utts[i];
        reqs.back()->want_latents = want_latents;
        reqs.back()->out = &out[i];
        ptrs.push_back(reqs.back().get());
    }
    if (ptrs.empty()) return;
    submit_all(ptrs.data(), ptrs.size());
    std::exception_ptr first;
    for (auto& r : reqs) {
        std::unique_lock<std::mutex> lk(r->mu);
        r->cv.wait(lk, [&] { return r->done; });
        if (r->err && !first) first = r->err;
    }
    if (first) std::rethrow_exception(first);
}

int Engine::generate_streaming(const Utterance& utt, int chunk_frames, const ChunkCallback& cb, const std::atomic<bool>* cancel_flag,
                               bool poll_cancel, const std::function<void(int)>& progress, UtteranceResult* capture) {
    if (!cb) throw std::invalid_argument("Callback must be provided");   // src/pocket_tts.cpp:693-695
    if (std::getenv("POCKET_TTS_INLINE_STREAMING")) return generate_streaming_inline(utt, chunk_frames, cb, cancel_flag, poll_cancel, progress, capture);
    if (chunk_frames <= 0) chunk_frames = 1;
    PendingGuard pend(pending_, 1);
    Request r;
    UtteranceResult local;
    r.utt = utt;
    r.streaming = true;
    r.chunk_frames = chunk_frames;
    r.cancel = cancel_flag;
    r.poll_cancel = poll_cancel;
    r.want_progress = static_cast<bool>(progress);
    r.want_latents = capture != nullptr;
    r.out = capture ? capture : &local;
    *r.out = UtteranceResult();
    submit(r);
    int total = 0;
    bool stopped = false;
    // r and `local` live on this stack frame while the stepping and delivery threads hold pointers to them: nothing may
    // unwind out of this loop before r.done.  A callback that throws abandons the stream (the worker stops it at its next
    // step), the remaining events are drained and dropped, and the exception is rethrown afterwards.
    std::exception_ptr user_err;
    for (;;) {
        Request::Event ev;
        {
            std::unique_lock<std::mutex> lk(r.mu);
            r.cv.wait(lk, [&] { return r.done || !r.events.empty(); });
            if (r.events.empty()) break;   // done and drained
            ev = std::move(r.events.front());
            r.events.pop_front();
        }
        if (user_err) continue;
        try {
            if (ev.kind == 1) {
                if (!stopped && progress) progress(ev.frame);                                  // :891-893
                continue;
            }
            // a cancel request ends the stream at the next frame when polling is on (:745-750) and always drops the tail flush (:897)
            const bool cancelled = cancel_flag && cancel_flag->load();
            if (stopped || (cancelled && (poll_cancel || ev.is_final))) { stopped = true; continue; }
            if (capture) capture->audio.insert(capture->audio.end(), ev.samples.begin(), ev.samples.end());
            cb(ev.samples.data(), static_cast<int>(ev.samples.size()), ev.is_final);
            total += static_cast<int>(ev.samples.size());
        } catch (...) {
            user_err = std::current_exception();
            r.abandon.store(true);
        }
    }
    if (user_err) std::rethrow_exception(user_err);
    if (r.err) std::rethrow_exception(r.err);
    return total;
}

Engine::SchedulerStats Engine::scheduler_stats() {
    std::lock_guard<std::mutex> lk(q_mu_);
    return sched_stats_;
}

void Engine::stop_worker() {
    {
        std::lock_guard<std::mutex> lk(q_mu_);
        worker_stop_ = true;
        q_cv_.notify_all();
    }
    if (worker_.joinable()) worker_.join();
}

void Engine::drop_table() { delete table_; table_ = nullptr; }

void Engine::worker_main() {
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(q_mu_);
            q_cv_.wait(lk, [&] { return worker_stop_ || !queue_.empty(); });
            if (worker_stop_ && queue_.empty()) return;
        }
        try {
            PTTS_TRACE("worker: activation wanted");
            std::lock_guard<std::recursive_mutex> call(call_mu_);   // a lock-step generate() and the table take turns
            PTTS_TRACE("worker: activation begins");
            run_table();
            PTTS_TRACE("worker: activation ended");
        } catch (...) {
            PTTS_TRACE("worker: activation failed");
            // a failure of the table itself (a HIP error): everyone waiting learns of it
            const std::exception_ptr err = std::current_exception();
            bool keep_queue = false;
            try { std::rethrow_exception(err); } catch (const RingTimeout&) { keep_queue = true; } catch (...) {}
            std::vector<Request*> victims;
            if (!keep_queue) {
                std::lock_guard<std::mutex> lk(q_mu_);
                victims.assign(queue_.begin(), queue_.end());
                queue_.clear();
            }
            if (table_) {
                Table& T = *table_;
                {   // what the decoder stream still owes is handed over (or failed) by the delivery thread first
                    std::unique_lock<std::mutex> lk(T.d_mu);
                    T.d_cv.wait(lk, [&] { return T.deliveries.empty() && T.in_flight == 0; });
                }
                // left over: utterances still in a slot, and those with decoder items that were never issued; neither can
                // have been finished by the delivery thread
                for (size_t j = T.due_pushed; j < T.due_reqs.size(); ++j)
                    if (std::find(victims.begin(), victims.end(), T.due_reqs[j]) == victims.end()) victims.push_back(T.due_reqs[j]);
                for (size_t j = 0; j < T.staged.size(); ++j)
                    if (!T.staged_in[j] && std::find(victims.begin(), victims.end(), T.staged[j]) == victims.end()) victims.push_back(T.staged[j]);
                T.staged.clear(); T.staged_in.clear();
                for (auto& sl : T.slots)
                    if (sl.req) {
                        if (std::find(victims.begin(), victims.end(), sl.req) == victims.end()) victims.push_back(sl.req);
                        sl.req = nullptr;
                    }
                // this engine's streams only: a device-wide wait is refused while another engine's thread captures a graph
                (void)hipStreamSynchronize(stream_);
                (void)hipStreamSynchronize(dec_stream_);
                (void)hipStreamSynchronize(prefill_stream_);
                (void)hipGetLastError();
                drop_table();
            }
            for (Request* r : victims) Table::finish(r, err);
        }
    }
}

// One activation of the table: from the first queued utterance until the table and the queue are empty.
void Engine::run_table() {
    const Spec& s = spec();
    const int L = s.latent, D = s.d_model, fs = s.frame_size(), F = opt_.max_frames;
    std::unique_lock<std::mutex> gpu(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    if (!table_) {
        table_ = new Table();
        Table& N = *table_;
        N.b = std::make_unique<Batch>(*this, kSlots);
        N.pb = std::make_unique<Batch>(*this, kSlots, true);
        N.fs = fs; N.R = N.b->lat_frames; N.L = L; N.F = opt_.max_frames; N.device = opt_.device;
        if (const char* e = std::getenv("POCKET_TTS_WAIT_LIMIT_MS")) N.wait_limit_ms = 10.0 * std::max(1, std::atoi(e));
        for (auto& p : N.slabs) PTTS_HIP(hipHostMalloc(reinterpret_cast<void**>(&p), static_cast<size_t>(kSlots) * kDecChunk * fs * 4));
        for (auto& p : N.lat_slabs) PTTS_HIP(hipHostMalloc(reinterpret_cast<void**>(&p), static_cast<size_t>(kSlots) * kDecChunk * L * 4));
        for (auto& e : N.dec_mark) PTTS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        N.b->noise_all.alloc(static_cast<size_t>(kSlots) * N.R * L * 4);
        N.noise_stage.alloc(static_cast<size_t>(kSlots) * F * L * 4);
        PTTS_HIP(hipHostMalloc(reinterpret_cast<void**>(&N.h_small), static_cast<size_t>(3) * kSlots * 4));
        PTTS_HIP(hipHostMalloc(reinterpret_cast<void**>(&N.h_ids), static_cast<size_t>(kSlots) * opt_.max_text_tokens * 8));
        PTTS_HIP(hipHostMalloc(reinterpret_cast<void**>(&N.h_noise), static_cast<size_t>(kSlots) * F * L * 4));
        N.pb->ids_stage.alloc(static_cast<size_t>(kSlots) * opt_.max_text_tokens * 8);
        PTTS_HIP(hipEventCreateWithFlags(&N.staged_done, hipEventDisableTiming));
        PTTS_HIP(hipEventCreateWithFlags(&N.admit_done, hipEventDisableTiming));
        N.d_thread = std::thread([&N] { N.deliver_main(); });
    }
    Table& T = *table_;
    Batch& b = *T.b;
    Batch& pb = *T.pb;
    hipStream_t st = stream_, dst = dec_stream_;
    b.st = st; b.dec_st = dst; b.ready = latents_ready_;
    pb.st = st; pb.dec_st = st;
    const int R = b.lat_frames;
    b.step_ring = R;
    b.eos_ring = std::max(F, 16);
    b.x0_src = b.noise_all.f(); b.x0_stride = static_cast<long long>(R) * L; b.x0_step_stride = L;
    // positions and the step counter start at zero; the decoder histories of a slot are cleared when an utterance takes it
    PTTS_HIP(hipMemsetAsync(b.lm_pos.p, 0, b.lm_pos.bytes, st));
    PTTS_HIP(hipMemsetAsync(b.dec_pos.p, 0, b.dec_pos.bytes, st));
    std::fill(b.dec_pos_host.begin(), b.dec_pos_host.end(), 0);
    PTTS_HIP(hipMemsetAsync(b.step_dev.p, 0, 4, st));
    PTTS_HIP(hipMemsetAsync(b.xcur.p, 0, b.xcur.bytes, st));
    PTTS_HIP(hipStreamSynchronize(dst));
    *b.h_flag = 0;
    for (auto& sl : T.slots) { sl.req = nullptr; }
    T.occupied = 0;
    T.due_reqs.clear(); T.due_pushed = 0;
    T.staged.clear(); T.staged_in.clear(); T.staged_len.clear(); T.staged_noise.clear(); T.staged_side = false; T.admit_pending = false;
    if (!bos_row_.p) {
        bos_row_.alloc(static_cast<size_t>(L) * 4);
        std::vector<float> nanrow(L, std::numeric_limits<float>::quiet_NaN());
        PTTS_HIP(hipMemcpyAsync(b.seq_in.p, nanrow.data(), static_cast<size_t>(L) * 4, hipMemcpyHostToDevice, st));
        launch_prep_latent(b.seq_in.f(), bos_row_.f(), 1, L, 0, hd_.bos, nullptr, st);   // NaN marker -> BOS embedding (:579)
        PTTS_HIP(hipStreamSynchronize(st));
    }
    const bool graph = use_graph_ && !timing_;
    const size_t kvb = b.kv16 ? 2 : 4;
    if (graph && !T.graphs_ready) {
        // every launch width's frame graph now, so that no caller's first chunk pays for a capture later
        for (int w : kWidths) { b.Bw = w; (void)ensure_frame_graph(b); }
        T.graphs_ready = true;
    }
    {
        std::lock_guard<std::mutex> lk(q_mu_);
        sched_stats_.batches += 1;
    }
    // the decoder's streaming state: the rows in front of each chunk area are all that carries over from chunk to chunk
    DecResetArgs dr;
    {
        const int M = s.mimi_dim, U = s.upsample, T0max = kDecChunk * U;
        auto add = [&](DevBuf& buf, size_t per_slot_bytes, size_t hist_bytes) {
            dr.buf[dr.nbuf] = static_cast<char*>(buf.p); dr.per_slot[dr.nbuf] = static_cast<long long>(per_slot_bytes);
            dr.hist[dr.nbuf] = static_cast<int>(hist_bytes); ++dr.nbuf;
        };
        add(b.qbuf, static_cast<size_t>(1 + kDecChunk) * M * 4, static_cast<size_t>(M) * 4);
        add(b.ybuf, static_cast<size_t>(s.kernel - 1 + T0max) * M * 4, static_cast<size_t>(s.kernel - 1) * M * 4);
        int c = 8 * s.n_filters, Tm = T0max;
        add(b.e0buf, static_cast<size_t>(1 + Tm) * c * 4, static_cast<size_t>(c) * 4);
        for (int j = 0; j < 3; ++j) {
            const int co = c / 2, hn = j < 2 ? 1 : s.last_kernel - 1;
            Tm *= s.ratios[j];
            const size_t esz = dec_fused_[j] ? 2 : 4;   // a fused stage keeps bf16 rows in these buffers
            add(b.uebuf[j], static_cast<size_t>(s.res_kernel - 1 + Tm) * co * esz, static_cast<size_t>(s.res_kernel - 1) * co * esz);
            add(b.ebuf[j], static_cast<size_t>(hn + Tm) * co * esz, static_cast<size_t>(hn) * co * esz);
            c = co;
        }
        dr.dec_pos = b.dec_pos.as<int>();
    }

    auto get_event = [&]() {
        hipEvent_t e = nullptr;
        {
            std::lock_guard<std::mutex> lk(T.d_mu);
            if (!T.free_events.empty()) { e = T.free_events.back(); T.free_events.pop_back(); }
        }
        if (!e) PTTS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return e;
    };
    auto take_slab = [&]() {
        std::unique_lock<std::mutex> lk(T.d_mu);
        for (;;) {
            for (int i = 0; i < kSlabs; ++i)
                if (!T.slab_busy[i]) { T.slab_busy[i] = true; return i; }
            T.d_cv.wait(lk);
        }
    };

    int gather_us = 100;
    if (const char* e = std::getenv("POCKET_TTS_GATHER_US")) gather_us = std::max(0, std::atoi(e));
    const bool sync_prefill = std::getenv("POCKET_TTS_SYNC_PREFILL") != nullptr;   // every prefill pass in line, in front of the next frame
    FlagWait flag_wait;
    int gs = 0;   // the table's clock: frame steps launched so far
    for (;;) {
        // ---- admission, in two parts: a prefill pass over queued utterances in the staging batch, then the move of finished
        // rows into free slots.  With an empty table both happen here, in line.  While slots are stepping, the pass runs on
        // the prefill stream beside the frame loop (for up to a table's worth of queued utterances, whether or not a slot is
        // free yet) and its rows move in at the first step of the clock that suits them after it has ended, so nobody's frame
        // waits for somebody else's prompt.
        auto staged_left = [&]() { size_t n = 0; for (char c : T.staged_in) n += c ? 0 : 1; return n; };
        if (!T.staged.empty() && staged_left() == 0) { T.staged.clear(); T.staged_in.clear(); T.staged_len.clear(); T.staged_noise.clear(); T.staged_side = false; }
        if (T.staged.empty()) {
            std::vector<Request*> incoming;
            bool side = T.occupied > 0 && !sync_prefill;
            {
                std::unique_lock<std::mutex> lk(q_mu_);
                // a stream that arrives at a lightly loaded table is prefilled in line after all: its first chunk matters more
                // than the fraction of a millisecond that the few running slots wait for it
                if (side && T.occupied < kSlots / 4)
                    for (Request* r : queue_) if (r->streaming) { side = false; break; }
                if (T.occupied == 0 && gather_us > 0) {
                    // an idle table waits a moment for the rest of a burst: callers that arrive together then share one prefill
                    // pass and one chunk phase instead of trailing each other by a step (POCKET_TTS_GATHER_US, default 100)
                    const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(4 * gather_us);
                    for (size_t seen = queue_.size(); seen < static_cast<size_t>(kSlots) && std::chrono::steady_clock::now() < t_end;) {
                        q_cv_.wait_for(lk, std::chrono::microseconds(gather_us));
                        if (queue_.size() == seen) break;
                        seen = queue_.size();
                    }
                }
                int room = side ? kSlots : kSlots - T.occupied;
                for (auto it = queue_.begin(); it != queue_.end() && room > 0;) {
                    Request* r = *it;
                    // in line, only what can take a slot at this step is prefilled (see the rule at the move below)
                    const int period = r->streaming ? (T.occupied < kSlots / 4 ? 1 : std::min(r->chunk_frames, kDecChunk)) : kDecChunk;
                    if (side || T.occupied == 0 || gs % period == 0) {
                        incoming.push_back(r);
                        it = queue_.erase(it);
                        --room;
                    } else {
                        ++it;
                    }
                }
            }
            if (!incoming.empty()) {
                const int k = static_cast<int>(incoming.size());
                hipStream_t ps = side ? prefill_stream_ : st;
                PTTS_TRACEF("step %d: prefill of %d %s (%d in the table)", gs, k, side ? "beside the frame loop" : "in line", T.occupied);
                // the staging batch and the pinned arrays are free again once the last move out of them has run
                if (side && T.admit_pending) PTTS_HIP(hipStreamWaitEvent(ps, T.admit_done, 0));
                T.admit_pending = false;
                pb.st = ps; pb.dec_st = ps;
                // Prefill of all newcomers in one causal pass over [voice; text] each (the reference's two passes, :572 and :575,
                // write the same cache), padded to the longest, in the staging batch.  Inputs are gathered in pinned memory and
                // uploaded once.
                int vmax = 1, rpb = 1;
                int *nv = T.h_small, *nt = T.h_small + kSlots, *len = T.h_small + 2 * kSlots;
                for (int i = 0; i < k; ++i) {
                    nv[i] = incoming[i]->utt.voice_frames; nt[i] = static_cast<int>(incoming[i]->utt.token_ids.size());
                    len[i] = nv[i] + nt[i];
                    vmax = std::max(vmax, nv[i]); rpb = std::max(rpb, len[i]);
                }
                if (T.h_voice_rows < static_cast<size_t>(vmax)) {
                    if (T.h_voice) { PTTS_HIP(hipStreamSynchronize(st)); PTTS_HIP(hipStreamSynchronize(prefill_stream_)); PTTS_HIP(hipHostFree(T.h_voice)); T.h_voice = nullptr; }
                    PTTS_HIP(hipHostMalloc(reinterpret_cast<void**>(&T.h_voice), static_cast<size_t>(kSlots) * vmax * D * 4));
                    T.h_voice_rows = static_cast<size_t>(vmax);
                    pb.voice_stage.alloc(static_cast<size_t>(kSlots) * vmax * D * 4);
                }
                T.staged = incoming;
                T.staged_in.assign(static_cast<size_t>(k), 0);
                T.staged_len.assign(len, len + k);
                T.staged_noise.assign(static_cast<size_t>(k), 0);
                T.staged_side = side;
                bool any_noise = false;
                for (int i = 0; i < k; ++i) {
                    const Request* r = incoming[i];
                    const Utterance& u = r->utt;
                    if (nv[i]) std::memcpy(T.h_voice + static_cast<size_t>(i) * vmax * D, u.voice, static_cast<size_t>(nv[i]) * D * 4);
                    std::memcpy(T.h_ids + static_cast<size_t>(i) * opt_.max_text_tokens, u.token_ids.data(), static_cast<size_t>(nt[i]) * 8);
                    // start points of frames 0..: explicit ones, drawn ones, or zeros (temperature 0, :606-613)
                    const float* nz = u.noise ? u.noise : (r->noise.empty() ? nullptr : r->noise.data());
                    T.staged_noise[static_cast<size_t>(i)] = nz ? 1 : 0;
                    if (nz) { std::memcpy(T.h_noise + static_cast<size_t>(i) * F * L, nz, static_cast<size_t>(F) * L * 4); any_noise = true; }
                }
                PTTS_HIP(hipMemcpyAsync(pb.voice_stage.p, T.h_voice, static_cast<size_t>(k) * vmax * D * 4, hipMemcpyHostToDevice, ps));
                PTTS_HIP(hipMemcpyAsync(pb.ids_stage.p, T.h_ids, static_cast<size_t>(k) * opt_.max_text_tokens * 8, hipMemcpyHostToDevice, ps));
                if (any_noise) PTTS_HIP(hipMemcpyAsync(T.noise_stage.p, T.h_noise, static_cast<size_t>(k) * F * L * 4, hipMemcpyHostToDevice, ps));
                PTTS_HIP(hipMemcpyAsync(pb.st_nv.p, nv, static_cast<size_t>(k) * 4, hipMemcpyHostToDevice, ps));
                PTTS_HIP(hipMemcpyAsync(pb.st_nt.p, nt, static_cast<size_t>(k) * 4, hipMemcpyHostToDevice, ps));
                PTTS_HIP(hipMemcpyAsync(pb.st_len.p, len, static_cast<size_t>(k) * 4, hipMemcpyHostToDevice, ps));
                PTTS_HIP(hipMemsetAsync(pb.lm_pos.p, 0, pb.lm_pos.bytes, ps));
                std::fill(pb.lm_pos_host.begin(), pb.lm_pos_host.end(), 0);
                pb.ensure_rows(static_cast<size_t>(k) * rpb, D, s.lm_ffn);
                launch_assemble_prefill(pb.voice_stage.f(), static_cast<long long>(vmax) * D, pb.ids_stage.as<long long>(), opt_.max_text_tokens,
                                        pb.st_nv.as<int>(), pb.st_nt.as<int>(), hd_.text_embed, s.vocab + 1, D, rpb, k, pb.x.f(), ps);
                lm_rows(pb, 0, k, rpb, pb.x.f(), pb.st_len.as<int>(), len);
                if (side) PTTS_HIP(hipEventRecord(T.staged_done, ps));
                check_launches("the prefill pass");
            }
        }
        if (!T.staged.empty()) {
            // a pass beside the frame loop must have ended; an empty table has nothing else to do and waits for it
            bool ready = true;
            if (T.staged_side) {
                if (T.occupied == 0) PTTS_HIP(hipEventSynchronize(T.staged_done));
                const hipError_t q = hipEventQuery(T.staged_done);
                if (q == hipErrorNotReady) { ready = false; (void)hipGetLastError(); }
                else if (q != hipSuccess) PTTS_HIP(q);
            }
            // one launch moves every row that can go now into its slot (KV rows, position, BOS row, start points); a second one,
            // on the decoder stream behind whatever it still has queued for the slots' previous occupants, clears their
            // decoder state (the reference zero-initialises every state tensor, :133-186)
            AdmitArgs aa;
            int k = 0, next_free = 0;
            const bool empty = T.occupied == 0;
            for (size_t j = 0; ready && j < T.staged.size() && T.occupied < kSlots; ++j) {
                if (T.staged_in[j]) continue;
                Request* r = T.staged[j];
                // a stream joins at once while the table is lightly loaded (first-chunk latency; its chunks then fall on steps
                // of their own and cost decoder calls of their own), on a chunk boundary of the clock once it is not; offline
                // utterances join on multiples of the decoder chunk, so that one decoder call serves them all
                const int period = r->streaming ? (T.occupied < kSlots / 4 ? 1 : std::min(r->chunk_frames, kDecChunk)) : kDecChunk;
                if (!(empty || gs % period == 0)) continue;
                while (T.slots[next_free].req) ++next_free;
                Table::Slot& S = T.slots[next_free];
                S.req = r; S.g0 = gs; S.frames = 0; S.eos_step = -1; S.decoded = 0;
                ++T.occupied;
                T.staged_in[j] = 1;
                aa.slot[k] = static_cast<unsigned char>(next_free); aa.src[k] = static_cast<unsigned char>(j);
                aa.len[k] = T.staged_len[j]; aa.has_noise[k] = T.staged_noise[j];
                dr.slot[k] = static_cast<unsigned char>(next_free);
                b.dec_pos_host[next_free] = 0;
                ++k;
            }
            if (k > 0) {
                PTTS_TRACEF("step %d: %d admitted (%d in the table)", gs, k, T.occupied - k);
                if (T.staged_side) PTTS_HIP(hipStreamWaitEvent(st, T.staged_done, 0));
                aa.k = k; aa.layers = s.lm_layers; aa.row_bytes = static_cast<int>(D * kvb); aa.cap = b.lm_cap; aa.dst_slots = b.B; aa.src_slots = pb.B;
                aa.k_dst = static_cast<char*>(b.lm_k.p); aa.v_dst = static_cast<char*>(b.lm_v.p);
                aa.k_src = static_cast<const char*>(pb.lm_k.p); aa.v_src = static_cast<const char*>(pb.lm_v.p);
                aa.lm_pos = b.lm_pos.as<int>(); aa.xcur = b.xcur.f(); aa.bos = bos_row_.f(); aa.L = L; aa.ring = R; aa.frames = F; aa.g0 = gs;
                aa.noise_ring = b.noise_all.f(); aa.noise_stage = T.noise_stage.f();
                launch_admit(aa, st);
                PTTS_HIP(hipEventRecord(T.admit_done, st));
                T.admit_pending = true;
                dr.k = k;
                launch_dec_reset(dr, dst);
                check_launches("the admission");
                std::lock_guard<std::mutex> lk(q_mu_);
                sched_stats_.admissions += 1;
                sched_stats_.utterances += k;
                sched_stats_.largest = std::max(sched_stats_.largest, static_cast<long>(T.occupied));
            }
        }
        if (T.occupied == 0) {
            std::lock_guard<std::mutex> lk(q_mu_);
            if (queue_.empty() && staged_left() == 0) break;
            continue;   // something queued that waits for a boundary of an empty table: cannot happen, but never spin on it
        }
        // ---- launch width: the occupied prefix, rounded up to a few sizes so that few graphs exist
        int top = 0;
        unsigned long long idle_mask = 0;
        for (int i = 0; i < kSlots; ++i) if (T.slots[i].req) top = i + 1;
        int width = 1;
        for (int w : kWidths) if (w >= top) { width = w; break; }
        for (int i = 0; i < width; ++i) if (!T.slots[i].req) idle_mask |= 1ull << i;
        // slots that step without an utterance restart at position 0 now and then, so that they never leave their cache
        if (idle_mask && gs % 8 == 0) launch_reset_pos_mask(b.lm_pos.as<int>(), idle_mask, st);
        if (b.Bw != width) b.Bw = width;
        // Step gs writes the ring row of step gs - R.  Whatever reads that row on the decoder stream was issued by step
        // gs - R + max_frames - 1 at the latest (a chunk is never longer than an utterance), and R - max_frames >= 30: the
        // mark recorded at step gs - 16 covers it for this whole period of 15 steps.
        if (gs % kDecChunk == 0 && gs >= 2 * kDecChunk) PTTS_HIP(hipStreamWaitEvent(st, T.dec_mark[(gs / kDecChunk - 2) % 4], 0));
        if (graph) {
            hipGraphExec_t* exec = ensure_frame_graph(b);
            PTTS_HIP(hipGraphLaunch(exec[gs & 1], st));
        } else {
            frame_body(b, gs & 1);
        }
        gpu.unlock();
        // ---- the EOS logits of this step (the flow head of the step is still running when they arrive)
        wait_flag(b.h_flag, gs + 1, st, flag_wait);
        gpu.lock();
        if (__atomic_load_n(b.h_flag + 1, __ATOMIC_ACQUIRE) != 0) {
            // the persistent flow head of an earlier step gave up on a hand-off: every slot's latents since then are stale
            PTTS_HIP(hipStreamSynchronize(st));
            sync_fill(b.flow_ctl.p, 0, 4);
            b.h_flag[1] = 0;
            ring_failed();
            throw RingTimeout("flow head: a hand-off between workgroups timed out at table step " + std::to_string(gs) +
                              " (the persistent launch was not fully resident); the utterances in the table fail, later ones run the head as a launch chain");
        }
        const float* e_now = b.h_eos_all + static_cast<size_t>(gs % b.eos_ring) * b.B;
        struct Due { Request* req; int slot, n, f0, gstart; bool last_sub, is_final, closes; };
        std::vector<Due> due;
        // T.due_reqs follows `due` from the first append on (it is empty here), so that a failure anywhere below finds every
        // request whose slot has already been cleared; it is put into issue order once `due` is sorted
        auto add_due = [&](const Due& d) { due.push_back(d); T.due_reqs.push_back(d.req); };
        long frames_now = 0;
        for (int i = 0; i < width; ++i) {
            Table::Slot& S = T.slots[i];
            Request* r = S.req;
            if (!r) continue;
            const int k = gs - S.g0;          // this slot's own step
            const float e = e_now[i];
            if (r->want_latents) r->out->eos.push_back(e);
            bool stop = false, cancelled = false;
            if (e > kEosThreshold && S.eos_step < 0) S.eos_step = k;                           // :596
            if (S.eos_step >= 0 && k >= S.eos_step + opt_.frames_after_eos) stop = true;       // :601 (this step's flow output is not used)
            if (!stop && r->streaming && r->poll_cancel && r->cancel && r->cancel->load()) { stop = true; cancelled = true; }   // :745-750
            if (!stop && r->abandon.load()) { stop = true; cancelled = true; }
            if (!stop) {
                S.frames = k + 1;
                ++frames_now;
                if (r->want_progress) {
                    Request::Event ev;
                    ev.kind = 1; ev.frame = k + 1;
                    std::lock_guard<std::mutex> lk(r->mu);
                    r->events.push_back(std::move(ev));
                    r->cv.notify_all();
                }
                if (S.frames >= F) stop = true;
            }
            const int pending = S.frames - S.decoded;
            const size_t first_due = due.size();
            if (cancelled || (stop && r->streaming && r->cancel && r->cancel->load())) {
                // a cancelled stream gets no tail flush (:897)
            } else if (r->streaming) {
                if (pending > 0 && (stop || pending >= r->chunk_frames)) {                    // :790; tail flush :897-965
                    // a full chunk is an ordinary flush even when it is the last one; is_final comes only with the tail
                    // flush of a remainder (:879 is unreachable in the reference, :897-965 is what delivers it)
                    const bool tail = pending < r->chunk_frames;
                    for (int f0 = 0; f0 < pending; f0 += kDecChunk) {                          // sub-chunks of <= 15 frames (:796)
                        const int n = std::min(kDecChunk, pending - f0);
                        add_due({r, i, n, S.decoded + f0, S.g0 + S.decoded + f0, f0 + n == pending, tail, false});
                    }
                    S.decoded = S.frames;
                }
            } else if (pending > 0 && (stop || pending == kDecChunk)) {
                add_due({r, i, pending, S.decoded, S.g0 + S.decoded, true, false, false});
                S.decoded = S.frames;
            }
            if (stop) {
                r->out->frames = S.frames;
                // nothing left to decode: the request closes through an item without audio (ordered behind its earlier chunks)
                if (due.size() == first_due) add_due({r, i, 0, S.frames, gs + 1, true, false, false});
                due.back().closes = true;
                // every item of the request is counted before it is marked closed: the delivery thread finishes a closed
                // request when the count reaches zero
                r->outstanding.fetch_add(static_cast<int>(due.size() - first_due));
                r->closed.store(true);
                S.req = nullptr;
                --T.occupied;
                launch_poke_int(b.lm_pos.as<int>() + i, 0, st);   // the slot idles from position 0 until it is taken again
            } else if (due.size() != first_due) {
                r->outstanding.fetch_add(static_cast<int>(due.size() - first_due));
            }
        }
        {
            std::lock_guard<std::mutex> lk(q_mu_);
            sched_stats_.frame_steps += 1;
            sched_stats_.slot_steps += width;
            sched_stats_.frames += frames_now;
        }
        // ---- decoder chunks of this step: one call per (chunk length, ring position) over the slot range that holds the due
        // slots, in order of ring position, which keeps the sub-chunks of one streaming flush in order
        if (!due.empty()) {
            std::stable_sort(due.begin(), due.end(), [](const Due& x, const Due& y) { return x.gstart != y.gstart ? x.gstart < y.gstart : x.n < y.n; });
            T.due_reqs.clear();
            for (auto& q : due) T.due_reqs.push_back(q.req);
            T.due_pushed = 0;
            PTTS_HIP(hipEventRecord(b.ready, st));
            PTTS_HIP(hipStreamWaitEvent(dst, b.ready, 0));
            for (size_t a0 = 0; a0 < due.size();) {
                size_t a1 = a0;
                while (a1 < due.size() && due[a1].gstart == due[a0].gstart && due[a1].n == due[a0].n) ++a1;
                const int n = due[a0].n;
                Table::Delivery dv;
                dv.n = n;
                unsigned long long mask = 0;
                int lo = kSlots, hi = -1;
                for (size_t j = a0; j < a1; ++j) { mask |= 1ull << due[j].slot; lo = std::min(lo, due[j].slot); hi = std::max(hi, due[j].slot); }
                dv.slot0 = lo;
                if (n > 0) {
                    gpu.unlock();
                    dv.slab = take_slab();   // may wait for the delivery thread
                    gpu.lock();
                    float* chunk = b.audio.f() + static_cast<size_t>(lo) * kDecChunk * fs;
                    mimi_decode_dev(b, nullptr, 0, n, chunk, static_cast<long long>(kDecChunk) * fs, dst, lo, hi - lo + 1, due[a0].gstart % R, mask);
                    PTTS_HIP(hipMemcpy2DAsync(T.slabs[dv.slab], static_cast<size_t>(n) * fs * 4, chunk, static_cast<size_t>(kDecChunk) * fs * 4,
                                              static_cast<size_t>(n) * fs * 4, hi - lo + 1, hipMemcpyDeviceToHost, dst));
                    for (size_t j = a0; j < a1; ++j) dv.latents = dv.latents || due[j].req->want_latents;
                    if (dv.latents) {
                        // the chunk's rows of the latent ring (in two pieces where the chunk runs over the end of the ring)
                        const int r0 = due[a0].gstart % R, n1 = std::min(n, R - r0);
                        const float* ring = b.latents.f() + static_cast<size_t>(lo) * R * L;
                        PTTS_HIP(hipMemcpy2DAsync(T.lat_slabs[dv.slab], static_cast<size_t>(n) * L * 4, ring + static_cast<size_t>(r0) * L,
                                                  static_cast<size_t>(R) * L * 4, static_cast<size_t>(n1) * L * 4, hi - lo + 1, hipMemcpyDeviceToHost, dst));
                        if (n1 < n)
                            PTTS_HIP(hipMemcpy2DAsync(T.lat_slabs[dv.slab] + static_cast<size_t>(n1) * L, static_cast<size_t>(n) * L * 4, ring,
                                                      static_cast<size_t>(R) * L * 4, static_cast<size_t>(n - n1) * L * 4, hi - lo + 1,
                                                      hipMemcpyDeviceToHost, dst));
                    }
                    std::lock_guard<std::mutex> lk(q_mu_);
                    sched_stats_.decode_calls += 1;
                }
                for (size_t j = a0; j < a1; ++j) {
                    const Due& q = due[j];
                    dv.items.push_back(Table::Item{q.req, q.slot, q.f0, q.n, q.last_sub, q.is_final});
                }
                dv.ev = get_event();
                PTTS_HIP(hipEventRecord(dv.ev, dst));
                PTTS_TRACEF("step %d: decoder call, %d frames from ring step %d, slots %d..%d, %d due", gs, n, due[a0].gstart, lo, hi, static_cast<int>(a1 - a0));
                {
                    std::lock_guard<std::mutex> lk(T.d_mu);
                    T.deliveries.push_back(std::move(dv));
                    T.due_pushed = a1;
                    T.d_cv.notify_all();
                }
                a0 = a1;
            }
            T.due_reqs.clear(); T.due_pushed = 0;
        }
        if (gs % kDecChunk == kDecChunk - 1) PTTS_HIP(hipEventRecord(T.dec_mark[(gs / kDecChunk) % 4], dst));
        ++gs;
    }
    // ---- the table is empty: everything still in flight lands, then the activation ends
    {
        std::unique_lock<std::mutex> lk(T.d_mu);
        T.d_cv.wait(lk, [&] { return T.deliveries.empty() && T.in_flight == 0; });
        if (T.d_err) { std::exception_ptr e = T.d_err; T.d_err = nullptr; std::rethrow_exception(e); }
    }
    PTTS_HIP(hipStreamSynchronize(st));
    PTTS_HIP(hipStreamSynchronize(dst));
    check_flow_ring(b.flow_ctl.as<unsigned>(), kSlots);
}

}  // namespace ptts
