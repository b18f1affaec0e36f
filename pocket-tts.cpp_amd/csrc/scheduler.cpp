// Host scheduler of the single-utterance entry points.
//
// The reference serves one utterance per call and is not thread-safe (src/pocket_tts.cpp:548-657 runs on the
// instance's sessions and state).  Here any number of threads may call generate on one handle; the GPU engine
// works on lock-step batches, so the calls are coalesced: the first caller that finds no batch in flight becomes
// the leader, takes everything queued so far (at most kMaxCoalesce utterances) and runs it as one batch; callers
// that arrive meanwhile queue up and form the next batch.  Nobody waits for a timer: a lone caller runs at once
// as a batch of one, and under load the batch size grows to whatever accumulates during one batch's run time.
#include "engine.h"

namespace ptts {

void Engine::generate_one(const Utterance& utt, UtteranceResult& out, bool want_latents) {
    // reject what would fail the whole batch before it can join one
    if (utt.token_ids.empty()) throw std::runtime_error("Text cannot be empty");
    if (static_cast<int>(utt.token_ids.size()) > opt_.max_text_tokens) throw std::runtime_error("text has more tokens than the engine was sized for");
    if (utt.voice_frames < 0 || (utt.voice_frames > 0 && !utt.voice)) throw std::runtime_error("voice rows missing");

    Pending me{&utt, &out, want_latents, nullptr, false};
    std::unique_lock<std::mutex> lk(q_mu_);
    queue_.push_back(&me);
    while (!me.done) {
        if (leader_active_) {
            q_cv_.wait(lk);
            continue;
        }
        leader_active_ = true;
        std::vector<Pending*> batch;
        while (!queue_.empty() && batch.size() < kMaxCoalesce) {
            batch.push_back(queue_.front());
            queue_.pop_front();
        }
        lk.unlock();
        std::exception_ptr err;
        try {
            std::vector<Utterance> utts;
            bool want = false;
            for (Pending* p : batch) { utts.push_back(*p->utt); want = want || p->want_latents; }
            std::vector<UtteranceResult> res;
            generate(utts, res, want);
            for (size_t i = 0; i < batch.size(); ++i) *batch[i]->out = std::move(res[i]);
        } catch (...) {
            err = std::current_exception();
        }
        lk.lock();
        for (Pending* p : batch) { p->err = err; p->done = true; }
        sched_stats_.batches += 1;
        sched_stats_.utterances += static_cast<long>(batch.size());
        sched_stats_.largest = std::max(sched_stats_.largest, static_cast<long>(batch.size()));
        leader_active_ = false;
        q_cv_.notify_all();
    }
    lk.unlock();
    if (me.err) std::rethrow_exception(me.err);
}

Engine::SchedulerStats Engine::scheduler_stats() {
    std::lock_guard<std::mutex> lk(q_mu_);
    return sched_stats_;
}

}  // namespace ptts
