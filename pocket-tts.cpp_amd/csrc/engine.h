// Device-side model and the batched inference engine that replaces the reference's five ONNX Runtime
// sessions (/root/reference/src/pocket_tts.cpp:59-63, 89-113) and its per-call state plumbing
// (:133-239).  All recurrent state (KV caches, convolution histories, positions) lives in HBM for the
// life of a batch and is never copied to the host.
#pragma once
#include <cstring>
#include <cstdint>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <exception>
#include <functional>
#include <list>
#include <map>
#include <mutex>
#include <random>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace ptts {

// Throws if a kernel launch since the last call left an error behind (bad grid or LDS size, missing code object).
void check_launches(const char* where);


// float32 -> nearest-even bfloat16 bits
inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40u);  // keep NaN a NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return static_cast<uint16_t>(u >> 16);
}


enum class Precision { FP32, BF16, INT8 };   // INT8: bf16 engine with int8 LM-layer and flow-block matrices

struct DevWeight {
    const void* p = nullptr;
    int wtype = W_F32;
    int N = 0, K = 0;
    const void* frag = nullptr;   // fragment-major copy for the skinny kernels (same type), where the model keeps one
    const void* lo = nullptr;     // W_BF16 with a second term: bf16(W - bf16(W)); p holds bf16(W)  (the voice encoder's matrices)
    const void* f32 = nullptr;    // ... and the f32 matrix itself, for products too small to fill the chip with 128 x 128 tiles
    // operand choice for a product of M rows: two bf16 terms when its 128 x 128 tiles give at least half the CUs a
    // workgroup, the f32 matrix (64 x 64 tiles, f32 MFMA) below that
    void bind(struct GemmArgs& g, int M) const;
};

struct TfLayer {
    const float *ln1g, *ln1b, *ln2g, *ln2b, *ls1, *ls2;
    DevWeight qkv, out, ff1, ff2;
};

// Weights resident on one GPU, in f32 plus (for GEMM operands) the active compute precision.
class Model {
public:
    Model(const Pack& pack, Precision prec);
    const Spec& spec() const { return spec_; }
    Precision precision() const { return prec_; }
    const float* f(const std::string& name) const;  // f32 tensor on device
    DevWeight w(const std::string& name) const;     // GEMM weight [N][K] in the compute precision
    std::vector<TfLayer> tf_layers(const std::string& prefix, int layers, bool has_ls) const;
    size_t gemm_weight_bytes(const std::string& prefix) const;  // bytes of GEMM weights whose name starts with prefix

    // derived tensors
    DevWeight ada_all;           // stacked adaLN weights [blocks*3C + 2C][C]
    const float* ada_all_b = nullptr;
    int mod_width = 0;
    const float *lm_cos = nullptr, *lm_sin = nullptr; int lm_tab_len = 0;
    const float *mimi_cos = nullptr, *mimi_sin = nullptr; int mimi_tab_len = 0;
    FlowHostTabs flow_tabs;      // per-block device pointers of the flow head (host arrays)
    const void *flow_in_packed = nullptr, *flow_out_packed = nullptr;  // fragment-major flow.in_proj.w / flow.final.lin.w
    // fragment-major copies of the weights the batched decode step streams (decode_kernels.hip)
    struct PackedLayer {
        const void *qkv, *out, *ff1, *ff2;
        const float *s_qkv = nullptr, *s_out = nullptr, *s_ff1 = nullptr, *s_ff2 = nullptr;
        const void *qkv_lo = nullptr, *out_lo = nullptr, *ff2_lo = nullptr;   // second planes of two-term bf16 matrices (null: one term)
        // int8 matrices in the dynamic uint8 activation scheme: per K-slice sums of each weight row, indexed by the slice count
        // (1..8 and 16, where that many slices divide K into whole k-steps)
        const int *ws_qkv[17] = {}, *ws_out[17] = {}, *ws_ff1[17] = {}, *ws_ff2[17] = {};
    };
    int q8_act = 0;   // int8 engine's activation scheme (Spec::q8_act_scheme, POCKET_TTS_Q8_ACT=block|dyn overrides)
    std::set<std::string> two_term;   // matrix families held as two bf16 terms (see two_term_families in engine.cpp)
    int lm_wtype = W_F32;        // weight type of lm_packed (W_I8 in int8 mode)
    int flow_block_wtype = W_F32;
    bool flow_ring_split = false;   // float engine: flow_tabs carries two-term bf16 planes for the persistent flow head
    std::vector<PackedLayer> lm_packed;
    const void* enc_ds_packed = nullptr;   // voice encoder's down-sampling matrix, fragment-major f32 (K-slice product of short clips)
    const void* flow_cond_packed = nullptr;
    const void *flow_cond_packed_lo = nullptr, *ada_packed_lo = nullptr;   // second planes (two-term families)
    const void* lm_in_packed = nullptr;  // LM input projection, fragment-major (null when its shape does not tile)
    const void* ada_packed = nullptr;   // stacked adaLN matrix, fragment-major (null when its shape does not tile)
    int flow_wtype = W_F32;

    void build_rope_tables(int lm_len, int mimi_len);

private:
    struct Entry { size_t off_f32 = 0; size_t off_lp = 0; bool has_lp = false; std::vector<int64_t> shape; };
    Spec spec_;
    Precision prec_;
    std::map<std::string, Entry> entries_;
    DevBuf blob_f32_, blob_lp_, rope_;
    std::vector<DevBuf> extra_;
    std::map<std::string, const void*> frag_;   // fragment-major copies by tensor name
    std::map<std::string, std::pair<const void*, const void*>> split_;   // two-term bf16 copies (hi, lo) by tensor name
};

struct EngineOptions {
    std::string pack_path;
    Precision precision = Precision::BF16;
    int device = 0;
    int lsd_steps = 10;
    int max_frames = 500;
    int frames_after_eos = 3;
    float temperature = 0.7f;
    int max_text_tokens = 512;
    uint64_t seed = 0;         // 0: seed from std::random_device like the reference (src/pocket_tts.cpp:72)
    bool has_seed = false;
    int lanes = 0;             // concurrent sub-batches per staged batch (0: POCKET_TTS_LANES or the built-in default)
};

struct Utterance {
    std::vector<int64_t> token_ids;
    const float* voice = nullptr;  // [voice_frames][d_model] host
    int voice_frames = 0;
    const float* noise = nullptr;  // optional [max_frames][latent] host start points (already scaled); overrides temperature
};

struct UtteranceResult {
    int frames = 0;
    std::vector<float> audio;    // frames * frame_size
    std::vector<float> latents;  // frames * latent (filled when requested)
    std::vector<float> eos;      // one logit per LM step run (filled when requested)
};

// Scratch rows of one transformer pass: h [M][d], qkv [M][3d], q [M][d], att [M][d], ff [M][ffn].
struct TfScratch { float *h, *qkv, *q, *att, *ff; };

// (samples, count, is_final)
using ChunkCallback = std::function<void(const float*, int, bool)>;

class Batch;

// A staged batch: the utterances cut into contiguous parts, each part a lock-step Batch that runs on its own
// stream ("lane").  The frame loop is a chain of small dependent launches that leaves most of the chip idle,
// so several parts in flight at once multiply throughput at nearly unchanged per-frame latency.
struct Job {
    std::vector<Batch*> parts;
    std::vector<int> first;   // index of each part's first utterance
    int B = 0;
};

class Engine {
public:
    explicit Engine(const EngineOptions& opt);
    ~Engine();
    const Spec& spec() const { return model_->spec(); }
    const EngineOptions& options() const { return opt_; }
    const Model& model() const { return *model_; }
    hipStream_t stream() const { return stream_; }

    // ---- whole-path entry points (lock-step batch of independent utterances) ----
    // Offline: /root/reference/src/pocket_tts.cpp:548-657 for every utterance of the batch.
    void generate(const std::vector<Utterance>& utts, std::vector<UtteranceResult>& out, bool want_latents);
    // One utterance from any thread, through the slot table (scheduler.cpp): a worker thread steps up to kSlots utterances
    // per frame; an utterance takes a free slot when it arrives (offline calls at the next 15-frame boundary of the table's
    // clock, so that their decoder chunks stay aligned) and gives it back when its stop rule fires, so concurrent callers of
    // the single-utterance API share the GPU and nobody waits for the longest utterance of a batch.
    void generate_one(const Utterance& utt, UtteranceResult& out, bool want_latents);
    // Any number of utterances through the same table (queue with slot refill); returns when all are done.
    void generate_queue(const std::vector<Utterance>& utts, std::vector<UtteranceResult>& out, bool want_latents);
    struct SchedulerStats {
        long batches = 0;       // activations of the slot table (idle -> busy)
        long utterances = 0;    // utterances served
        long largest = 0;       // most slots occupied at once
        long frame_steps = 0;   // frame steps launched
        long slot_steps = 0;    // sum of the launch width over those steps
        long frames = 0;        // frames delivered to callers
        long decode_calls = 0;  // decoder chunk calls
        long admissions = 0;    // prefill passes
    };
    SchedulerStats scheduler_stats();
    // utterances handed to this engine's slot table and not yet returned (what a dispatcher over several engines balances)
    int pending() const { return pending_.load(); }
    // The same in three parts, so a benchmark can time the device-resident middle on its own:
    //   stage (host -> HBM), run_staged (prefill + frame loop + decode, audio left in HBM), fetch_staged (HBM -> host).
    // Hold call_mutex() across the three when other threads may use the engine.
    // The Job belongs to the engine and stays valid until the next stage() call.
    Job* stage(const std::vector<Utterance>& utts);
    void run_staged(Job& job, std::vector<int>& frames, std::vector<std::vector<float>>* eos_log);
    void fetch_staged(Job& job, const std::vector<int>& frames, std::vector<UtteranceResult>& out, bool want_latents);
    std::recursive_mutex& call_mutex() { return call_mu_; }
    // Streaming, single utterance: /root/reference/src/pocket_tts.cpp:686-979.  Returns total samples.
    // cancel_flag (may be null) is polled once per frame when poll_cancel is set, and always suppresses the
    // tail flush when raised.
    // Rides in the slot table beside other callers' utterances; the callback runs on the calling thread.
    // POCKET_TTS_INLINE_STREAMING=1: the whole loop on the calling thread with a batch of its own (generate_streaming_inline).
    int generate_streaming(const Utterance& utt, int chunk_frames, const ChunkCallback& cb,
                           const std::atomic<bool>* cancel_flag, bool poll_cancel, const std::function<void(int)>& progress,
                           UtteranceResult* capture);
    int generate_streaming_inline(const Utterance& utt, int chunk_frames, const ChunkCallback& cb,
                                  const std::atomic<bool>* cancel_flag, bool poll_cancel, const std::function<void(int)>& progress,
                                  UtteranceResult* capture);
    // Voice encoder: audio f32[S] at the model rate -> [frames][d_model]  (src/pocket_tts.cpp:261-291)
    std::vector<float> encode_voice(const float* audio, int S);
    void encode_clips(const float* const* clips, int B, int S, std::vector<std::vector<float>>& out);
    void encode_enqueue(const float* const* clips, int B, int S, int slot);
    void encode_collect(int slot, std::vector<std::vector<float>>& out);
    void encode_voice_batch(const std::vector<const float*>& clips, const std::vector<int>& lens,
                            std::vector<std::vector<float>>& out);

    // ---- graph-level entry points (one per reference Run call site), host buffers in and out ----
    void op_text_conditioner(const int64_t* ids, int T, float* out);
    Batch* batch_new(int B);
    void batch_free(Batch* b);
    void batch_reset(Batch* b);
    void op_lm_prefill(Batch* b, int slot, const float* emb, int rows);
    void op_lm_step(Batch* b, const float* seq /*[B][latent]*/, float* cond /*[B][d_model]*/, float* eos /*[B]*/);
    void op_flow_euler(const float* cond, const float* x0, int B, float* x_out);
    // The same on device buffers with the scratch in a static device array: no allocation of any kind (noalloc.cpp).
    void op_flow_euler_static(const float* cond_dev, const float* x0_dev, int B, float* x_out_dev);
    static size_t static_scratch_capacity();
    void op_mimi_decode(Batch* b, const float* latents /*[B][n][latent]*/, int n, float* audio /*[B][n*frame]*/);
    // the int8 product kernel alone; scheme: 0 per row and block of 64, 1 dynamic uint8 with zero point
    void q8_linear(const float* A, int M, int K, const float* W, int N, float* out, const signed char* q = nullptr, const float* qscale = nullptr, int scheme = 0);
    void bf16_linear(const float* A, int M, int K, const float* W, int N, const float* bias, const float* colscale, const float* R, int act,
                     int flags, float* out);   // the bf16 GEMM dispatch alone

    // timing hooks for the benchmark: HIP events recorded on the engine stream around named phases
    struct PhaseTimes { double lm_ms = 0, flow_ms = 0, dec_ms = 0, head_ms = 0; long lm_calls = 0, flow_calls = 0, dec_calls = 0, head_calls = 0; };
    void enable_phase_timing(bool on);
    PhaseTimes phase_times();

    std::mutex& mutex() { return mu_; }

private:
    friend class Batch;
    void transformer(const std::vector<TfLayer>& layers, void* kc, void* vc, int kv16, long long layer_stride, int cap,
                     const int* pos_dev, int slot0, int nb, int rpb, float* X, int d, int heads, int ffn, float eps,
                     int window, const float* cos_tab, const float* sin_tab, int tab_len, const TfScratch& scratch,
                     float* last_out, long long last_bs, int last_ld, hipStream_t st, bool split_a = false, bool bf16_rows = false);
    void stage_part(Batch& b, const Utterance* utts, int n);
    // One frame of the autoregressive loop for a staged batch, free of host-side frame state (replayable as a graph):
    // LM step -> EOS logits to the host -> flow head from the start points of frame *step -> latent filed under *step.
    void frame_body(Batch& b, int parity);
    hipGraphExec_t* ensure_frame_graph(Batch& b);
    void lm_bookkeeping(Batch& b);
    void lm_rows_skinny(Batch& b, int slot0, int nb, int rpb, float* X);
    void lm_rows(Batch& b, int slot0, int nb, int rpb, float* X, const int* adv_dev = nullptr, const int* adv_host = nullptr);
    // prepped: seq_dev already holds the BOS embedding where the reference passes NaN (no preparation launch)
    void lm_step_dev(Batch& b, const float* seq_dev, float* cond_dev, float* eos_dev, bool prepped = false);
    void flow_dev(Batch& b, const float* cond_dev, const float* x0_dev, long long x0_stride, float* x_out_dev, int B,
                  const int* x0_step = nullptr, long long x0_step_stride = 0, hipEvent_t probe_a = nullptr, hipEvent_t probe_b = nullptr,
                  bool frame_mode = false);   // frame_mode: EOS publication, frame counter and latent filing ride in the phase's launches
    struct FlowScratch { float *cemb, *ymod, *mods, *ws; unsigned* ctl; };
    void flow_core(const FlowScratch& fs, hipStream_t st, const float* cond_dev, const float* x0_dev, long long x0_stride, float* x_out_dev,
                   int B, const int* x0_step, long long x0_step_stride, hipEvent_t probe_a, hipEvent_t probe_b, Batch* frame);
    void flow_scratch_layout(int rows, size_t off[6]) const;
    // After the stream has drained: throws if a hand-off wait of the persistent flow head gave up during the calls since the last check.
    void check_flow_ring(const unsigned* ctl, int rows);
    void ring_failed();
    // slot0 / nb: the contiguous range of slots the chunk is decoded for (nb < 0: all).  ring_start >= 0: the latent rows come from
    // the batch's latent rings starting at that ring index instead of lat_dev.  due_mask: slots (bit = slot index) whose streaming state
    // the chunk advances; the others are computed along and left as they were.
    void mimi_decode_dev(Batch& b, const float* lat_dev, long long lat_bs, int n, float* audio_dev, long long audio_bs, hipStream_t st,
                         int slot0 = 0, int nb = -1, int ring_start = -1, unsigned long long due_mask = ~0ull);
    void prefill_utterance(Batch& b, int slot, const Utterance& u);
    void fill_noise(std::vector<std::mt19937>& rngs, const std::vector<Utterance>& utts, int step, float* host_x0);
    Batch* acquire_batch(int B, int lane = 0);
    void phase_begin(int which, hipStream_t st = nullptr);
    void phase_end(int which, hipStream_t st = nullptr);

    // Device pointers resolved once at construction (name lookups stay off the per-frame path).
    struct Handles {
        std::vector<TfLayer> lm, dec, enc;
        DevWeight lm_in, flow_cond, flow_in, flow_out, dec_in, dec_conv0, dec_up[3], dec_c1[3], dec_c2[3];
        DevWeight enc_c1[3], enc_c2[3], enc_down[3], enc_final, enc_ds, enc_spk;
        const float *bos, *out_g, *out_b, *eos_w, *eos_b, *flow_cond_b, *flow_in_b, *flow_out_b;
        const float *emb_std, *emb_mean, *dec_up_w, *dec_conv0_b, *dec_up_b[3], *dec_c1_b[3], *dec_c2_b[3], *dec_final_w, *dec_final_b;
        const float *enc_conv0_w, *enc_conv0_b, *enc_c1_b[3], *enc_c2_b[3], *enc_down_b[3], *enc_final_b, *text_embed;
    } hd_;

    EngineOptions opt_;
    std::unique_ptr<Model> model_;
    hipStream_t stream_ = nullptr;
    hipStream_t prefill_stream_ = nullptr;      // prefill passes of the slot table while other slots keep stepping (the process-wide side stream of the device)
    hipStream_t dec_stream_ = nullptr;          // the Mimi decoder runs here, beside the frame loop
    hipEvent_t latents_ready_ = nullptr, dec_done_ = nullptr;
    std::mutex mu_;
    std::recursive_mutex call_mu_;
    std::mt19937 rng_;
    std::mutex rng_mu_;        // rng_ is drawn from by queued calls, staged batches and in-line streams alike
    unsigned char* static_scratch_ = nullptr;   // this device's copy of the static scratch array (noalloc.cpp)
    void init_static_scratch();
    DevBuf flow_te_;  // [lsd_steps][C] averaged time embeddings of the fixed (s_j, t_j) grid
    struct EncSlot {             // buffers of one voice-encoder pass, kept between calls
        DevBuf arena;            // every device buffer of the pass
        float* pin_in = nullptr; size_t pin_in_bytes = 0;      // pinned staging of the clips
        float* pin_out = nullptr; size_t pin_out_bytes = 0;    // pinned landing area of the embeddings
        hipEvent_t done = nullptr;
        int clips = 0; size_t row_floats = 0;
    };
    EncSlot enc_slot_[2];
    std::mutex enc_mu_;
    std::atomic<int> pending_{0};
    bool owns_ring_ = false;   // this engine holds its GPU's one persistent flow-head slot (see flow_ring.hip)
    bool ring_gave_up_ = false;   // ... and stopped using it after a failed hand-off (the slot stays taken until the engine goes)
    std::map<std::pair<int, int>, std::unique_ptr<Batch>> batch_pool_;   // (rows, lane)
    std::list<std::pair<int, int>> batch_lru_;                           // keys of batch_pool_, most recently used first
    static constexpr size_t kMaxPooled = 8;                              // batch sizes kept warm per engine
    std::vector<hipStream_t> lane_streams_;     // lane 0 = stream_
    std::vector<hipEvent_t> lane_events_;
    bool lane_shared_decoder_ = false;          // parts > 1: decoder chunks on dec_stream_ instead of the lane's own stream
    int lanes_ = 1;
    bool dec_fused_[3] = {false, false, false};   // decoder stages whose residual unit runs as one fused kernel
    bool use_graph_ = true;                     // replay each frame as one hipGraph (POCKET_TTS_NO_GRAPH=1: eager launches)
    Job job_;
    // slot-table scheduler (scheduler.cpp)
public:
    struct Request;
    static constexpr int kSlots = 64;
private:
    struct Table;
    void validate_request(const Utterance& utt) const;
    // Waiting for the per-frame counter the GPU writes to mapped host memory (the step whose EOS logits have arrived).
    // The thread polls (one core per engine while a frame loop runs).  POCKET_TTS_WAIT_SLEEP_PCT=p lets it sleep through the
    // first p % of the delay it has seen on average before polling: measured at p = 60 on the 64 x 125 bf16 workload, the
    // process drops from 100 % to 37 % of a core and the frame from 0.85 ms to 1.10 ms (the wake-up is late), so it is off
    // by default.  Throws if the stream drains without the counter reaching `target`.
    // A wall-clock limit applies (100 x the running average, at least POCKET_TTS_WAIT_LIMIT_MS, default 2000 ms): a frame that
    // never publishes fails the call with a message instead of hanging it; a HIP error on the stream is thrown as it is.
    struct FlagWait { double avg_us = 0; };
    void wait_flag(const int* flag, int target, hipStream_t st, FlagWait& w);
    void submit(Request& r);
    void submit_all(Request* const* reqs, size_t n);
    void worker_main();
    void stop_worker();
    void drop_table();
    void run_table();
    Table* table_ = nullptr;   // owned; created by the first activation, freed by drop_table()
    std::thread worker_;
    bool worker_started_ = false, worker_stop_ = false;
    std::mutex q_mu_;
    std::condition_variable q_cv_;
    std::deque<Request*> queue_;
    SchedulerStats sched_stats_;
    DevBuf bos_row_;   // the BOS embedding as a latent row (what the reference's NaN marker stands for)
    std::mutex ev_mu_;
    bool timing_ = false;
    hipEvent_t ev_[2] = {nullptr, nullptr};
    PhaseTimes times_;
    struct PhaseEvent { int which = 0; hipStream_t st = nullptr; hipEvent_t a = nullptr, b = nullptr; bool closed = false; };
    std::vector<PhaseEvent> pending_events_;
};

// Recurrent state and scratch of B lock-step utterances on one GPU.
class Batch {
public:
    // lm_only: a staging batch for prefill passes (LM state and transformer scratch only)
    Batch(Engine& e, int B, bool lm_only = false);
    void reset(hipStream_t s);

    int B;            // capacity: slots, and the slot stride of every state buffer
    int Bw;           // rows the frame step and the flow phase are launched over (<= B; the slot table narrows it to its occupied prefix)
    bool lm_only = false;
    // Slot-table mode (scheduler.cpp): latents and start points are rings of step_ring frames indexed by the table's global
    // step, the EOS rows in mapped host memory a ring of eos_ring steps.  0: indexed by the frame number of one run.
    int step_ring = 0, eos_ring = 0;
    int lat_frames = 0;   // frames per slot of `latents` (max_frames rounded up to whole decoder chunks, plus two)
    hipStream_t st = nullptr;       // frame-loop stream of this batch
    hipStream_t dec_st = nullptr;   // decoder stream (may equal st)
    hipEvent_t ready = nullptr;     // latents of a chunk are complete (st -> dec_st)
    int lm_cap;       // KV capacity per slot (positions)
    int dec_cap;      // ring capacity of the decoder transformer cache
    int max_rows;     // scratch rows for transformer passes
    // LM state
    int kv16 = 0;           // 1: K/V caches hold bf16 (bf16 and int8 engines), 0: f32
    DevBuf lm_k, lm_v;      // [layers][B][lm_cap][d_model]
    DevBuf lm_pos;          // int[B]
    std::vector<int> lm_pos_host;
    // decoder state
    DevBuf dec_k, dec_v;    // [layers][B][dec_cap][mimi_dim]
    DevBuf dec_pos;         // int[B]
    std::vector<long long> dec_pos_host;   // per slot
    // decoder activation buffers (history rows in front), see engine.cpp for the layout table
    DevBuf zbuf, qbuf, ybuf, e0buf, ubuf[3], uebuf[3], h1buf[3], ebuf[3];
    // transformer scratch (grown on demand by ensure_rows)
    DevBuf x, h, qkv, q, att, ff;
    size_t scratch_rows_d = 0, scratch_rows_f = 0;  // capacity in floats of x/h/q/att and of ff
    void ensure_rows(size_t rows, int d, int ffn);
    void drop_graphs();   // destroys the captured frame graphs (they hold buffer addresses)
    DevBuf st_nv, st_nt, st_len;  // per-slot int arrays of the staged batch
    // LM I/O
    DevBuf seq_in, cond, eos, x0, xcur, latents, ids;
    TfScratch scratch() const { return TfScratch{h.f(), qkv.f(), q.f(), att.f(), ff.f()}; }
    DevBuf dx, dh, dqkv, dq, datt, dff;  // decoder-transformer scratch (separate: the decoder overlaps the frame loop)
    TfScratch dec_scratch() const { return TfScratch{dh.f(), dqkv.f(), dq.f(), datt.f(), dff.f()}; }
    // flow scratch
    DevBuf cemb, ymod, mods, flow_ws, flow_ctl;
    DevBuf partial;   // split-K slabs of the decode step [splits][B][d_model] f32
    DevBuf hq, hd, attq, attd, ffq, ffd;   // int8 engine: the decode step's matrix inputs as int8 rows + block scales [K / 64][Mq]
    int q_rows = 0;                        // Mq = B rounded up to 16
    DevBuf mm_h, mm_att, mm_ff;            // int8 engine, dynamic uint8 scheme: (min, max) of the rows of h, of the attention output per head, of the FFN hidden rows per 64 columns
    DevBuf pre_partial;   // K-slice slabs [splits][rows][d_model] of a prefill pass on the skinny kernels
    // audio out
    DevBuf audio;
    // staged inputs / outputs of a whole batch (Engine::stage / run_staged)
    DevBuf voice_stage, ids_stage, noise_all, audio_all;
    std::vector<int> st_voice_frames, st_ntok;
    int st_vmax = 1;
    bool st_has_noise = false;
    // pinned host staging
    float* h_eos = nullptr;
    float* h_x0 = nullptr;
    // frame loop state that lives on the device / in mapped host memory (see Engine::frame_body)
    DevBuf step_dev;                 // int: index of the frame being generated
    float* h_eos_all = nullptr;      // [max_frames][B] EOS logits, written by the GPU
    int* h_flag = nullptr;           // number of frames whose EOS logits have arrived
    const float* x0_src = nullptr;   // start points of the staged batch and their strides
    long long x0_stride = 0, x0_step_stride = 0;
    // captured frame graphs by launch width, one per frame parity (each has its own timing events)
    struct FrameGraphs { hipGraphExec_t exec[2] = {nullptr, nullptr}; const float* x0 = nullptr; long long x0_stride = -1; bool timed = false; int ring = 0; };
    std::map<int, FrameGraphs> frame_graphs;
    hipEvent_t tev[2][8] = {};       // lm begin/end, flow begin, probe begin/end, flow end, and an empty back-to-back pair
    ~Batch();
};

}  // namespace ptts
