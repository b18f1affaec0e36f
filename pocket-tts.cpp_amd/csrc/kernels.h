// Host-side launch interface of the gfx950 kernels.  All activations are float32, channels-last
// ([rows][channels]); "batched rows" are addressed as
//     row m  ->  base + (m / rpb) * batch_stride + (m % rpb) * ld
// so that one launch covers B utterances whose buffers carry per-utterance history rows in front.
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

namespace ptts {

enum Act : int { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2, ACT_ELU = 3 };
enum WType : int { W_F32 = 0, W_BF16 = 1, W_I8 = 2 };

// C[m][n] = epilogue( sum_k act_a(A[m][k]) * W[n][k] )
//   v = acc + bias[n % bias_mod]; v = act_pre(v); v *= G[m][n]; v *= colscale[n] * alpha; v += R[m][n];
//   rows with (m % rpb) >= row_len[m / rpb] store 0;  C = c_act(v);  C2 = c2_act(v).
// A rows may overlap (lda < K): that is how a k-tap convolution over channels-last rows is expressed.
struct GemmArgs {
    const float* A = nullptr; long long a_bs = 0; int lda = 0;
    const void* W = nullptr; int wtype = W_F32;
    float* C = nullptr; long long c_bs = 0; int ldc = 0; int c_act = ACT_NONE;
    float* C2 = nullptr; long long c2_bs = 0; int ldc2 = 0; int c2_act = ACT_NONE;
    const float* bias = nullptr; int bias_mod = 0;
    const float* colscale = nullptr; float alpha = 1.0f;
    const float* R = nullptr; long long r_bs = 0; int ldr = 0;
    const float* G = nullptr; long long g_bs = 0; int ldg = 0;
    const int* row_len = nullptr;
    int rpb = 1 << 30;
    int M = 0, N = 0, K = 0;
    int a_act = ACT_NONE, act_pre = ACT_NONE;
    int a_bf16 = 0, c_bf16 = 0;   // A rows / C rows are bf16 in memory (bf16 weights only; strides count elements)
    int split_a = 0;  // bf16 weights only: stage A as hi + lo bf16 terms (two MFMAs per tile) instead of rounding it once
    // bf16 weights only, with split_a: the second bf16 term of the weights, W2 = bf16(W - bf16(W)).  The product then carries
    // both operands to ~16 mantissa bits (A_hi W_hi + A_lo W_hi + A_hi W_lo, f32 accumulation): f32-grade results from three
    // bf16 MFMAs, at a third of the bf16 rate instead of the f32 MFMA's sixteenth.
    const void* W2 = nullptr;
};
void launch_gemm(const GemmArgs& a, hipStream_t s);

// Y[m][:] = (X[m][:] - mean) * rstd * g + b   (g, b may be null)
struct LayerNormArgs {
    const float* X = nullptr; long long x_bs = 0; int ldx = 0;
    float* Y = nullptr; long long y_bs = 0; int ldy = 0;   // y_bf16: Y holds bf16 rows (strides in bf16 elements)
    int y_bf16 = 0;
    int rpb = 1 << 30, M = 0, D = 0;
    const float* g = nullptr; const float* b = nullptr; float eps = 1e-5f;
};
void launch_layernorm(const LayerNormArgs& a, hipStream_t s);

// Split qkv[M][3d], rotate q and k by the row's absolute position (interleaved pairs), write q_out[M][d]
// and append k, v to the per-slot cache at index pos % cap.  Row m belongs to slot slot0 + m / rpb at
// position pos[m / rpb] + m % rpb.  cos_tab/sin_tab: [tab_len][dh/2].
struct RopeAppendArgs {
    const float* qkv = nullptr; float* q_out = nullptr;
    void* kcache = nullptr; void* vcache = nullptr;  // [slots][cap][d] of one layer, f32 or (kv16) bf16
    int kv16 = 0;
    const int* pos = nullptr; int slot0 = 0; int rpb = 1;
    int M = 0, d = 0, heads = 0, cap = 0;
    const float* cos_tab = nullptr; const float* sin_tab = nullptr; int tab_len = 0;
};
void launch_rope_append(const RopeAppendArgs& a, hipStream_t s);

// out[m][h*64..] = softmax(q . K / 8) V over keys max(0, p-window+1)..p of the row's slot (head dim 64).
struct AttentionArgs {
    const float* q = nullptr; float* out = nullptr;
    const void* kcache = nullptr; const void* vcache = nullptr;   // f32 or (kv16) bf16
    int kv16 = 0;
    const int* pos = nullptr; int slot0 = 0; int rpb = 1;
    int M = 0, d = 0, heads = 0, cap = 0, window = 0;  // window <= 0: all earlier keys
};
void launch_attention(const AttentionArgs& a, hipStream_t s);
void launch_attention_tiled(const AttentionArgs& a, hipStream_t s);   // rpb rows per slot, M % rpb == 0

// pos[i] += delta for i < n
// mask (with slot0 = index of pos[0] in it): only slots whose bit is set advance; ~0: all
void launch_advance_pos(int* pos, int n, int delta, hipStream_t s, unsigned long long mask = ~0ull, int slot0 = 0);
// pos[i] += delta[i] for i < n
void launch_advance_pos_vec(int* pos, const int* delta, int n, hipStream_t s);

// Prefill rows of a whole batch: out[b][t][:] = voice[b][t][:] for t < nv[b], the embedding of
// ids[b][t - nv[b]] for the next nt[b] rows, zero for the padding up to rpb rows.
void launch_assemble_prefill(const float* voice, long long voice_bs, const long long* ids, int ids_stride, const int* nv, const int* nt,
                             const float* table, int vocab_rows, int D, int rpb, int B, float* out, hipStream_t s);

// out[t][:] = table[clamp(ids[t])][:]
void launch_embed_rows(const long long* ids, int T, const float* table, int vocab_rows, int D, float* out, hipStream_t s);

// mode 0: y = isnan(x) ? bos[c] : x        (BOS substitution in front of the LM input projection)
// mode 1: y = x * std[c] + mean[c]          (latent de-normalisation in front of the decoder)
void launch_prep_latent(const float* x, float* y, int rows, int L, int mode, const float* p0, const float* p1, hipStream_t s);

// Decoder input of a chunk taken from per-slot latent rings: z[b][i][:] = lat[(slot0 + b)][(start + i) % ring][:] * std + mean
// for b < nb, i < n  (lat: [slots][ring][L]).
void launch_gather_latents(const float* lat, int ring, int slot0, int nb, int start, int n, int L, const float* stdv, const float* mean, float* z,
                           hipStream_t s);
// *p = v
void launch_poke_int(int* p, int v, hipStream_t s);
// pos[i] = 0 for every i < 64 whose bit is set in mask
void launch_reset_pos_mask(int* pos, unsigned long long mask, hipStream_t s);

// Admission of up to 64 prefilled utterances into slots of a slot table, one launch: the K/V rows [layer][len] of staging
// slot i go to table slot slot[i] (caches laid out [layer][slot][cap][row_bytes]), the slot's position becomes len[i], its
// current latent row the BOS row, and its start-point ring rows (g0 + f) % ring, f < frames, are taken from
// noise_stage[i][frames][L] (has_noise[i]) or zeroed.
struct AdmitArgs {
    int k = 0, layers = 0, row_bytes = 0, cap = 0, dst_slots = 0, src_slots = 0;
    char *k_dst = nullptr, *v_dst = nullptr;
    const char *k_src = nullptr, *v_src = nullptr;
    int* lm_pos = nullptr;
    float* xcur = nullptr;
    const float* bos = nullptr;
    int L = 0, ring = 0, frames = 0, g0 = 0;
    float* noise_ring = nullptr;
    const float* noise_stage = nullptr;
    unsigned char slot[64] = {}, src[64] = {}, has_noise[64] = {};   // destination slot, row of the staging batch, start points given
    int len[64] = {};
};
void launch_admit(const AdmitArgs& a, hipStream_t s);

// Streaming-decoder state of the given slots back to "no history": for each of nbuf buffers laid out [slot][per_slot bytes],
// the first hist bytes of the slot are zeroed, and dec_pos[slot] = 0.
struct DecResetArgs {
    int k = 0, nbuf = 0;
    char* buf[16] = {};
    long long per_slot[16] = {};
    int hist[16] = {};
    int* dec_pos = nullptr;
    unsigned char slot[64] = {};
};
void launch_dec_reset(const DecResetArgs& a, hipStream_t s);

// eos[m] = dot(X[m], w) + b
void launch_rowdot(const float* X, int ldx, int M, int D, const float* w, const float* b, float* out, hipStream_t s);

// Depthwise transposed conv, kernel 2U stride U: q[B][1 + n][M] (row 0 = previous frame) -> y rows (B*n*U)[M] flat.
void launch_upsample_dw(const float* q, long long q_bs, int n, int M, int U, const float* w, float* y, int B, hipStream_t s);

// 1-output-channel conv over a contiguous window: out[b][t] = bias + dot(X[b][t .. t+k-1][:], w[k*C])
void launch_conv_to_mono(const void* X, int x_bf16, long long x_bs, int T, int KC, int C, const float* w, const float* bias,
                         float* out, long long out_bs, int B, hipStream_t s);

// 1-input-channel causal conv: X[b][t][co] = bias[co] + sum_tap w[co][tap] * audio[b][t + tap - (k-1)]
// writes raw and ELU'd copies; rows t >= len[b] store 0.
void launch_conv_from_mono(const float* audio, long long a_bs, const int* len, int T, int k, int Co, const float* w,
                           const float* bias, float* raw, long long raw_bs, float* elu, long long elu_bs, int B,
                           hipStream_t s);

// First stage of the voice encoder in one launch (encoder_fused.hip): out[b][t][64] = ELU(skip + conv1(ELU(conv3(ELU(x)))))
// with x[b][t][64] = conv_k(audio[b]) (one input channel, causal), for t < T; rows the unit pads with are zero.
struct EncStage0Args {
    const float* audio = nullptr; long long a_bs = 0; int T = 0, B = 0;
    int k = 0, c0 = 0, c1 = 0, rk = 0;                         // first convolution's taps; channels 64 / 32; unit kernel 3
    const float* w0 = nullptr; const float* b0 = nullptr;      // [c0][k], [c0]
    const void* w1_hi = nullptr; const void* w1_lo = nullptr; const float* b1 = nullptr;   // [c1][rk * c0] as two bf16 terms
    const void* w2_hi = nullptr; const void* w2_lo = nullptr; const float* b2 = nullptr;   // [c0][c1]
    float* out = nullptr; long long out_bs = 0;
};
bool enc_stage0_supported(const EncStage0Args& a);
void launch_enc_stage0(const EncStage0Args& a, hipStream_t s);

// Fused residual unit of a SEANet decoder stage (decoder_fused.hip):
//   E[b][e_off + t][:] = ELU(y_t + W2 . ELU(W1 * ELU(y_{t-2..t}) + b1) + b2),  y_t = Y[b][t + 2][:]
// Y holds the two previous rows of the stream in front of the chunk's T rows; bf16 weights, bf16 Y and E rows,
// 64 or 128 channels.
struct ResblockArgs {
    const void* Y = nullptr; long long y_bs = 0;           // bf16 rows; strides count elements
    void* E = nullptr; long long e_bs = 0; int e_off = 0;   // bf16 rows
    const void* W1 = nullptr; const float* b1 = nullptr;   // [co/2][3*co] (K index = tap*co + ci), bf16
    const void* W2 = nullptr; const float* b2 = nullptr;   // [co][co/2], bf16
    int T = 0, B = 0, co = 0;
};
bool resblock_supported(int co, int taps, int wtype);
void launch_resblock(const ResblockArgs& a, hipStream_t s);

// Carry history: for each b, rows [T, T+H) of buf[b] move to rows [0, H)  (row length C floats).
void launch_shift_history(float* buf, long long bs, int T, int H, int C, int B, hipStream_t s, unsigned long long mask = ~0ull, int slot0 = 0);

// ---- LSD flow head --------------------------------------------------------------------------------
// All `steps` Euler iterations of the flow MLP for every row:
//   x <- x + dt * v(c, s_j, t_j, x),  j = 0..steps-1
// as a chain of per-block launches (flow_head.hip).  mods[j][b][:] holds the precomputed adaLN vectors of
// Euler step j (they depend on c and j only): per block (shift, scale, gate) x C, then final (shift, scale) x C.
struct FlowHostTabs {   // host arrays of per-block device pointers
    std::vector<const void*> w_fc1, w_fc2;           // pack_flow_weight (or pack_q8_weight) layout, [C][C]
    std::vector<const float*> b_fc1, b_fc2, ln_g, ln_b;
    std::vector<const float*> s_fc1, s_fc2;          // int8 blocks: per-output-column weight scales
    // Matrices of the persistent form as bf16 fragment-major planes.  bf16 engine: the matrices above, no second plane.
    // float engine: every matrix as two planes, W = hi + lo (lo = bf16(W - hi)), which the kernel multiplies with activations
    // split the same way on three bf16 MFMAs per tile: float-grade products at bf16 MFMA speed.
    std::vector<const void*> r_fc1, r_fc2, r_fc1_lo, r_fc2_lo;
    const void *r_in = nullptr, *r_out = nullptr, *r_in_lo = nullptr, *r_out_lo = nullptr;
};
struct FlowHeadArgs {
    const float* x0 = nullptr;    // start points, row b at x0 + b * x0_stride (+ *x0_step * x0_step_stride if x0_step is set)
    long long x0_stride = 0;
    const int* x0_step = nullptr; long long x0_step_stride = 0;
    int x0_step_bias = 0;         // added to *x0_step wherever it is used
    int step_ring = 0;            // > 0: the frame index (*x0_step + bias) is taken modulo this (start points and latents kept in rings)
    float* latents = nullptr; int max_frames = 0;   // optional (needs x0_step): x_out is also filed as latents[b][*x0_step + bias][:]
    hipEvent_t probe_a = nullptr, probe_b = nullptr;   // optional: recorded around the whole chain of steps * (blocks + 1) launches
    float* x_out = nullptr;       // [B][L]; also carries x between Euler steps (must not alias x0)
    const float* mods = nullptr;  // [steps][B][mod_width]
    int B = 0, L = 0, C = 0, blocks = 0, steps = 0, mod_width = 0;
    const void* w_in = nullptr; const float* b_in = nullptr;       // [C][L], pack_flow_weight layout
    const void* w_out = nullptr; const float* b_out = nullptr;     // [L][C], pack_flow_weight layout
    const FlowHostTabs* host_tabs = nullptr;
    float* workspace = nullptr;   // flow_head_workspace_bytes(B, C, wtype) bytes
    int wtype = W_F32;            // of w_in / w_out
    int block_wtype = -1;         // of the residual blocks' fc1 / fc2 (-1: same as wtype; W_I8: int8 blocks)
    float eps = 1e-6f;
    long long* cycle_probe = nullptr;   // optional device [8]: cycle counter at the phase boundaries of one mid-chain block launch
    // Control words of the persistent form (flow_ring.hip): flow_ring_ctl_bytes(B, blocks) bytes that an earlier kernel on the
    // same stream has zeroed for this call.  Null: the launch chain is used.
    unsigned* ring_ctl = nullptr;
    bool ring_split = false;     // float engine: host_tabs carries two-term bf16 planes of every matrix for the persistent form
    int ring_wg_budget = 224;    // workgroups the persistent form may occupy (all must be resident at once; other rings in flight share the chip)
};
void launch_flow_head(const FlowHeadArgs& a, hipStream_t s);
int flow_head_launch_count(const FlowHeadArgs& a);   // kernel launches one call makes (only C, B, steps, blocks, wtype, block_wtype are read)
size_t flow_head_workspace_bytes(int B, int C, int wtype, int blocks);
// Persistent form: one launch per call, every matrix held in registers (bf16 weights, flow_dim 512, latent 32).
constexpr int kFlowRingMaxBlocks = 8;
bool flow_ring_supported(const FlowHeadArgs& a);
void launch_flow_ring(const FlowHeadArgs& a, hipStream_t s);
size_t flow_ring_workspace_bytes(int B, int blocks);
size_t flow_ring_ctl_bytes(int B, int blocks);
size_t flow_ring_err_word(int B, int blocks);
size_t flow_ring_counter_word();
void flow_ring_prepare();   // once per device, outside stream capture
void flow_head_prepare(int C, int wtype);   // once per device, outside stream capture
// Host-side reorder of a flow-head weight W[N][K] into the fragment-major layout the kernel streams.
std::vector<unsigned char> pack_flow_weight(const float* W, int N, int K, int wtype);
// int8 form of the same layout (k-steps of 64, 16 values per lane): per output row n, scale[n] = max|W[n][:]| / 127
// and q = rint(W * 127 / max) in [-127, 127].  K must be a multiple of 64.
std::vector<unsigned char> pack_q8_weight(const float* W, int N, int K, std::vector<float>& scale);
// The layout step alone, for int8 values that come with their own scales (a pack's `<name>.q8` form).
std::vector<unsigned char> pack_q8_layout(const signed char* q, int N, int K);
// sums[s][n] = sum over K-slice s of q[n][k], for SkinnyArgs::wsum
std::vector<int> q8_row_sums(const signed char* q, int N, int K, int splits);
// the int8 values pack_q8_weight derives from a float matrix (before the layout step)
std::vector<signed char> quantize_q8_rows(const float* W, int N, int K, std::vector<float>& scale);

// ---- batched decode step (decode_kernels.hip) -----------------------------------------------------
// C[M][N] = A[M][K] * W^T with f32 A and W fragment-major (pack_flow_weight layout, f32 or bf16).
// splits == 1 and P == null: C = act(acc + bias).  P != null: K is cut into `splits` slices and slice s
// writes the raw f32 slab P[s][M][N].
struct SkinnyArgs {
    const float* A = nullptr; int lda = 0;
    const void* Wp = nullptr; int wtype = W_F32;
    const void* Wp2 = nullptr;       // W_BF16 only: second plane of a two-term matrix, bf16(W - bf16(W)), same layout; a third MFMA per fragment
    int M = 0, N = 0, K = 0, splits = 1;
    const float* bias = nullptr; int act = ACT_NONE;
    float* C = nullptr; int ldc = 0;
    float* P = nullptr;
    // C form only (f32 / bf16 weights): C = act(acc + bias) * colscale + R
    const float* colscale = nullptr; const float* R = nullptr; int ldr = 0;
    const float* wscale = nullptr;   // W_I8: per-output-column weight scale
    // W_I8, activation scheme.  0: per row and block of 64 k, symmetric int8 (s = max|a|, q = rint(a * 127 / s)).
    // 1: ONNX DynamicQuantizeLinear per row: uint8 with a zero point over [min(0, min a), max(0, max a)]; needs
    // wsum[splits][N] = per K-slice sums of each weight row (q8_row_sums).
    int q8_act = 0; const int* wsum = nullptr;
    // scheme 1, optional: the rows' value ranges as their producers left them, a_mm[mm_parts][M][2] = (min, max) of a part of
    // the row (the product reduces the parts and includes 0); without it the product scans its rows over the full K itself.
    // c_mm (C form, N % 64 == 0): the ranges of the result rows after the activation, one part per 64 columns, [N / 64][M][2]
    const float* a_mm = nullptr; int mm_parts = 0; float* c_mm = nullptr;
    int a_frag = 0;                  // A holds two-term bf16 fragments (launch_flow_mod_input's frag form); only where skinny_wide_rows_applies()
    int mtiles = 0;                  // 16-row tiles per workgroup: 1, 2 or 4 (0 = 4)
    int kchunk = 0;                  // K elements staged per step: 128, 256 or 512 (bf16, mtiles <= 2); 0 = 128
    // W_I8 only, activations quantised by the kernel that produced them (per row and block of 64 k: q = rint(a * 127 / max|a|),
    // d = max|a| / 127): Aq[M][ldaq] int8 and Ad[K / 64][Mq] f32 replace A; Mq = M rounded up to 16
    const signed char* Aq = nullptr; int ldaq = 0; const float* Ad = nullptr; int Mq = 0;
    // ... and the same form of the result instead of C (needs N % 64 == 0): Cq[M][N], Cd[N / 64][Mq]
    signed char* Cq = nullptr; float* Cd = nullptr;
};
void launch_skinny_gemm(const SkinnyArgs& a, hipStream_t s);
// Whether launch_skinny_gemm runs this product on the rows-stationary kernel (which alone understands a_frag).
bool skinny_wide_rows_applies(const SkinnyArgs& a);

// Per row m: X[m] += colscale * (sum_s P[s][m] + pbias)  (skipped when P is null), then
// y = LayerNorm(X[m]; g, b) -> H, optional second copy H32, optional dot_out[m] = y . dot_w + dot_b,
// optional pos[m] += pos_delta.  g == null: no normalisation outputs.
struct AddNormArgs {
    float* X = nullptr; int ldx = 0;
    const float* P = nullptr; int splits = 0;
    const float* colscale = nullptr; const float* pbias = nullptr;
    const float* g = nullptr; const float* b = nullptr; float eps = 1e-5f;
    float* H = nullptr; float* H32 = nullptr;
    signed char* Hq = nullptr; float* Hd = nullptr; int Mq = 0;   // y also (or instead of H) as int8 blocks of 64 + scales [D / 64][Mq]
    float* H_mm = nullptr;                                        // (min, max) of every y row, [M][2]
    const float* dot_w = nullptr; const float* dot_b = nullptr; float* dot_out = nullptr;
    int* pos = nullptr; int pos_delta = 0;
    int M = 0, D = 0;
    // optional prologue: X[m][:] = W_in[D][L] . seq[m][:L] (row-major W_in, f32 or bf16) instead of reading X; L % 8 == 0
    const float* seq = nullptr; const void* w_in = nullptr; int w_in_type = W_F32; int L = 0;
};
void launch_add_norm(const AddNormArgs& a, hipStream_t s);

// One new token per row: rotate q/k of qkv[M][3d] at position pos[m], append k/v to slot slot0+m of the
// cache, attend over the slot's keys (window as in AttentionArgs), write out[M][d].
struct AttnDecodeArgs {
    const float* qkv = nullptr;       // [qkv_splits][M][3d]: K-slice slabs of the QKV product, added in slice order
    int qkv_splits = 1;
    void* kcache = nullptr; void* vcache = nullptr;   // f32 or (kv16) bf16
    int kv16 = 0;
    const int* pos = nullptr; int slot0 = 0;
    int M = 0, d = 0, heads = 0, cap = 0, window = 0;
    const float* cos_tab = nullptr; const float* sin_tab = nullptr; int tab_len = 0;
    float* out = nullptr;
    signed char* out_q = nullptr; float* out_d = nullptr; int Mq = 0;   // instead of out: int8 per (row, head) + scales [heads][Mq]
    float* out_mm = nullptr;                                            // (min, max) of every (head, row) of out, [heads][M][2]
};
void launch_attn_decode(const AttnDecodeArgs& a, hipStream_t s);

// Y[j][b][:] = silu(te[j][:] + sum_s cemb[s][b][:] + bias)   (input rows of the stacked adaLN GEMM)
// Optional frame bookkeeping riding in the same launch (block 0): the EOS logits of frame *step go to h_eos_all[*step][B]
// in mapped host memory, then h_flag = *step + 1 and *step += 1 (so later kernels of the frame see the index plus one).
struct FramePublish {
    const float* eos = nullptr; int* step = nullptr; float* h_eos_all = nullptr; int* h_flag = nullptr;
    int ring = 0;     // > 0: h_eos_all is a ring of this many steps (row = *step % ring)
    int stride = 0;   // > 0: floats per step row of h_eos_all (default: B)
    // optional: a device word (the persistent flow head's error word, as the launches of earlier frames left it) copied to
    // mapped host memory in front of the flag, so that the host sees a failed hand-off one frame after it happened
    const unsigned* ring_err = nullptr; unsigned* h_ring_err = nullptr;
};
// zero / zero_words (optional): words the launch also sets to zero (the control words of a later kernel of the same frame)
// frag: Y is written as the two bf16 terms of every value in MFMA fragment order instead of f32 rows (the same bytes in
// all): 16-byte unit ((row / 16 * (C / 32) + k / 32) * 2 + term) * 64 + (k % 32 / 8) * 16 + row % 16 holds k % 8 = 0..7 of
// row steps * B + b; needs C % 32 == 0 and steps * B % 16 == 0.  SkinnyArgs::a_frag reads it.
void launch_flow_mod_input(const float* te, const float* cemb, int splits, const float* bias, float* Y, int steps, int B, int C, hipStream_t s,
                           const FramePublish* pub = nullptr, unsigned* zero = nullptr, int zero_words = 0, int frag = 0);

}  // namespace ptts
