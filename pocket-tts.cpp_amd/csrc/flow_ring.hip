// LSD flow head as ONE persistent launch per frame: all lsd_steps Euler iterations of the adaLN MLP
// (/root/reference/src/pocket_tts.cpp:414-451 called from the loop at :615-622) with every weight matrix held in
// registers for the whole launch.
//
// flow_head.hip runs the same arithmetic as a chain of steps * blocks launches and re-reads every matrix once per
// Euler step; a launch there costs ~3.3 us before its first useful instruction.  Here the chain's links are
// workgroups that stay resident and hand rows to each other through L2:
//
//   grid = replicas x blocks x G workgroups of 256 threads (one wave per SIMD, so each wave may hold up to 512
//   registers), one per CU.  Workgroup (rep, blk, g) owns, for residual block blk, the column slice g of fc1 and the
//   matching K-slice of fc2 (64 KB + 64 KB of bf16 at flow_dim 512, G = 8: 128 registers per lane; twice that with G = 4,
//   the default, or with two-term matrices), loaded once.
//   Rows travel in groups of RPG utterances; replica rep serves the row groups q = rep, rep + replicas, ...  A "visit" is one row group passing one workgroup: rebuild the residual
//   stream from the previous block's G partial slabs, LayerNorm + adaLN modulation, fc1 slice, SiLU, fc2 K-slice,
//   publish the partial slab.  The workgroups of block 0 also run, for the row group they are about to start, the
//   tail of the previous Euler step (last residual update, final adaLN, output projection, x += dt * v) and the
//   input projection, redundantly in each of the G workgroups, so a step has exactly `blocks` hand-offs.
//
// Hand-off (producer -> the G consumers of the next block): payload stored write-through (sc1) in whole 128-byte
// lines, every storing wave drains its stores (s_waitcnt vmcnt(0)), workgroup barrier, one lane adds 1 to the
// (block, row group) counter with an agent-scope atomic; the consumer's lane 0 polls that counter with sc1 loads
// until it reaches G * (visits so far), a workgroup barrier follows, then every load of handed-off bytes is an sc1
// buffer load.  No dispatch order, co-location or cache state is assumed.  Counters only grow within a launch and are
// zeroed by an earlier kernel of the same frame.  Every spin is bounded: a workgroup that gives up raises the error
// word, which every other spin also watches, and the launch drains.
//
// Sums are formed in a fixed order (partials 0..G-1), so results are bitwise reproducible (and, with G = 8 and bf16
// matrices, equal to the launch chain's 8-slice variant).
//
// The float engine uses the same kernel: its matrices come as two bf16 planes each (W = hi + lo) and every tile takes a
// third MFMA (hi_a * lo_w beside hi_a * hi_w and lo_a * hi_w), which carries weights and activations to about 16 mantissa
// bits at bf16 MFMA speed; the float launch chain it replaces runs f32 MFMAs at a sixteenth of that rate.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <set>
#include <stdexcept>

#include "kernels.h"

namespace ptts {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;

constexpr int kThreads = 256, kWaves = 4;
constexpr int C = 512, L = 32, kMaxG = 8;
constexpr int KSTEP = 32, NT = C / 16, KS = C / KSTEP;
// Row pitches of the A-operand planes: 64 bytes past a multiple of 256, so that the 16 pieces (4 rows x 4 k-groups) one
// fragment read touches fall on 16 different groups of 4 banks.
constexpr int LDA = C + 32, LDX = L + 8;
constexpr int lda2_of(int g) { return C / g + 32; }
constexpr int NTL = L / 16, KPARTS = kWaves / NTL, KSP = KS / KPARTS;   // output projection: K split over the waves
constexpr int kCtlStride = 32;                               // words between counters: one 128-byte line each
constexpr unsigned kSpinLimit = 600000u;
constexpr int kTw = 128 + 8;                                 // row pitch of the per-wave transpose buffer (floats)
constexpr int T2 = NT / kWaves, TI = NT / kWaves;            // fc2 / input-projection tiles per wave
static_assert(KSP == 8 && NTL == 2 && T2 == 8, "flow ring: tiling written for flow_dim 512, latent 32, 4 waves");

// v * sigmoid(v) on the hardware exp2 / reciprocal (about 1 ulp each): this form serves the bf16 engine only
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f)); }

template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 64 lanes, the same value in every lane (four DPP steps inside each row of 16, then the four row sums)
__device__ __forceinline__ float wsum(float v) {
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x140>(v);
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}

// Mean and (biased) variance of one 512-wide row held as two float4 pieces per lane.  ONEPASS: the sums of x and of x * x
// are reduced side by side (two independent reductions whose latencies overlap) and var = E[x^2] - mean^2; otherwise the
// deviations are formed first, as the launch chain and the int8 form do (the int8 form must round as its oracle does).
template <bool ONEPASS>
__device__ __forceinline__ void row_moments(const float4 h0, const float4 h1, float& mean, float& var) {
    const float s1 = ((h0.x + h0.y) + (h0.z + h0.w)) + ((h1.x + h1.y) + (h1.z + h1.w));
    if constexpr (ONEPASS) {
        const float s2 = ((h0.x * h0.x + h0.y * h0.y) + (h0.z * h0.z + h0.w * h0.w)) + ((h1.x * h1.x + h1.y * h1.y) + (h1.z * h1.z + h1.w * h1.w));
        const float a1 = wsum(s1), a2 = wsum(s2);
        mean = a1 * (1.0f / (float)C);
        var = fmaxf(a2 * (1.0f / (float)C) - mean * mean, 0.f);
    } else {
        mean = wsum(s1) * (1.0f / (float)C);
        const float d0 = h0.x - mean, d1 = h0.y - mean, d2 = h0.z - mean, d3 = h0.w - mean;
        const float d4 = h1.x - mean, d5 = h1.y - mean, d6 = h1.z - mean, d7 = h1.w - mean;
        var = wsum(((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7))) * (1.0f / (float)C);
    }
}

// A-operand fragment of k-step k0 / 32: only the first RPG rows of a plane are live, MFMA row r reads row r % RPG (lanes
// that share a row read the same address, which the LDS broadcasts), so output rows >= RPG repeat live rows and are
// never stored.
template <int RPG>
__device__ __forceinline__ bf16x8 afrag(const __bf16* A, int lda, int k0, int lane) {
    return *(const bf16x8*)(A + (lane & (RPG - 1)) * lda + k0 + 8 * (lane >> 4));
}
// acc += (hi + lo) * fragment; with two-term weights (WS) also hi * (the fragment's second term)
template <bool WS>
__device__ __forceinline__ void mma2(const bf16x8 hi, const bf16x8 lo, const uint4 w, const uint4 wl, f32x4& acc) {
    const bf16x8 b8 = __builtin_bit_cast(bf16x8, w);
    if constexpr (WS) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, __builtin_bit_cast(bf16x8, wl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo, b8, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, b8, acc, 0, 0, 0);
}

// Output rows repeat with period 4 (see afrag), so the four 16-lane groups of an accumulator hold the same four rows:
// lane group q keeps row q.  One element per lane instead of four, and no lane is idle.
__device__ __forceinline__ float own_row(const f32x4 acc, int q) { return q == 0 ? acc[0] : q == 1 ? acc[1] : q == 2 ? acc[2] : acc[3]; }

__device__ __forceinline__ void put1(__bf16* A0, __bf16* A1, int idx, float v) {
    const __bf16 hi = (__bf16)v;
    A0[idx] = hi;
    A1[idx] = (__bf16)(v - (float)hi);
}

__device__ __forceinline__ void put4(__bf16* A0, __bf16* A1, int idx, const float4 v) {
    bf16x4 hi, lo;
    hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
    lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
    lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
    *(bf16x4*)(A0 + idx) = hi;
    *(bf16x4*)(A1 + idx) = lo;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }

// ---- int8 blocks (the int8 engine): the arithmetic of flow_block_q8_kernel in flow_head.hip.  A matrix input is quantised
// per row and per block of 64 consecutive k: s = max|a|, q = rint(a * 127 / s), d = s / 127;
// y[n] = (sum over blocks of (float)isum * d) * wscale[n] + bias[n].
typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ unsigned q8_pack4(const float4 v, float inv) {
    const int q0 = (int)rintf(v.x * inv), q1 = (int)rintf(v.y * inv), q2 = (int)rintf(v.z * inv), q3 = (int)rintf(v.w * inv);
    return (unsigned)(q0 & 0xff) | ((unsigned)(q1 & 0xff) << 8) | ((unsigned)(q2 & 0xff) << 16) | ((unsigned)(q3 & 0xff) << 24);
}
// maximum of |v| over the 16 consecutive lanes that hold one 64-wide block (one float4 each)
__device__ __forceinline__ float q8_absmax16(const float4 v) {
    float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xF, 0xF, true)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x141, 0xF, 0xF, true)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x140, 0xF, 0xF, true)));
    return m;
}
__device__ __forceinline__ int own_row_i(const i32x4 acc, int q) { return q == 0 ? acc[0] : q == 1 ? acc[1] : q == 2 ? acc[2] : acc[3]; }
__device__ __forceinline__ float silu_exact(float v) { return v / (1.0f + expf(-v)); }
constexpr int KSQ = C / 64, LDQ = C + 16;     // int8 k-steps of a full row; row pitch of the int8 A plane (bytes)
constexpr size_t q8_lds_bytes(int g) {         // Aq, dA, Sf, Sq, dS for 4 rows, rounded to 16 bytes
    return ((size_t)4 * LDQ + (size_t)KSQ * 4 * 4 + (size_t)4 * (C / g) * 4 + (size_t)4 * (C / g + 16) + (size_t)(KSQ / g) * 4 * 4 + 15) / 16 * 16;
}

// handed-off bytes: write-through stores and L1-bypassing loads (sc1), 16 bytes per lane
__device__ __forceinline__ float4 ld_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16));
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, const float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, byte_off, 0, 16);
}

// One lane waits until *cnt >= target.  false: gave up (or another workgroup did); the error word is then set.
__device__ __forceinline__ bool ring_wait(unsigned* cnt, unsigned target, unsigned* err) {
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        if (spins >= kSpinLimit) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        if ((spins & 31u) == 31u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

}  // namespace

struct FlowRingArgs {
    const float* x0; long long x0_stride; const int* x0_step; long long x0_step_stride; int x0_step_bias, step_ring;
    float* x_out; float* latents; int max_frames;
    const float* mods;
    int B, blocks, steps, mod_width, replicas, ngroups;
    const void* w_in; const float* b_in; const void* w_out; const float* b_out;
    const void* w_in_lo; const void* w_out_lo;   // second terms of two-term matrices (float engine), else null
    const void* w1[kFlowRingMaxBlocks]; const void* w2[kFlowRingMaxBlocks];
    const void* w1l[kFlowRingMaxBlocks]; const void* w2l[kFlowRingMaxBlocks];
    const float* s1[kFlowRingMaxBlocks]; const float* s2[kFlowRingMaxBlocks];   // int8 blocks: per-output-column weight scales
    const float* b1[kFlowRingMaxBlocks]; const float* b2[kFlowRingMaxBlocks];
    const float* ln_g[kFlowRingMaxBlocks]; const float* ln_b[kFlowRingMaxBlocks];
    float* ws; unsigned ws_bytes;        // hstage [blocks][B][C] then part [blocks][G][B][C], f32
    unsigned* ctl;                       // the error word, then counters [blocks][ngroups], one per 128-byte line
    float eps, dt;
    long long* stamps;                   // optional [16]: cycle counter along one mid-chain visit (tuning aid)
};

// RPG: utterance rows per row group (one wave per row in the element-wise phases).  G: workgroups per residual block
// (column slices of fc1 = K-slices of fc2): 4 keeps 256 registers of matrices per lane and halves the partial slabs a
// visit reads and the number of consumers that read them; 8 keeps 128.  WS: every matrix comes as two bf16 terms (the float
// engine's matrices to ~16 mantissa bits) and every tile takes a third MFMA; the register budget then allows G = 8 only.
// Q8: the residual blocks' matrices are int8 with per-column scales and their inputs are quantised per row and 64-wide block
// (v_mfma_i32_16x16x64_i8, half the matrix registers); the in / out projections stay bf16.
// IO2: the in / out projections alone come as two bf16 terms (the bf16 engine: those two small matrices sit at the ends of
// every Euler step and their storage rounding reaches the latent directly; their second planes live in block 0's LDS).
template <int RPG, int G, bool WS, bool Q8, bool IO2 = false>
__global__ __launch_bounds__(kThreads) void flow_ring_kernel(FlowRingArgs a) {
    constexpr bool WIO = WS || IO2;
#ifdef PTTS_RING_TWO_PASS_LN
    constexpr bool ONEPASS = false;
#else
    constexpr bool ONEPASS = !Q8;
#endif
    static_assert(!(WS && IO2) && !(Q8 && IO2), "flow ring: IO2 is the bf16 form with two-term projections");
    constexpr int NT1 = NT / G, NW1 = NT1 / kWaves, KS2 = KS / G, CS = C / G, LDA2 = lda2_of(G);
    constexpr int KS2Q = KSQ / G, LDSQ = CS + 16;
    static_assert(NW1 >= 1 && NW1 * kWaves == NT1 && KS2 * G == KS, "flow ring: slices must divide the tiles of fc1 and the k-steps of fc2");
    static_assert(!(WS && Q8) && (!Q8 || (KS2Q * G == KSQ && (CS / 4) % 16 == 0)), "flow ring: int8 blocks need whole 64-wide blocks per slice");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* hbuf = (float*)smem;                              // [RPG][C] f32: input-projection result; K-part sums of the output projection
    __bf16* A0 = (__bf16*)(hbuf + RPG * C);                   // [2][RPG][LDA]
    __bf16* A1 = A0 + RPG * LDA;
    __bf16* S0 = A1 + RPG * LDA;                             // [2][RPG][LDA2]
    __bf16* S1 = S0 + RPG * LDA2;
    __bf16* X0 = S1 + RPG * LDA2;                            // [2][RPG][LDX]
    __bf16* X1 = X0 + RPG * LDX;
    float* tbuf = (float*)(X1 + RPG * LDX);                   // [waves][RPG][128] f32: per-wave transpose of the fc2 tiles
    float* xkeep = tbuf + kWaves * RPG * kTw;                 // [local groups][RPG][L] f32: x of this workgroup's row groups (block 0)
    int* sflag = (int*)(xkeep + ((a.ngroups + a.replicas - 1) / a.replicas) * RPG * L);
    // int8 blocks: the quantised LayerNorm rows and their block scales, the SiLU slice in f32, quantised, and its block scales
    signed char* Aq = (signed char*)(sflag + 4);             // [RPG][LDQ]
    float* dA = (float*)(Aq + RPG * LDQ);                     // [KSQ][RPG]
    float* Sf = dA + KSQ * RPG;                               // [RPG][CS]
    signed char* Sq = (signed char*)(Sf + RPG * CS);          // [RPG][LDSQ]
    float* dS = (float*)(Sq + RPG * LDSQ);                    // [KS2Q][RPG]
    uint4* win_l = (uint4*)((unsigned char*)(sflag + 4) + (Q8 ? q8_lds_bytes(G) : 0));   // block 0: input projection [NT][64 lanes] fragments, then output projection [NTL * KS][64]
    uint4* wout_l = win_l + NT * 64;
    uint4* win_lo = wout_l + NTL * KS * 64;                   // WS: the second terms, same shapes
    uint4* wout_lo = win_lo + NT * 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = blockIdx.x;
    const int rep = wg / (a.blocks * G), blk = (wg / G) % a.blocks, g = wg % G;
    const int lq = lane >> 4, ccol = lane & 15;
    static_assert(RPG == 4, "flow ring: one accumulator row per 16-lane group");
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.ws, 0, (int)a.ws_bytes, 0x00020000);
    const size_t BC = (size_t)a.B * C;
    unsigned* err = a.ctl;                               // word 0; the counters follow from word kCtlStride on
    unsigned* const cnt0 = a.ctl + kCtlStride;
    const bool first_blk = blk == 0;

    // ---- this workgroup's matrices, once: fc1 tile `wave` of slice g over all of K; fc2 tiles 8 * wave .. 8 * wave + 7
    // (128 consecutive columns: four whole 128-byte lines of every output row) for K-slice g
    uint4 wf1[Q8 ? 1 : NW1][Q8 ? 1 : KS], wf2[Q8 ? 1 : T2][Q8 ? 1 : KS2];
    uint4 wq1[Q8 ? NW1 : 1][Q8 ? KSQ : 1], wq2[Q8 ? T2 : 1][Q8 ? KS2Q : 1];
    float sc1[Q8 ? NW1 : 1], sc2[Q8 ? T2 : 1];
    uint4 wf1l[WS ? NW1 : 1][WS ? KS : 1], wf2l[WS ? T2 : 1][WS ? KS2 : 1];
    if constexpr (Q8) {
        wf1[0][0] = make_uint4(0u, 0u, 0u, 0u); wf2[0][0] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int t = 0; t < NW1; ++t) {
            const uint4* w1p = (const uint4*)a.w1[blk] + ((long long)(g * NT1 + wave * NW1 + t) * KSQ) * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < KSQ; ++ks) wq1[t][ks] = w1p[ks * 64];
            sc1[t] = a.s1[blk][g * CS + (wave * NW1 + t) * 16 + ccol];
        }
        const uint4* w2p = (const uint4*)a.w2[blk] + (long long)(g * KS2Q) * 64 + lane;
#pragma unroll
        for (int i = 0; i < T2; ++i) {
#pragma unroll
            for (int ks = 0; ks < KS2Q; ++ks) wq2[i][ks] = w2p[((long long)(T2 * wave + i) * KSQ + ks) * 64];
            sc2[i] = a.s2[blk][(T2 * wave + i) * 16 + ccol];
        }
    } else {
        wq1[0][0] = make_uint4(0u, 0u, 0u, 0u); wq2[0][0] = make_uint4(0u, 0u, 0u, 0u); sc1[0] = 0.f; sc2[0] = 0.f;
    }
    if constexpr (WS) {
#pragma unroll
        for (int t = 0; t < NW1; ++t) {
            const uint4* w1p = (const uint4*)a.w1l[blk] + ((long long)(g * NT1 + wave * NW1 + t) * KS) * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf1l[t][ks] = w1p[ks * 64];
        }
        const uint4* w2p = (const uint4*)a.w2l[blk] + (long long)(g * KS2) * 64 + lane;
#pragma unroll
        for (int i = 0; i < T2; ++i) {
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) wf2l[i][ks] = w2p[((long long)(T2 * wave + i) * KS + ks) * 64];
        }
    } else {
        wf1l[0][0] = make_uint4(0u, 0u, 0u, 0u); wf2l[0][0] = make_uint4(0u, 0u, 0u, 0u);
    }
    if constexpr (!Q8) {
#pragma unroll
        for (int t = 0; t < NW1; ++t) {
            const uint4* w1p = (const uint4*)a.w1[blk] + ((long long)(g * NT1 + wave * NW1 + t) * KS) * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf1[t][ks] = w1p[ks * 64];
        }
        const uint4* w2p = (const uint4*)a.w2[blk] + (long long)(g * KS2) * 64 + lane;
#pragma unroll
        for (int i = 0; i < T2; ++i) {
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) wf2[i][ks] = w2p[((long long)(T2 * wave + i) * KS + ks) * 64];
        }
    }
    float bias1[NW1];
#pragma unroll
    for (int t = 0; t < NW1; ++t) bias1[t] = a.b1[blk][g * CS + (wave * NW1 + t) * 16 + ccol];
    // block 0 also needs the input projection (tiles wave, wave + 4, ...) and its share of the output projection: in LDS,
    // fragment-major as in global memory, so a wave's read of one fragment is 1 KB contiguous
    float bin[TI];
    const int to = wave % NTL, kpo = wave / NTL;
    if (first_blk) {
        for (int i = tid; i < NT * 64; i += kThreads) win_l[i] = ((const uint4*)a.w_in)[i];
        for (int i = tid; i < NTL * KS * 64; i += kThreads) wout_l[i] = ((const uint4*)a.w_out)[i];
        if constexpr (WIO) {
            for (int i = tid; i < NT * 64; i += kThreads) win_lo[i] = ((const uint4*)a.w_in_lo)[i];
            for (int i = tid; i < NTL * KS * 64; i += kThreads) wout_lo[i] = ((const uint4*)a.w_out_lo)[i];
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) bin[i] = a.b_in[(wave + kWaves * i) * 16 + ccol];
        __syncthreads();
    }
    const int c0 = 4 * lane, c1 = 4 * (lane + 64);          // this lane's two float4 pieces of a row
    // LayerNorm affine of this block and the fc2 bias of the block before it (block 0: of the last block, for the tail)
    const int pblk = first_blk ? a.blocks - 1 : blk - 1;
    const float4 lg0 = ld4(a.ln_g[blk] + c0), lg1 = ld4(a.ln_g[blk] + c1), lb0 = ld4(a.ln_b[blk] + c0), lb1 = ld4(a.ln_b[blk] + c1);
    const float4 pb0 = ld4(a.b2[pblk] + c0), pb1 = ld4(a.b2[pblk] + c1);

    const unsigned h_off_prev = (unsigned)((size_t)pblk * BC * 4);
    const unsigned p_off_prev = (unsigned)(((size_t)a.blocks * BC + (size_t)pblk * G * BC) * 4);
    const unsigned h_off_mine = (unsigned)((size_t)blk * BC * 4);
    const unsigned p_off_mine = (unsigned)(((size_t)a.blocks * BC + ((size_t)blk * G + g) * BC) * 4);

    long long fidx = 0;   // frame index of this call, optionally wrapped into a ring
    if (a.x0_step) { const int v = *a.x0_step + a.x0_step_bias; fidx = a.step_ring > 0 ? v % a.step_ring : v; }
    // One pass of the tail of an Euler step for the row group (block-0 workgroups): h of the last block's output ->
    // final adaLN -> output projection -> x += dt * v.  Leaves x in xk (LDS) and, bf16-split, in the X planes.
    auto step_tail = [&](const float* mods_prev, int row0, int nrows, float* xk, const float* xsrc, long long xsrc_stride, bool write_out) {
        const bool live = wave < nrows;
        if (live) {
            const long long grow = row0 + wave;
            const float* mrow = mods_prev + grow * a.mod_width;
            const float* mfin = mrow + a.blocks * 3 * C;
            const float* gate = mrow + (a.blocks - 1) * 3 * C + 2 * C;
            const float4 sh0 = ld4(mfin + c0), sh1 = ld4(mfin + c1), sc0 = ld4(mfin + C + c0), sc1v = ld4(mfin + C + c1);
            const float4 gt0 = ld4(gate + c0), gt1 = ld4(gate + c1);
            const unsigned ro = (unsigned)(grow * C * 4);
            float4 h0 = ld_sc1(rs, h_off_prev + ro + c0 * 4), h1 = ld_sc1(rs, h_off_prev + ro + c1 * 4);
            float4 p0[G], p1[G];
#pragma unroll
            for (int s = 0; s < G; ++s) {
                p0[s] = ld_sc1(rs, p_off_prev + (unsigned)(s * BC * 4) + ro + c0 * 4);
                p1[s] = ld_sc1(rs, p_off_prev + (unsigned)(s * BC * 4) + ro + c1 * 4);
            }
            float4 s0 = p0[0], s1 = p1[0];
#pragma unroll
            for (int s = 1; s < G; ++s) {
                s0.x += p0[s].x; s0.y += p0[s].y; s0.z += p0[s].z; s0.w += p0[s].w;
                s1.x += p1[s].x; s1.y += p1[s].y; s1.z += p1[s].z; s1.w += p1[s].w;
            }
            h0.x += gt0.x * (s0.x + pb0.x); h0.y += gt0.y * (s0.y + pb0.y); h0.z += gt0.z * (s0.z + pb0.z); h0.w += gt0.w * (s0.w + pb0.w);
            h1.x += gt1.x * (s1.x + pb1.x); h1.y += gt1.y * (s1.y + pb1.y); h1.z += gt1.z * (s1.z + pb1.z); h1.w += gt1.w * (s1.w + pb1.w);
            float mean, var;
            row_moments<ONEPASS>(h0, h1, mean, var);
            const float d0 = h0.x - mean, d1 = h0.y - mean, d2 = h0.z - mean, d3 = h0.w - mean;
            const float d4 = h1.x - mean, d5 = h1.y - mean, d6 = h1.z - mean, d7 = h1.w - mean;
            const float rstd = __builtin_amdgcn_rsqf(var + a.eps);
            float4 y0, y1;
            y0.x = (d0 * rstd) * (1.0f + sc0.x) + sh0.x; y0.y = (d1 * rstd) * (1.0f + sc0.y) + sh0.y;
            y0.z = (d2 * rstd) * (1.0f + sc0.z) + sh0.z; y0.w = (d3 * rstd) * (1.0f + sc0.w) + sh0.w;
            y1.x = (d4 * rstd) * (1.0f + sc1v.x) + sh1.x; y1.y = (d5 * rstd) * (1.0f + sc1v.y) + sh1.y;
            y1.z = (d6 * rstd) * (1.0f + sc1v.z) + sh1.z; y1.w = (d7 * rstd) * (1.0f + sc1v.w) + sh1.w;
            put4(A0, A1, wave * LDA + c0, y0);
            put4(A0, A1, wave * LDA + c1, y1);
        }
        __syncthreads();
        f32x4 acco = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            bf16x8 fh[KSP], fl[KSP];
#pragma unroll
            for (int ks = 0; ks < KSP; ++ks) { fh[ks] = afrag<RPG>(A0, LDA, (kpo * KSP + ks) * KSTEP, lane); fl[ks] = afrag<RPG>(A1, LDA, (kpo * KSP + ks) * KSTEP, lane); }
#pragma unroll
            for (int ks = 0; ks < KSP; ++ks) {
                const int wi = (to * KS + kpo * KSP + ks) * 64 + lane;
                mma2<WIO>(fh[ks], fl[ks], wout_l[wi], WIO ? wout_lo[wi] : wout_l[wi], acco);
            }
        }
        *(f32x4*)(hbuf + wave * 256 + lane * 4) = acco;
        __syncthreads();
        for (int i = tid; i < RPG * L; i += kThreads) {
            const int r = i / L, c = i % L;
            float v = 0.f;
            if (r < nrows) {
                const int tt = c / 16, ln = (r / 4) * 16 + (c % 16), ri = r % 4;   // accumulator of tile tt: lane ln, element ri
                float sacc = 0.f;
#pragma unroll
                for (int k = 0; k < KPARTS; ++k) sacc += hbuf[(k * NTL + tt) * 256 + ln * 4 + ri];
                const float xin = xsrc ? xsrc[(long long)(row0 + r) * xsrc_stride + c] : xk[r * L + c];
                v = xin + (sacc + a.b_out[c]) * a.dt;
                xk[r * L + c] = v;
                if (write_out) {
                    const long long row = row0 + r;
                    a.x_out[row * L + c] = v;
                    if (a.latents) a.latents[(row * a.max_frames + fidx) * L + c] = v;
                }
            }
            put1(X0, X1, r * LDX + c, v);
        }
    };

    const bool stamp_wg = a.stamps && rep == 0 && blk == (a.blocks > 3 ? 3 : 1) && g == 0 && tid == 0;
#define RING_STAMP(i) do { if (stamp_wg && j == a.steps / 2 && q == rep) a.stamps[i] = __builtin_readcyclecounter(); } while (0)
    const float* x0p = a.x0 + fidx * a.x0_step_stride;

    for (int j = 0; j < a.steps; ++j) {
        const float* mods_j = a.mods + (size_t)j * a.B * a.mod_width;
        int qi = 0;
        for (int q = rep; q < a.ngroups; q += a.replicas, ++qi) {
            const int row0 = q * RPG;
            const int nrows = (a.B - row0) < RPG ? (a.B - row0) : RPG;
            const bool live = wave < nrows;
            const long long grow = row0 + (live ? wave : 0);
            // ---- operands that do not depend on the chain: this step's adaLN rows for this block (and the previous block's gate)
            const float* mrow = mods_j + grow * a.mod_width;
            float4 sh0, sh1, sc0, sc1v, gt0, gt1;
            if (live) {
                const float* mblk = mrow + blk * 3 * C;
                sh0 = ld4(mblk + c0); sh1 = ld4(mblk + c1); sc0 = ld4(mblk + C + c0); sc1v = ld4(mblk + C + c1);
                if (!first_blk) { gt0 = ld4(mrow + (blk - 1) * 3 * C + 2 * C + c0); gt1 = ld4(mrow + (blk - 1) * 3 * C + 2 * C + c1); }
            }
            RING_STAMP(0);
            // ---- wait for the producers of this row group
            if (!first_blk || j > 0) {
                if (tid == 0) {
                    unsigned* cnt = cnt0 + ((size_t)pblk * a.ngroups + q) * kCtlStride;
                    const unsigned target = (unsigned)G * (unsigned)(first_blk ? j : j + 1);
                    *sflag = ring_wait(cnt, target, err) ? 1 : 0;
                    RING_STAMP(1);
                }
                __syncthreads();
                if (*sflag == 0) return;
            }
            RING_STAMP(2);
            float4 h0, h1;
            if (first_blk) {
                float* xk = xkeep + qi * RPG * L;
                if (j > 0) {
                    step_tail(a.mods + (size_t)(j - 1) * a.B * a.mod_width, row0, nrows, xk, j == 1 ? x0p : nullptr, a.x0_stride, false);
                } else {
                    for (int i = tid; i < RPG * L; i += kThreads) {
                        const int r = i / L, c = i % L;
                        put1(X0, X1, r * LDX + c, r < nrows ? x0p[(long long)(row0 + r) * a.x0_stride + c] : 0.f);
                    }
                }
                __syncthreads();
                // input projection: h = W_in x + b_in
                const bf16x8 xh = afrag<RPG>(X0, LDX, 0, lane), xl = afrag<RPG>(X1, LDX, 0, lane);
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    const int wi = (wave + kWaves * i) * 64 + lane;
                    mma2<WIO>(xh, xl, win_l[wi], WIO ? win_lo[wi] : win_l[wi], acc);
                    const int c = (wave + kWaves * i) * 16 + ccol;
                    hbuf[lq * C + c] = own_row(acc, lq) + bin[i];
                }
                __syncthreads();
                if (live) { h0 = ld4(hbuf + wave * C + c0); h1 = ld4(hbuf + wave * C + c1); }
            } else if (live) {
                const unsigned ro = (unsigned)(grow * C * 4);
                h0 = ld_sc1(rs, h_off_prev + ro + c0 * 4);
                h1 = ld_sc1(rs, h_off_prev + ro + c1 * 4);
                float4 p0[G], p1[G];
#pragma unroll
                for (int s = 0; s < G; ++s) {
                    p0[s] = ld_sc1(rs, p_off_prev + (unsigned)(s * BC * 4) + ro + c0 * 4);
                    p1[s] = ld_sc1(rs, p_off_prev + (unsigned)(s * BC * 4) + ro + c1 * 4);
                }
                float4 s0 = p0[0], s1 = p1[0];
#pragma unroll
                for (int s = 1; s < G; ++s) {
                    s0.x += p0[s].x; s0.y += p0[s].y; s0.z += p0[s].z; s0.w += p0[s].w;
                    s1.x += p1[s].x; s1.y += p1[s].y; s1.z += p1[s].z; s1.w += p1[s].w;
                }
                h0.x += gt0.x * (s0.x + pb0.x); h0.y += gt0.y * (s0.y + pb0.y); h0.z += gt0.z * (s0.z + pb0.z); h0.w += gt0.w * (s0.w + pb0.w);
                h1.x += gt1.x * (s1.x + pb1.x); h1.y += gt1.y * (s1.y + pb1.y); h1.z += gt1.z * (s1.z + pb1.z); h1.w += gt1.w * (s1.w + pb1.w);
            }
            RING_STAMP(3);
            if (live) {
                if (g == 0) {   // one workgroup of the block keeps the stream for the next block (whole 1 KB runs per wave)
                    const unsigned ro = (unsigned)(grow * C * 4);
                    st_sc1(rs, h_off_mine + ro + c0 * 4, h0);
                    st_sc1(rs, h_off_mine + ro + c1 * 4, h1);
                }
                float mean, var0;
                row_moments<ONEPASS>(h0, h1, mean, var0);
                const float d0 = h0.x - mean, d1 = h0.y - mean, d2 = h0.z - mean, d3 = h0.w - mean;
                const float d4 = h1.x - mean, d5 = h1.y - mean, d6 = h1.z - mean, d7 = h1.w - mean;
                const float var = var0 + a.eps;
                const float rstd = Q8 ? 1.0f / sqrtf(var) : __builtin_amdgcn_rsqf(var);
                float4 y0, y1;
                y0.x = (d0 * rstd * lg0.x + lb0.x) * (1.0f + sc0.x) + sh0.x; y0.y = (d1 * rstd * lg0.y + lb0.y) * (1.0f + sc0.y) + sh0.y;
                y0.z = (d2 * rstd * lg0.z + lb0.z) * (1.0f + sc0.z) + sh0.z; y0.w = (d3 * rstd * lg0.w + lb0.w) * (1.0f + sc0.w) + sh0.w;
                y1.x = (d4 * rstd * lg1.x + lb1.x) * (1.0f + sc1v.x) + sh1.x; y1.y = (d5 * rstd * lg1.y + lb1.y) * (1.0f + sc1v.y) + sh1.y;
                y1.z = (d6 * rstd * lg1.z + lb1.z) * (1.0f + sc1v.z) + sh1.z; y1.w = (d7 * rstd * lg1.w + lb1.w) * (1.0f + sc1v.w) + sh1.w;
                if constexpr (Q8) {   // a 64-wide block is the float4 pieces of 16 consecutive lanes
                    const float m0 = q8_absmax16(y0), m1 = q8_absmax16(y1);
                    *(unsigned*)(Aq + wave * LDQ + c0) = q8_pack4(y0, m0 > 0.f ? 127.0f / m0 : 0.f);
                    *(unsigned*)(Aq + wave * LDQ + c1) = q8_pack4(y1, m1 > 0.f ? 127.0f / m1 : 0.f);
                    if (ccol == 0) { dA[lq * RPG + wave] = m0 / 127.0f; dA[(4 + lq) * RPG + wave] = m1 / 127.0f; }
                } else {
                    put4(A0, A1, wave * LDA + c0, y0);
                    put4(A0, A1, wave * LDA + c1, y1);
                }
            }
            __syncthreads();
            RING_STAMP(4);
            // ---- fc1: this workgroup's C / G columns, NW1 tiles per wave over all of K
            if constexpr (Q8) {
                float v[NW1];
#pragma unroll
                for (int t = 0; t < NW1; ++t) v[t] = 0.f;
                // every fragment and block scale is requested before the first MFMA (one LDS latency, not one per k-step)
                i32x4 a4[KSQ];
                float d[KSQ];
#pragma unroll
                for (int ks = 0; ks < KSQ; ++ks) { a4[ks] = *(const i32x4*)(Aq + (lane & (RPG - 1)) * LDQ + ks * 64 + 16 * lq); d[ks] = dA[ks * RPG + lq]; }
                __builtin_amdgcn_sched_barrier(0);
                i32x4 ci[NW1][KSQ];
#pragma unroll
                for (int ks = 0; ks < KSQ; ++ks) {
#pragma unroll
                    for (int t = 0; t < NW1; ++t)
                        ci[t][ks] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4[ks], __builtin_bit_cast(i32x4, wq1[t][ks]), i32x4{0, 0, 0, 0}, 0, 0, 0);
                }
#pragma unroll
                for (int ks = 0; ks < KSQ; ++ks) {   // block sums in k order, as the launch chain forms them
#pragma unroll
                    for (int t = 0; t < NW1; ++t) v[t] += (float)own_row_i(ci[t][ks], lq) * d[ks];
                }
#pragma unroll
                for (int t = 0; t < NW1; ++t) Sf[lq * CS + (wave * NW1 + t) * 16 + ccol] = silu_exact(v[t] * sc1[t] + bias1[t]);
                __syncthreads();
                for (int id = tid; id < RPG * (CS / 4); id += kThreads) {   // whole waves: a block's 16 pieces sit in one DPP row
                    const int row = id / (CS / 4), piece = id % (CS / 4);
                    const float4 y = *(const float4*)(Sf + row * CS + 4 * piece);
                    const float m = q8_absmax16(y);
                    *(unsigned*)(Sq + row * LDSQ + 4 * piece) = q8_pack4(y, m > 0.f ? 127.0f / m : 0.f);
                    if ((piece & 15) == 0) dS[(piece >> 4) * RPG + row] = m / 127.0f;
                }
            } else {
                // groups of 4 k-steps: the fragment reads of group n + 1 are issued before the MFMAs of group n (the pins keep
                // the compiler from sinking each read to its use, which exposes one LDS latency per MFMA pair)
                constexpr int GK = 4, NG = KS / GK;
                bf16x8 fh[2][GK], fl[2][GK];
                f32x4 acc[NW1];
#pragma unroll
                for (int t = 0; t < NW1; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < GK; ++k) { fh[0][k] = afrag<RPG>(A0, LDA, k * KSTEP, lane); fl[0][k] = afrag<RPG>(A1, LDA, k * KSTEP, lane); }
#pragma unroll
                for (int n = 0; n < NG; ++n) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (n + 1 < NG) {
#pragma unroll
                        for (int k = 0; k < GK; ++k) {
                            fh[(n + 1) & 1][k] = afrag<RPG>(A0, LDA, ((n + 1) * GK + k) * KSTEP, lane);
                            fl[(n + 1) & 1][k] = afrag<RPG>(A1, LDA, ((n + 1) * GK + k) * KSTEP, lane);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < GK; ++k) {
#pragma unroll
                        for (int t = 0; t < NW1; ++t) mma2<WS>(fh[n & 1][k], fl[n & 1][k], wf1[t][n * GK + k], wf1l[WS ? t : 0][WS ? n * GK + k : 0], acc[t]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NW1; ++t) {
                    const int cl = (wave * NW1 + t) * 16 + ccol;
                    put1(S0, S1, lq * LDA2 + cl, silu_f(own_row(acc[t], lq) + bias1[t]));
                }
            }
            __syncthreads();
            RING_STAMP(5);
            // ---- fc2: every output column, this workgroup's K-slice -> partial slab, stored as whole lines
            {
                float* tw = tbuf + wave * RPG * kTw;
                if constexpr (Q8) {
                    i32x4 sq[KS2Q];
                    float ds[KS2Q];
#pragma unroll
                    for (int ks = 0; ks < KS2Q; ++ks) { sq[ks] = *(const i32x4*)(Sq + (lane & (RPG - 1)) * LDSQ + ks * 64 + 16 * lq); ds[ks] = dS[ks * RPG + lq]; }
                    i32x4 ci[T2][KS2Q];
#pragma unroll
                    for (int i = 0; i < T2; ++i) {
#pragma unroll
                        for (int ks = 0; ks < KS2Q; ++ks)
                            ci[i][ks] = __builtin_amdgcn_mfma_i32_16x16x64_i8(sq[ks], __builtin_bit_cast(i32x4, wq2[i][ks]), i32x4{0, 0, 0, 0}, 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < T2; ++i) {
                        float v = 0.f;
#pragma unroll
                        for (int ks = 0; ks < KS2Q; ++ks) v += (float)own_row_i(ci[i][ks], lq) * ds[ks];
                        tw[lq * kTw + i * 16 + ccol] = v * sc2[i];
                    }
                } else {
                bf16x8 sh_[KS2], sl_[KS2];
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks) { sh_[ks] = afrag<RPG>(S0, LDA2, ks * KSTEP, lane); sl_[ks] = afrag<RPG>(S1, LDA2, ks * KSTEP, lane); }
                f32x4 acc2[T2];
#pragma unroll
                for (int i = 0; i < T2; ++i) {
                    acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS2; ++ks) mma2<WS>(sh_[ks], sl_[ks], wf2[i][ks], wf2l[WS ? i : 0][WS ? ks : 0], acc2[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < T2; ++i) tw[lq * kTw + i * 16 + ccol] = own_row(acc2[i], lq);
                }
                __builtin_amdgcn_wave_barrier();
                // lane -> (row u, 16-byte chunk c of the wave's 512-byte run): 8 consecutive lanes write one whole 128-byte line
#pragma unroll
                for (int it = 0; it < RPG / 2; ++it) {
                    const int u = it * 2 + (lane >> 5), c = lane & 31;
                    const float4 v = *(const float4*)(tw + u * kTw + 4 * c);
                    if (u < nrows) st_sc1(rs, p_off_mine + (unsigned)(((size_t)(row0 + u) * C + 128 * wave + 4 * c) * 4), v);
                }
            }
            RING_STAMP(6);
            // ---- publish: every storing wave drains, the workgroup meets, one lane signals (a signal per wave was measured
            // slower: four times the atomics on one counter)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            RING_STAMP(7);
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(cnt0 + ((size_t)blk * a.ngroups + q) * kCtlStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            RING_STAMP(8);
        }
    }
    // ---- the last Euler step's tail: slice 0 of block 0 finishes each of its row groups and writes the result
    if (first_blk && g == 0) {
        int qi = 0;
        for (int q = rep; q < a.ngroups; q += a.replicas, ++qi) {
            const int row0 = q * RPG;
            const int nrows = (a.B - row0) < RPG ? (a.B - row0) : RPG;
            if (tid == 0) {
                unsigned* cnt = cnt0 + ((size_t)pblk * a.ngroups + q) * kCtlStride;
                *sflag = ring_wait(cnt, (unsigned)G * (unsigned)a.steps, err) ? 1 : 0;
            }
            __syncthreads();
            if (*sflag == 0) return;
            step_tail(a.mods + (size_t)(a.steps - 1) * a.B * a.mod_width, row0, nrows, xkeep + qi * RPG * L, a.steps == 1 ? x0p : nullptr, a.x0_stride, true);
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side

// Workgroups per residual block: 4 by default, FLOW_RING_SLICES=8 selects the earlier form; two-term matrices need 8.
static int ring_slices(bool split) {
    static const int g = [] { const char* e = std::getenv("FLOW_RING_SLICES"); return e && std::atoi(e) == 8 ? 8 : 4; }();
    return split ? 8 : g;
}

static void flow_ring_shape(int B, int blocks, int budget, bool split, int& rpg, int& replicas);
static size_t ring_smem(int rpg, int ngroups, int replicas, bool split, bool q8, bool io2);

// bf16 blocks with the second planes of the in / out projections present: those two products take the third MFMA
static bool ring_io2(const FlowHeadArgs& a, bool split, bool q8) {
    return !split && !q8 && a.host_tabs && a.host_tabs->r_in_lo && a.host_tabs->r_out_lo;
}

// Workgroups of the persistent launch that can be resident at once on the current device: what the runtime says one CU holds
// of the largest variant (512 registers per lane and up to 160 KB of LDS: one) times the CU count.  Queried once per device
// by flow_ring_prepare(), outside any stream capture; 0 = not queried (no limit known).
static std::atomic<int> g_ring_resident[64];

static int ring_resident_limit() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return g_ring_resident[dev & 63].load();
}

bool flow_ring_supported(const FlowHeadArgs& a) {
    if (std::getenv("FLOW_HEAD_NO_RING")) return false;
    const int bw = a.block_wtype < 0 ? a.wtype : a.block_wtype;
    const bool split = a.wtype == W_F32 && bw == W_F32 && a.ring_split;
    if (!split && !(a.wtype == W_BF16 && (bw == W_BF16 || bw == W_I8))) return false;
    const int G = ring_slices(split);
    if (!(a.C == C && a.L == L && a.blocks >= 2 && a.blocks <= kFlowRingMaxBlocks && a.steps >= 1 &&
          a.ring_ctl != nullptr && a.B >= 1 && a.ring_wg_budget >= a.blocks * G))
        return false;
    // the row groups a workgroup serves keep their x rows in LDS: many rows on few replicas may not fit (the launch chain runs then)
    int rpg = 4, replicas = 1;
    flow_ring_shape(a.B, a.blocks, a.ring_wg_budget, split, rpg, replicas);
    // every workgroup spins on its producers: a grid that cannot be resident all at once never finishes a hand-off
    const int limit = ring_resident_limit();
    if (limit > 0 && replicas * a.blocks * G > limit) return false;
    return ring_smem(rpg, (a.B + rpg - 1) / rpg, replicas, split, bw == W_I8, ring_io2(a, split, bw == W_I8)) <= 160 * 1024;
}

// Row-group height and replica count for B rows.  More replicas shorten each workgroup's queue of row groups; the
// chain of one row group (steps * blocks hand-offs) is the floor.  FLOW_RING_REPLICAS overrides.
static void flow_ring_shape(int B, int blocks, int budget, bool split, int& rpg, int& replicas) {
    const int G = ring_slices(split);
    rpg = 4;   // one row per wave
    const int ngroups = (B + rpg - 1) / rpg;
    // measured at B = 64 beside the decoder stream with 8 slices: 2 replicas 0.65 ms per flow phase, 3 0.49, 4 0.43 (launch
    // chain 0.54).  With 4 slices 192 workgroups would make 8 replicas, and alone on the chip that is the fastest form; beside
    // the decoder stream, which is confined to 128 of the 256 CUs, 4 or 5 replicas (96 / 120 workgroups: every one finds a CU
    // the decoder never uses, and a workgroup of this kernel needs a CU's whole register file) are faster than 6 to 8, whose
    // launch waits for decoder workgroups to drain: 6400 against 6250 RTFx
    replicas = G == 8 ? 4 : 5;
    if (const char* e = std::getenv("FLOW_RING_REPLICAS")) replicas = std::atoi(e);
    const int max_rep = budget / (blocks * G);     // every workgroup must be resident at once; leave CUs for other streams
    if (replicas > max_rep) replicas = max_rep;
    if (replicas > ngroups) replicas = ngroups;
    if (replicas < 1) replicas = 1;
}

size_t flow_ring_workspace_bytes(int B, int blocks) { return (size_t)blocks * (1 + kMaxG) * B * C * sizeof(float); }   // sized for either form

size_t flow_ring_ctl_bytes(int B, int blocks) {
    const int ngroups = (B + 3) / 4;   // the smaller row-group height
    return ((size_t)blocks * ngroups * kCtlStride + kCtlStride) * sizeof(unsigned);
}

// Control block: word 0 = error word (set when a hand-off wait gave up; cleared by the host only), then from word
// flow_ring_counter_word() on the (block, row group) counters, which an earlier kernel of the frame zeroes.
size_t flow_ring_err_word(int, int) { return 0; }
size_t flow_ring_counter_word() { return kCtlStride; }

static size_t ring_smem(int rpg, int ngroups, int replicas, bool split, bool q8, bool io2) {
    const int local = (ngroups + replicas - 1) / replicas;
    const int LDA2 = lda2_of(ring_slices(split));
    return (size_t)rpg * C * 4 + (size_t)2 * rpg * (LDA + LDA2 + LDX) * 2 + (size_t)kWaves * rpg * kTw * 4 + (size_t)local * rpg * L * 4 + 16 +
           (q8 ? q8_lds_bytes(ring_slices(split)) : 0) + (size_t)(NT + NTL * KS) * 64 * 16 * (split || io2 ? 2 : 1);
}

static void ring_lds_limit(const void* kernel) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    if (done.insert({kernel, dev}).second) (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

void flow_ring_prepare() {
    {
        int dev = 0, per_cu = 0;
        (void)hipGetDevice(&dev);
        hipDeviceProp_t prop;
        ring_lds_limit((const void*)flow_ring_kernel<4, 4, false, false, true>);
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)flow_ring_kernel<4, 4, false, false, true>, kThreads, 158 * 1024) == hipSuccess && per_cu > 0)
            g_ring_resident[dev & 63].store(per_cu * prop.multiProcessorCount);
        else
            (void)hipGetLastError();
    }
    ring_lds_limit((const void*)flow_ring_kernel<4, 8, false, false>);
    ring_lds_limit((const void*)flow_ring_kernel<4, 4, false, false>);
    ring_lds_limit((const void*)flow_ring_kernel<4, 8, true, false>);
    ring_lds_limit((const void*)flow_ring_kernel<4, 8, false, true>);
    ring_lds_limit((const void*)flow_ring_kernel<4, 4, false, true>);
    ring_lds_limit((const void*)flow_ring_kernel<4, 8, false, false, true>);
    ring_lds_limit((const void*)flow_ring_kernel<4, 4, false, false, true>);
}

// The caller has zeroed a.ring_ctl (flow_ring_ctl_bytes) earlier on the same stream.
void launch_flow_ring(const FlowHeadArgs& a, hipStream_t s) {
    if (!flow_ring_supported(a)) throw std::runtime_error("flow ring: unsupported shape");
    if (a.x_out == a.x0) throw std::runtime_error("flow ring: x_out must not alias x0");
    int rpg = 4, replicas = 1;
    const bool split = a.wtype == W_F32;
    const bool q8 = (a.block_wtype < 0 ? a.wtype : a.block_wtype) == W_I8;
    flow_ring_shape(a.B, a.blocks, a.ring_wg_budget, split, rpg, replicas);
    FlowRingArgs r{};
    r.x0 = a.x0; r.x0_stride = a.x0_stride; r.x0_step = a.x0_step; r.x0_step_stride = a.x0_step_stride; r.x0_step_bias = a.x0_step_bias; r.step_ring = a.step_ring;
    r.x_out = a.x_out; r.latents = a.latents; r.max_frames = a.max_frames;
    r.mods = a.mods; r.B = a.B; r.blocks = a.blocks; r.steps = a.steps; r.mod_width = a.mod_width;
    r.replicas = replicas; r.ngroups = (a.B + rpg - 1) / rpg;
    const FlowHostTabs& t = *a.host_tabs;
    if (!t.r_in || !t.r_out || (int)t.r_fc1.size() < a.blocks || (int)t.r_fc2.size() < a.blocks) throw std::runtime_error("flow ring: matrix planes missing");
    if (split && (!t.r_in_lo || !t.r_out_lo || (int)t.r_fc1_lo.size() < a.blocks || (int)t.r_fc2_lo.size() < a.blocks))
        throw std::runtime_error("flow ring: second matrix planes missing");
    r.w_in = t.r_in; r.b_in = a.b_in; r.w_out = t.r_out; r.b_out = a.b_out;
    const bool io2 = ring_io2(a, split, q8);
    r.w_in_lo = split || io2 ? t.r_in_lo : nullptr; r.w_out_lo = split || io2 ? t.r_out_lo : nullptr;
    for (int i = 0; i < a.blocks; ++i) {
        r.w1[i] = t.r_fc1[i]; r.w2[i] = t.r_fc2[i]; r.b1[i] = t.b_fc1[i]; r.b2[i] = t.b_fc2[i]; r.ln_g[i] = t.ln_g[i]; r.ln_b[i] = t.ln_b[i];
        r.w1l[i] = split ? t.r_fc1_lo[i] : nullptr; r.w2l[i] = split ? t.r_fc2_lo[i] : nullptr;
        r.s1[i] = q8 ? t.s_fc1[i] : nullptr; r.s2[i] = q8 ? t.s_fc2[i] : nullptr;
        if (q8 && (!r.s1[i] || !r.s2[i])) throw std::runtime_error("flow ring: int8 blocks without weight scales");
    }
    r.ws = a.workspace;
    const size_t wsb = flow_ring_workspace_bytes(a.B, a.blocks);
    if (wsb > 0x7fffffffu) throw std::runtime_error("flow ring: workspace beyond 2 GiB");
    r.ws_bytes = (unsigned)wsb;
    r.ctl = a.ring_ctl;
    r.eps = a.eps; r.dt = 1.0f / (float)a.steps;
    r.stamps = a.cycle_probe;
    size_t smem = ring_smem(rpg, r.ngroups, replicas, split, q8, io2);
    if (const char* e = std::getenv("FLOW_RING_LDS_PAD")) smem = std::min<size_t>(smem + (size_t)std::atoi(e), 160 * 1024);   // experiment: what the LDS size alone costs
    if (smem > 160 * 1024) throw std::runtime_error("flow ring: LDS request too large");
    const int G = ring_slices(split);
    int grid = replicas * a.blocks * G;
    // test hook: leave the last workgroup out, so that its consumers wait, give up after the spin limit and raise the error word
    if (const char* e = std::getenv("FLOW_RING_TEST_DROP_WG")) if (std::atoi(e) > 0 && grid > 1) grid -= 1;
    if (a.probe_a) (void)hipEventRecord(a.probe_a, s);
    if (split) hipLaunchKernelGGL((flow_ring_kernel<4, 8, true, false>), dim3(grid), dim3(kThreads), smem, s, r);
    else if (q8 && G == 8) hipLaunchKernelGGL((flow_ring_kernel<4, 8, false, true>), dim3(grid), dim3(kThreads), smem, s, r);
    else if (q8) hipLaunchKernelGGL((flow_ring_kernel<4, 4, false, true>), dim3(grid), dim3(kThreads), smem, s, r);
    else if (io2 && G == 8) hipLaunchKernelGGL((flow_ring_kernel<4, 8, false, false, true>), dim3(grid), dim3(kThreads), smem, s, r);
    else if (io2) hipLaunchKernelGGL((flow_ring_kernel<4, 4, false, false, true>), dim3(grid), dim3(kThreads), smem, s, r);
    else if (G == 8) hipLaunchKernelGGL((flow_ring_kernel<4, 8, false, false>), dim3(grid), dim3(kThreads), smem, s, r);
    else hipLaunchKernelGGL((flow_ring_kernel<4, 4, false, false>), dim3(grid), dim3(kThreads), smem, s, r);
    if (a.probe_b) (void)hipEventRecord(a.probe_b, s);
}

}  // namespace ptts
