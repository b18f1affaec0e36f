// LSD flow-matching head: lsd_steps Euler iterations of the adaLN MLP v(c, s, t, x)
// (/root/reference/src/pocket_tts.cpp:414-451 run lsd_steps times per frame, :615-622 the x += v * dt loop).
//
// Shape of the problem on MI355X: per frame the head is a chain of 2 * blocks * steps (= 120) dependent
// [B x C] x [C x C] products with B ~ 64 rows.  One workgroup that walks the whole chain has to pull every
// weight byte through one CU's L2 port (~40-100 GB/s), so the chain is cut where the data dependence forces
// an exchange anyway and every link is spread over several CUs:
//
//   flow_block_kernel   one launch per residual block, grid (G, row groups of 16).  Workgroup g owns the
//       column slice [g*C/G, (g+1)*C/G) of fc1 and the matching K-slice of fc2, so the pair needs no
//       exchange in between; it leaves an f32 partial [16][C] of the block's output.  Every workgroup of a
//       row group first rebuilds the same residual stream h = h_prev + gate * (sum of the previous block's G
//       partials + bias) - a fixed-order, bitwise reproducible sum - and its LayerNorm/adaLN modulation.
//       All weight fragments of the launch (C*C/G*2 elements per workgroup) are requested before anything
//       else, so they stream in while the partials arrive and the norm is computed.
//   flow_final_kernel   one launch per Euler step: last residual update, final adaLN, output projection,
//       x += dt * v.
//
// A kernel boundary on one stream costs ~1.5 us on this part, about what an in-launch flag hand-off between
// workgroups costs, and needs no co-residency or cache-scope protocol.
//
// Weights are fragment-major (pack_flow_weight): one wave-level load = 1 KiB contiguous, each lane receiving
// exactly its MFMA B fragment.  f32 weights use v_mfma_f32_16x16x4_f32.  bf16 weights use
// v_mfma_f32_16x16x32_bf16 with every f32 activation staged as two bf16 terms (hi = bf16(a), lo = bf16(a - hi))
// and two MFMAs per fragment, so the only rounding relative to f32 arithmetic is the weights' own storage.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <stdexcept>
#include <type_traits>
#include <vector>

#include "kernels.h"

namespace ptts {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kFlowThreads = 512;
constexpr int kFlowWaves = kFlowThreads / 64;
constexpr int kMaxPieces = 4;  // float4 pieces per lane per row: C <= 1024

template <typename WT> struct FlowTraits { static constexpr int pad = 4, kstep = 16, planes = 1; };
template <> struct FlowTraits<__bf16> { static constexpr int pad = 8, kstep = 32, planes = 2; };

// frame index read from device memory, optionally wrapped into a ring
__device__ __forceinline__ long long step_idx(const int* p, int bias, int ring) {
    const int v = *p + bias;
    return ring > 0 ? v % ring : v;
}

__device__ __forceinline__ float silu_dev(float v) { return v / (1.0f + expf(-v)); }

// Sum over the 64 lanes, the same value in every lane.  Four DPP steps (swap within pairs, swap pairs within
// quads, mirror each half row, mirror the row) leave each 16-lane row's sum in all of its lanes; the four row sums
// are then read as scalars.  About a fifth of the latency of six ds_bpermute butterflies.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
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

// acc += A[16][k0 .. k0+kstep) * fragment, A in LDS (plane 0 = value or hi term, plane 1 = lo term)
// ONE (f32 weights, a row group whose only live row is row 0): no matrix instruction.  Each lane multiplies its four
// weights with row 0's four activations and keeps the sum in acc[0]; one_row_finish() then adds the four k-groups
// (lanes l, l+16, l+32, l+48), which leaves row 0's value where the MFMA would have put it (lanes 0..15, element 0).
// A 16x16x4 f32 MFMA takes 32 cycles whatever the number of live rows, and 64 of them per wave were the longest
// part of a single-utterance launch.
template <typename WT, bool ONE = false>
__device__ __forceinline__ void mma_frag(const WT* A0, const WT* A1, int lda, int k0, const uint4 w, f32x4& acc, int lane) {
    const int r = lane & 15, g4 = lane >> 4;
    if constexpr (ONE) {
        static_assert(std::is_same<WT, float>::value, "flow head: the one-row form exists for f32 weights only");
        const float4 a4 = *(const float4*)(A0 + k0 + 4 * g4);
        acc[0] += (a4.x * __uint_as_float(w.x) + a4.y * __uint_as_float(w.y)) + (a4.z * __uint_as_float(w.z) + a4.w * __uint_as_float(w.w));
    } else if constexpr (std::is_same<WT, float>::value) {
        const float4 a4 = *(const float4*)(A0 + r * lda + k0 + 4 * g4);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, __uint_as_float(w.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, __uint_as_float(w.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, __uint_as_float(w.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, __uint_as_float(w.w), acc, 0, 0, 0);
    } else {
        const bf16x8 lo = *(const bf16x8*)(A1 + r * lda + k0 + 8 * g4);
        const bf16x8 hi = *(const bf16x8*)(A0 + r * lda + k0 + 8 * g4);
        const bf16x8 b8 = __builtin_bit_cast(bf16x8, w);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo, b8, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, b8, acc, 0, 0, 0);
    }
}

// acc += A_hi[16][k0 .. k0+32) * (second term of a two-term bf16 weight fragment); issued in front of mma_frag, as the
// persistent form does
template <typename WT>
__device__ __forceinline__ void mma_frag_w2(const WT* A0, int lda, int k0, const uint4 wl, f32x4& acc, int lane) {
    if constexpr (!std::is_same<WT, float>::value) {
        const bf16x8 hi = *(const bf16x8*)(A0 + (lane & 15) * lda + k0 + 8 * (lane >> 4));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, __builtin_bit_cast(bf16x8, wl), acc, 0, 0, 0);
    }
}

template <bool ONE>
__device__ __forceinline__ void one_row_finish(f32x4& acc) {
    if constexpr (ONE) {
        float v = acc[0];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        acc[0] = v;
    }
}

// one activation into the A-operand plane(s)
template <typename WT>
__device__ __forceinline__ void put_act(WT* A0, WT* A1, int idx, float v) {
    if constexpr (std::is_same<WT, float>::value) {
        A0[idx] = v;
    } else {
        const __bf16 hi = (__bf16)v;
        A0[idx] = hi;
        A1[idx] = (__bf16)(v - (float)hi);
    }
}

template <typename WT>
__device__ __forceinline__ void put_act4(WT* A0, WT* A1, int idx, const float4 v) {
    if constexpr (std::is_same<WT, float>::value) {
        *(float4*)(A0 + idx) = v;
    } else {
        bf16x4 hi, lo;
        hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
        lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
        lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
        *(bf16x4*)(A0 + idx) = hi;
        *(bf16x4*)(A1 + idx) = lo;
    }
}

// Row mapping of the element-wise phases: wave w owns rows RPW*w .. RPW*w + RPW-1 of the row group (RPW = 2 covers
// 16 rows with 8 waves, RPW = 1 serves row groups of up to 8 rows); lane l holds the
// float4 pieces l, l + 64, ... of each row.  Everything below has compile-time trip counts and no data-dependent
// branches around loads, so the compiler can put every load of a phase in flight before the first use.
template <int C> struct RowMap {
    static constexpr int PPR = C / 4, NP = (PPR + 63) / 64;
    static __device__ __forceinline__ bool valid(int lane, int i) { return lane + 64 * i < PPR; }
    static __device__ __forceinline__ int col(int lane, int i) { return valid(lane, i) ? 4 * (lane + 64 * i) : 0; }
};

__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }

// h = h_prev + gate * (part[0] + part[1] + ... + part[G-1] + bias), partials added in index order.
// The operands are requested by residual_load and consumed by residual_finish, so that a kernel can put other
// loads (its weight stream) behind them: vmcnt retires in issue order, and what is needed first must be asked first.
template <int C, int G, int RPW = 2> struct ResRegs {
    float4 hv[RPW][RowMap<C>::NP], gt[RPW][RowMap<C>::NP], bs[RowMap<C>::NP], pt[G][RPW][RowMap<C>::NP];
};

template <int C, int G, int RPW = 2>
__device__ __forceinline__ void residual_load(ResRegs<C, G, RPW>& r, const float* h_prev, const float* part, long long part_gs, const float* gate_base,
                                              int mod_width, const float* bias, int row0, int nrows, int wave, int lane) {
    using RM = RowMap<C>;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int lr = RPW * wave + rr;
        const long long grow = row0 + (lr < nrows ? lr : nrows - 1);
#pragma unroll
        for (int i = 0; i < RM::NP; ++i) {
            const int c = RM::col(lane, i);
            r.hv[rr][i] = ld4(h_prev + grow * C + c);
            r.gt[rr][i] = ld4(gate_base + grow * mod_width + c);
#pragma unroll
            for (int g = 0; g < G; ++g) r.pt[g][rr][i] = ld4(part + g * part_gs + grow * C + c);
        }
    }
#pragma unroll
    for (int i = 0; i < RM::NP; ++i) r.bs[i] = ld4(bias + RM::col(lane, i));
}

template <int C, int G, int RPW = 2>
__device__ __forceinline__ void residual_finish(float4 (&h)[RPW][RowMap<C>::NP], const ResRegs<C, G, RPW>& r) {
    using RM = RowMap<C>;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int i = 0; i < RM::NP; ++i) {
            float4 sum = r.pt[0][rr][i];
#pragma unroll
            for (int g = 1; g < G; ++g) { sum.x += r.pt[g][rr][i].x; sum.y += r.pt[g][rr][i].y; sum.z += r.pt[g][rr][i].z; sum.w += r.pt[g][rr][i].w; }
            float4 o = r.hv[rr][i];
            o.x += r.gt[rr][i].x * (sum.x + r.bs[i].x); o.y += r.gt[rr][i].y * (sum.y + r.bs[i].y);
            o.z += r.gt[rr][i].z * (sum.z + r.bs[i].z); o.w += r.gt[rr][i].w * (sum.w + r.bs[i].w);
            h[rr][i] = o;
        }
}

template <int C, int G, int RPW = 2>
__device__ __forceinline__ void residual_update(float4 (&h)[RPW][RowMap<C>::NP], const float* h_prev, const float* part, long long part_gs,
                                                const float* gate_base, int mod_width, const float* bias, int row0, int nrows,
                                                int wave, int lane) {
    ResRegs<C, G, RPW> r;
    residual_load<C, G, RPW>(r, h_prev, part, part_gs, gate_base, mod_width, bias, row0, nrows, wave, lane);
    residual_finish<C, G, RPW>(h, r);
}

// adaLN shift / scale rows and LayerNorm affine vectors of the wave's two rows, requested apart from their use.
template <int C, int RPW = 2> struct ModRegs {
    float4 sh[RPW][RowMap<C>::NP], sc[RPW][RowMap<C>::NP], g4[RowMap<C>::NP], b4[RowMap<C>::NP];
};

template <int C, bool AFFINE, int RPW = 2>
__device__ __forceinline__ void load_mod(ModRegs<C, RPW>& m, const float* ln_g, const float* ln_b, const float* shift_base, const float* scale_base,
                                         int mod_width, int row0, int nrows, int wave, int lane) {
    using RM = RowMap<C>;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int lr = RPW * wave + rr;
        const long long grow = row0 + (lr < nrows ? lr : nrows - 1);
#pragma unroll
        for (int i = 0; i < RM::NP; ++i) {
            m.sh[rr][i] = ld4(shift_base + grow * mod_width + RM::col(lane, i));
            m.sc[rr][i] = ld4(scale_base + grow * mod_width + RM::col(lane, i));
        }
    }
#pragma unroll
    for (int i = 0; i < RM::NP; ++i) {
        if constexpr (AFFINE) { m.g4[i] = ld4(ln_g + RM::col(lane, i)); m.b4[i] = ld4(ln_b + RM::col(lane, i)); }
        else { m.g4[i] = make_float4(1.f, 1.f, 1.f, 1.f); m.b4[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
}

// LayerNorm (affine in m) + adaLN modulate of the wave's two rows -> A-operand planes
template <typename WT, int C, int RPW = 2>
__device__ __forceinline__ void norm_apply(const float4 (&h)[RPW][RowMap<C>::NP], const ModRegs<C, RPW>& m, WT* A0, WT* A1, int lda, float eps,
                                           int wave, int lane) {
    using RM = RowMap<C>;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < RM::NP; ++i)
            if (RM::valid(lane, i)) s += (h[rr][i].x + h[rr][i].y) + (h[rr][i].z + h[rr][i].w);
        const float mean = wsum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < RM::NP; ++i)
            if (RM::valid(lane, i)) {
                const float d0 = h[rr][i].x - mean, d1 = h[rr][i].y - mean, d2 = h[rr][i].z - mean, d3 = h[rr][i].w - mean;
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        const float rstd = 1.0f / sqrtf(wsum(q) / (float)C + eps);
#pragma unroll
        for (int i = 0; i < RM::NP; ++i)
            if (RM::valid(lane, i)) {
                float4 y;
                y.x = ((h[rr][i].x - mean) * rstd * m.g4[i].x + m.b4[i].x) * (1.0f + m.sc[rr][i].x) + m.sh[rr][i].x;
                y.y = ((h[rr][i].y - mean) * rstd * m.g4[i].y + m.b4[i].y) * (1.0f + m.sc[rr][i].y) + m.sh[rr][i].y;
                y.z = ((h[rr][i].z - mean) * rstd * m.g4[i].z + m.b4[i].z) * (1.0f + m.sc[rr][i].z) + m.sh[rr][i].z;
                y.w = ((h[rr][i].w - mean) * rstd * m.g4[i].w + m.b4[i].w) * (1.0f + m.sc[rr][i].w) + m.sh[rr][i].w;
                put_act4<WT>(A0, A1, (RPW * wave + rr) * lda + 4 * (lane + 64 * i), y);
            }
    }
}

template <typename WT, int C, bool AFFINE, int RPW = 2>
__device__ __forceinline__ void norm_modulate(const float4 (&h)[RPW][RowMap<C>::NP], WT* A0, WT* A1, int lda, const float* ln_g,
                                              const float* ln_b, const float* shift_base, const float* scale_base, int mod_width,
                                              float eps, int row0, int nrows, int wave, int lane) {
    ModRegs<C, RPW> m;
    load_mod<C, AFFINE, RPW>(m, ln_g, ln_b, shift_base, scale_base, mod_width, row0, nrows, wave, lane);
    norm_apply<WT, C, RPW>(h, m, A0, A1, lda, eps, wave, lane);
}

struct FlowBlockArgs {
    const float* x_in = nullptr; long long x_stride = 0;   // first block of a step: latent rows feeding the input projection
    const int* x_step = nullptr; long long x_step_stride = 0;          // optional device-side frame index (+ x_step_bias) added to x_in
    int x_step_bias = 0, step_ring = 0;
    const float* h_prev = nullptr; float* h_next = nullptr;           // [B][C]
    const float* part_prev = nullptr; float* part_next = nullptr;     // [G][B][C]
    const float* mods = nullptr;      // this Euler step's adaLN vectors [B][mod_width]
    const float* prev_b2 = nullptr;   // fc2 bias of the previous block
    int first = 0, blk = 0;
    const void* w_in = nullptr; const float* b_in = nullptr;
    const void* w_in_lo = nullptr; const void* w_out_lo = nullptr;   // bf16 weights: second terms of the in / out projections (null: one term)
    const void* w1 = nullptr; const float* b1 = nullptr; const void* w2 = nullptr;
    const float* s1 = nullptr; const float* s2 = nullptr;   // int8 blocks: per-output-column scales of fc1 / fc2
    const float* ln_g = nullptr; const float* ln_b = nullptr;
    int B = 0, mod_width = 0, rpg = 16;   // rpg: utterance rows per workgroup row group (<= 16, the MFMA tile height)
    long long* cycles = nullptr;          // optional [8]: s_memtime of workgroup (0,0) wave 0 at the phase boundaries (tuning aid)
    float eps = 1e-6f;
    // first block of Euler steps 1..: the tail of the previous step (last residual update, final adaLN, output projection,
    // x += dt * v) is done here by every workgroup of the row group instead of in a launch of its own (FIN kernels only)
    const float* fin_h = nullptr; const float* fin_part = nullptr;    // residual stream / partials of the previous step's last block
    const float* fin_mods = nullptr; const float* fin_b2 = nullptr;   // the previous step's adaLN vectors; fc2 bias of its last block
    const float* fin_x = nullptr; long long fin_x_stride = 0;         // x before the previous step
    const int* fin_x_step = nullptr; long long fin_x_step_stride = 0;
    float* fin_x_out = nullptr;                                       // x after it, [B][L] (written by workgroup 0 of the row group)
    const void* w_out = nullptr; const float* b_out = nullptr;
    int blocks = 0; float dt = 0.f;
};

// C = flow_dim, G = column/K split, L = latent_dim; PF: request every weight fragment of the launch up front
// (needs (C/16/G/8 * C/kstep + C/16/8 * C/kstep/G) * 4 VGPRs, so only for bf16 at C <= 512).
template <typename WT, int C, int G, int L, bool PF, int RPW, bool FIN = false, bool ONE = false>
__global__ __launch_bounds__(kFlowThreads) void flow_block_kernel(FlowBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using T = FlowTraits<WT>;
    using RM = RowMap<C>;
    constexpr int KSTEP = T::kstep, PAD = T::pad, PL = T::planes;
    constexpr int NT = C / 16, NT1 = NT / G, KS = C / KSTEP, KS2 = KS / G, CS = C / G, KSL = L / KSTEP;
    constexpr int lda = C + PAD, lda2 = CS + PAD, ldx = L + PAD;
    constexpr int T2 = (NT + kFlowWaves - 1) / kFlowWaves;
    // fc1: K is split over KP1 groups of WPP waves; a wave owns T1 column tiles for its K part and the parts' accumulators
    // meet in LDS.  Every wave then reads only 1/KP1 of the A operand from LDS (reading all of it in all 8 waves was the
    // phase's bound) and runs T1 independent MFMA chains.
    constexpr int KP1 = NT1 >= kFlowWaves ? 2 : kFlowWaves / NT1, WPP = kFlowWaves / KP1, T1 = (NT1 + WPP - 1) / WPP, KSW = KS / KP1;
    static_assert(NT % G == 0 && KS % G == 0 && L % KSTEP == 0, "flow head: split does not tile");
    static_assert(kFlowWaves % KP1 == 0 && KS % KP1 == 0 && (size_t)kFlowWaves * T1 * 256 <= (size_t)16 * C, "flow head: fc1 K split over waves does not tile");
    static_assert(RPW * kFlowWaves >= 16 || RPW == 1, "flow head: rows per wave");
    float* hbuf = (float*)smem_raw;                         // [16][C]   (first block only)
    WT* A0 = (WT*)(hbuf + 16 * C);                          // [PL][16][lda]
    WT* A1 = A0 + (PL - 1) * 16 * lda;
    WT* S0 = A0 + PL * 16 * lda;                            // [PL][16][lda2]  fc1 output slice
    WT* S1 = S0 + (PL - 1) * 16 * lda2;
    WT* X0 = S0 + PL * 16 * lda2;                           // [PL][16][ldx]
    WT* X1 = X0 + (PL - 1) * 16 * ldx;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, row0 = blockIdx.y * a.rpg;
    const int nrows = (a.B - row0) < a.rpg ? (a.B - row0) : a.rpg;
    const int crow = (lane >> 4) * 4, ccol = lane & 15;
    const uint4* w1p = (const uint4*)a.w1 + ((long long)g * NT1 * KS) * 64 + lane;   // + (t*KS + ks)*64
    const uint4* w2p = (const uint4*)a.w2 + ((long long)g * KS2) * 64 + lane;         // + (t*KS + ks2)*64

    if (a.cycles && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) a.cycles[0] = __builtin_readcyclecounter();
    // Order of requests = order of need (vmcnt retires in issue order): the adaLN vectors and the residual operands
    // of the waves that own live rows, then the fc1 fragments, and the fc2 fragments once the residual operands
    // have left their registers.  Waves whose rows lie past the row group's end skip the row work altogether
    // (their A rows stay undefined: MFMA rows are independent and those outputs are never stored).
    const bool live = RPW * wave < nrows;
    const float* mblk = a.mods + a.blk * 3 * C;
    ModRegs<C, RPW> mr;
    ResRegs<C, G, RPW> rres;
    constexpr bool EARLY = C <= 512;   // wider heads do not have the registers to hold everything at once
    if (live) {
        if constexpr (EARLY) load_mod<C, true, RPW>(mr, a.ln_g, a.ln_b, mblk, mblk + C, a.mod_width, row0, nrows, wave, lane);
        if (!a.first)
            residual_load<C, G, RPW>(rres, a.h_prev, a.part_prev, (long long)a.B * C, a.mods + (a.blk - 1) * 3 * C + 2 * C, a.mod_width, a.prev_b2,
                                row0, nrows, wave, lane);
    }
    uint4 wf1[PF ? T1 : 1][PF ? KSW : 1], wf2[PF ? T2 : 1][PF ? KS2 : 1];
    const int kp1 = wave / WPP;                                // this wave's K part of its fc1 tiles
    auto fc1_tile = [&](int i) { return (wave % WPP) + WPP * i; };
    // fc1 fragments in two halves: the first behind the row operands, the second once those have been consumed
    auto request_fc1 = [&](int half) {
        if constexpr (PF) {
#pragma unroll
            for (int i = 0; i < T1; ++i) {
                const int t = fc1_tile(i) < NT1 ? fc1_tile(i) : NT1 - 1;
#pragma unroll
                for (int ks = 0; ks < KSW; ++ks)
                    if ((ks >= KSW / 2) == (half != 0)) wf1[i][ks] = w1p[((long long)t * KS + kp1 * KSW + ks) * 64];
            }
        }
    };
    float bias1[T1];   // fc1 bias of this lane's output columns
#pragma unroll
    for (int i = 0; i < T1; ++i) {
        const int t = fc1_tile(i) < NT1 ? fc1_tile(i) : NT1 - 1;
        bias1[i] = a.b1[g * CS + t * 16 + ccol];
    }

    // ---- residual stream rows of this wave
    float4 h[RPW][RM::NP];
    if (a.first) {
        bool staged = false;
        if constexpr (FIN) {
            if (a.fin_part) {
                // ---- tail of the previous Euler step, redundantly in every workgroup of the row group
                constexpr int NTL = L / 16, KPARTS = kFlowWaves / NTL, KSP = KS / KPARTS;
                static_assert(kFlowWaves % NTL == 0 && KS % KPARTS == 0, "flow head: output projection does not tile");
                const float* mfin = a.fin_mods + a.blocks * 3 * C;
                ModRegs<C, RPW> mf;
                ResRegs<C, G, RPW> rf;
                if (live) {
                    load_mod<C, false, RPW>(mf, nullptr, nullptr, mfin, mfin + C, a.mod_width, row0, nrows, wave, lane);
                    residual_load<C, G, RPW>(rf, a.fin_h, a.fin_part, (long long)a.B * C, a.fin_mods + (a.blocks - 1) * 3 * C + 2 * C, a.mod_width,
                                             a.fin_b2, row0, nrows, wave, lane);
                }
                const int to = wave % NTL, kpo = wave / NTL;
                uint4 wo[KSP], wol[KSP];
                const bool out2 = !std::is_same<WT, float>::value && a.w_out_lo != nullptr;   // uniform
#pragma unroll
                for (int ks = 0; ks < KSP; ++ks) {
                    wo[ks] = ((const uint4*)a.w_out)[((long long)to * KS + kpo * KSP + ks) * 64 + lane];
                    wol[ks] = ((const uint4*)(out2 ? a.w_out_lo : a.w_out))[((long long)to * KS + kpo * KSP + ks) * 64 + lane];
                }
                if (live) {
                    float4 hf[RPW][RM::NP];
                    residual_finish<C, G, RPW>(hf, rf);
                    norm_apply<WT, C, RPW>(hf, mf, A0, A1, lda, a.eps, wave, lane);
                }
                __syncthreads();
                f32x4 acco = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KSP; ++ks) {
                    if (out2) mma_frag_w2<WT>(A0, lda, (kpo * KSP + ks) * KSTEP, wol[ks], acco, lane);
                    mma_frag<WT, ONE>(A0, A1, lda, (kpo * KSP + ks) * KSTEP, wo[ks], acco, lane);
                }
                one_row_finish<ONE>(acco);
                float* redo = hbuf;                            // [waves][256]; hbuf is not in use yet
                *(f32x4*)(redo + wave * 256 + lane * 4) = acco;
                __syncthreads();
                const float* xp = a.fin_x + (a.fin_x_step ? step_idx(a.fin_x_step, a.x_step_bias, a.step_ring) * a.fin_x_step_stride : 0);
                for (int i = tid; i < 16 * L; i += kFlowThreads) {
                    const int r = i / L, c = i % L;
                    float v = 0.f;
                    if (r < nrows) {
                        const int tt = c / 16, ln = (r / 4) * 16 + (c % 16), ri = r % 4;   // accumulator of tile tt: lane ln, element ri
                        float sacc = 0.f;
#pragma unroll
                        for (int k = 0; k < KPARTS; ++k) sacc += redo[(k * NTL + tt) * 256 + ln * 4 + ri];
                        const long long row = row0 + r;
                        v = xp[row * a.fin_x_stride + c] + (sacc + a.b_out[c]) * a.dt;
                        if (g == 0) a.fin_x_out[row * L + c] = v;
                    }
                    put_act<WT>(X0, X1, r * ldx + c, v);
                }
                staged = true;
            }
        }
        if (!staged) {
            const float* xin = a.x_in + (a.x_step ? step_idx(a.x_step, a.x_step_bias, a.step_ring) * a.x_step_stride : 0);
            for (int i = tid; i < 16 * L; i += kFlowThreads) {
                const int r = i / L, c = i % L;
                const float v = r < nrows ? xin[(long long)(row0 + r) * a.x_stride + c] : 0.f;
                put_act<WT>(X0, X1, r * ldx + c, v);
            }
        }
        __syncthreads();
        uint4 wi[T2][KSL], wil[T2][KSL];
        float bin[T2];
        const bool in2 = !std::is_same<WT, float>::value && a.w_in_lo != nullptr;   // uniform
#pragma unroll
        for (int i = 0; i < T2; ++i) {
            const int t = (wave + kFlowWaves * i) < NT ? (wave + kFlowWaves * i) : NT - 1;
#pragma unroll
            for (int ks = 0; ks < KSL; ++ks) {
                wi[i][ks] = ((const uint4*)a.w_in)[((long long)t * KSL + ks) * 64 + lane];
                wil[i][ks] = ((const uint4*)(in2 ? a.w_in_lo : a.w_in))[((long long)t * KSL + ks) * 64 + lane];
            }
            bin[i] = a.b_in[t * 16 + ccol];
        }
        request_fc1(0);
#pragma unroll
        for (int i = 0; i < T2; ++i) {
            const int t = wave + kFlowWaves * i;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSL; ++ks) {
                if (in2) mma_frag_w2<WT>(X0, ldx, ks * KSTEP, wil[i][ks], acc, lane);
                mma_frag<WT, ONE>(X0, X1, ldx, ks * KSTEP, wi[i][ks], acc, lane);
            }
            one_row_finish<ONE>(acc);
            if (t < NT) {
                const int c = t * 16 + ccol;
#pragma unroll
                for (int r = 0; r < 4; ++r) hbuf[(crow + r) * C + c] = acc[r] + bin[i];
            }
        }
        __syncthreads();
        if (live) {
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                for (int i = 0; i < RM::NP; ++i) h[rr][i] = ld4(hbuf + (RPW * wave + rr) * C + RM::col(lane, i));
        }
    } else {
        request_fc1(0);
        if (live) residual_finish<C, G, RPW>(h, rres);
    }
    request_fc1(1);
    if (a.cycles && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) a.cycles[1] = __builtin_readcyclecounter();
    if constexpr (PF) {
#pragma unroll
        for (int i = 0; i < T2; ++i) {
            const int t = (wave + kFlowWaves * i) < NT ? (wave + kFlowWaves * i) : NT - 1;
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) wf2[i][ks] = w2p[((long long)t * KS + ks) * 64];
        }
    }
    if (live) {
        if (g == 0) {   // one workgroup of the row group keeps the stream for the next launch
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                for (int i = 0; i < RM::NP; ++i)
                    if (RPW * wave + rr < nrows && RM::valid(lane, i))
                        *(float4*)(a.h_next + (long long)(row0 + RPW * wave + rr) * C + 4 * (lane + 64 * i)) = h[rr][i];
        }
        if constexpr (!EARLY) load_mod<C, true, RPW>(mr, a.ln_g, a.ln_b, mblk, mblk + C, a.mod_width, row0, nrows, wave, lane);
        norm_apply<WT, C, RPW>(h, mr, A0, A1, lda, a.eps, wave, lane);
    }
    if (a.cycles && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) a.cycles[2] = __builtin_readcyclecounter();
    __syncthreads();
    if (a.cycles && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) a.cycles[3] = __builtin_readcyclecounter();

    // ---- fc1: this workgroup's column slice; this wave's K part of its tiles
    float* red = hbuf;   // [waves][T1][256] accumulators of K parts 1..; hbuf has been read by now
    f32x4 acc1[T1];
#pragma unroll
    for (int i = 0; i < T1; ++i) acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) {
#pragma unroll
        for (int i = 0; i < T1; ++i) {
            const int tc = fc1_tile(i) < NT1 ? fc1_tile(i) : NT1 - 1;
            uint4 w;
            if constexpr (PF) w = wf1[i][ks];
            else w = w1p[((long long)tc * KS + kp1 * KSW + ks) * 64];
            mma_frag<WT, ONE>(A0, A1, lda, (kp1 * KSW + ks) * KSTEP, w, acc1[i], lane);
        }
    }
    if (kp1 > 0) {
#pragma unroll
        for (int i = 0; i < T1; ++i) *(f32x4*)(red + (wave * T1 + i) * 256 + lane * 4) = acc1[i];
    }
    __syncthreads();
    if (kp1 == 0) {
#pragma unroll
        for (int i = 0; i < T1; ++i) {
            const int t = fc1_tile(i);
            f32x4 acc = acc1[i];
#pragma unroll
            for (int k = 1; k < KP1; ++k) {
                const f32x4 o = *(const f32x4*)(red + ((k * WPP + wave) * T1 + i) * 256 + lane * 4);
                acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3];
            }
            one_row_finish<ONE>(acc);
            if (t < NT1) {
                const int cl = t * 16 + ccol;
#pragma unroll
                for (int r = 0; r < 4; ++r) put_act<WT>(S0, S1, (crow + r) * lda2 + cl, silu_dev(acc[r] + bias1[i]));
            }
        }
    }
    if (a.cycles && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) a.cycles[4] = __builtin_readcyclecounter();
    __syncthreads();
    if (a.cycles && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) a.cycles[5] = __builtin_readcyclecounter();

    // ---- fc2: every output column, this workgroup's K-slice -> partial
    float* pout = a.part_next + ((long long)g * a.B + row0) * C;
#pragma unroll
    for (int i = 0; i < T2; ++i) {
        const int t = wave + kFlowWaves * i;
        const int tc = t < NT ? t : NT - 1;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            uint4 w;
            if constexpr (PF) w = wf2[i][ks];
            else w = w2p[((long long)tc * KS + ks) * 64];
            mma_frag<WT, ONE>(S0, S1, lda2, ks * KSTEP, w, acc, lane);
        }
        one_row_finish<ONE>(acc);
        if (t < NT) {
            const int c = t * 16 + ccol;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (crow + r < nrows) pout[(long long)(crow + r) * C + c] = acc[r];
        }
    }
    if (a.cycles && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) a.cycles[6] = __builtin_readcyclecounter();
}

// ---- int8 residual block (v_mfma_i32_16x16x64_i8) ---------------------------------------------------------------
// Same decomposition as flow_block_kernel.  The two matrix inputs (the modulated LayerNorm output and the SiLU
// output) are quantised per row and per block of 64 consecutive k: s = max|a|, q = rint(a * 127 / s), d = s / 127;
// y[n] = (sum over blocks of (float)isum * d) * wscale[n] + bias[n].  The input projection of the first block of a
// step keeps its bf16 weights.
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ unsigned q8_pack4(const float4 v, float inv) {
    const int q0 = (int)rintf(v.x * inv), q1 = (int)rintf(v.y * inv), q2 = (int)rintf(v.z * inv), q3 = (int)rintf(v.w * inv);
    return (unsigned)(q0 & 0xff) | ((unsigned)(q1 & 0xff) << 8) | ((unsigned)(q2 & 0xff) << 16) | ((unsigned)(q3 & 0xff) << 24);
}

// maximum of |v| over the 16 consecutive lanes that hold one 64-wide block (one float4 each)
__device__ __forceinline__ float q8_absmax16(const float4 v) {
    float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    // 16-lane (one DPP row) maximum without touching LDS: swap within pairs, swap pairs within quads, mirror each
    // half row, mirror the row; after each step every lane holds the maximum of a group twice as large
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xF, 0xF, true)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x141, 0xF, 0xF, true)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x140, 0xF, 0xF, true)));
    return m;
}

template <int C, int G, int RPW>
__global__ __launch_bounds__(kFlowThreads) void flow_block_q8_kernel(FlowBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using RM = RowMap<C>;
    constexpr int L = 32, NT = C / 16, NT1 = NT / G, KS = C / 64, KS2 = KS / G, CS = C / G;
    constexpr int lda = C + 16, lda2 = CS + 16, ldx = L + 8;
    constexpr int T1 = (NT1 + kFlowWaves - 1) / kFlowWaves, T2 = (NT + kFlowWaves - 1) / kFlowWaves;
    static_assert(C % 256 == 0 && NT % G == 0 && KS % G == 0, "int8 flow block: flow_dim must be a multiple of 256 and split evenly");
    float* hbuf = (float*)smem_raw;                                 // [16][C]: input-projection result, later the SiLU output slice
    signed char* Aq = (signed char*)(hbuf + 16 * C);                // [16][lda]
    float* dA = (float*)(Aq + 16 * lda);                            // [KS][16]
    signed char* Sq = (signed char*)(dA + KS * 16);                 // [16][lda2]
    float* dS = (float*)(Sq + 16 * lda2);                           // [KS2][16]
    __bf16* X0 = (__bf16*)(dS + KS2 * 16);                          // [2][16][ldx]
    __bf16* X1 = X0 + 16 * ldx;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, row0 = blockIdx.y * a.rpg;
    const int nrows = (a.B - row0) < a.rpg ? (a.B - row0) : a.rpg;
    const int crow = (lane >> 4) * 4, ccol = lane & 15, g4 = lane >> 4;
    const uint4* w1p = (const uint4*)a.w1 + ((long long)g * NT1 * KS) * 64 + lane;
    const uint4* w2p = (const uint4*)a.w2 + ((long long)g * KS2) * 64 + lane;

    // row operands first, the weight fragments behind them (vmcnt retires in issue order)
    const bool live = RPW * wave < nrows;
    const float* mblk = a.mods + a.blk * 3 * C;
    ModRegs<C, RPW> mr;
    ResRegs<C, G, RPW> rres;
    constexpr bool EARLY = C <= 512;   // wider heads do not have the registers to hold everything at once
    if (live && EARLY) {
        load_mod<C, true, RPW>(mr, a.ln_g, a.ln_b, mblk, mblk + C, a.mod_width, row0, nrows, wave, lane);
        if (!a.first)
            residual_load<C, G, RPW>(rres, a.h_prev, a.part_prev, (long long)a.B * C, a.mods + (a.blk - 1) * 3 * C + 2 * C, a.mod_width, a.prev_b2,
                                     row0, nrows, wave, lane);
    }
    uint4 wf1[T1][KS], wf2[T2][KS2];
    float bias1[T1], sc1[T1], sc2[T2];
#pragma unroll
    for (int i = 0; i < T1; ++i) {
        const int t = (wave + kFlowWaves * i) < NT1 ? (wave + kFlowWaves * i) : NT1 - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf1[i][ks] = w1p[((long long)t * KS + ks) * 64];
        bias1[i] = a.b1[g * CS + t * 16 + ccol];
        sc1[i] = a.s1[g * CS + t * 16 + ccol];
    }

    float4 h[RPW][RM::NP];
    if (a.first) {
        const float* xin = a.x_in + (a.x_step ? step_idx(a.x_step, a.x_step_bias, a.step_ring) * a.x_step_stride : 0);
        for (int i = tid; i < 16 * L; i += kFlowThreads) {
            const int r = i / L, c = i % L;
            const float v = r < nrows ? xin[(long long)(row0 + r) * a.x_stride + c] : 0.f;
            put_act<__bf16>(X0, X1, r * ldx + c, v);
        }
        __syncthreads();
        uint4 wi[T2];
        float bin[T2];
#pragma unroll
        for (int i = 0; i < T2; ++i) {
            const int t = (wave + kFlowWaves * i) < NT ? (wave + kFlowWaves * i) : NT - 1;
            wi[i] = ((const uint4*)a.w_in)[(long long)t * 64 + lane];   // L = 32: one bf16 k-step
            bin[i] = a.b_in[t * 16 + ccol];
        }
#pragma unroll
        for (int i = 0; i < T2; ++i) {
            const int t = wave + kFlowWaves * i;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            mma_frag<__bf16>(X0, X1, ldx, 0, wi[i], acc, lane);
            if (t < NT) {
                const int c = t * 16 + ccol;
#pragma unroll
                for (int r = 0; r < 4; ++r) hbuf[(crow + r) * C + c] = acc[r] + bin[i];
            }
        }
        __syncthreads();
        if (live) {
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                for (int i = 0; i < RM::NP; ++i) h[rr][i] = ld4(hbuf + (RPW * wave + rr) * C + RM::col(lane, i));
        }
    } else if (live) {
        if constexpr (!EARLY)
            residual_load<C, G, RPW>(rres, a.h_prev, a.part_prev, (long long)a.B * C, a.mods + (a.blk - 1) * 3 * C + 2 * C, a.mod_width, a.prev_b2,
                                     row0, nrows, wave, lane);
        residual_finish<C, G, RPW>(h, rres);
    }
#pragma unroll
    for (int i = 0; i < T2; ++i) {
        const int t = (wave + kFlowWaves * i) < NT ? (wave + kFlowWaves * i) : NT - 1;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) wf2[i][ks] = w2p[((long long)t * KS + ks) * 64];
        sc2[i] = a.s2[t * 16 + ccol];
    }
    if (live) {
        if (g == 0) {
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                for (int i = 0; i < RM::NP; ++i)
                    if (RPW * wave + rr < nrows) *(float4*)(a.h_next + (long long)(row0 + RPW * wave + rr) * C + 4 * (lane + 64 * i)) = h[rr][i];
        }
        // LayerNorm + adaLN modulate of the wave's rows, quantised into Aq / dA
        if constexpr (!EARLY) load_mod<C, true, RPW>(mr, a.ln_g, a.ln_b, mblk, mblk + C, a.mod_width, row0, nrows, wave, lane);
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < RM::NP; ++i) sum += (h[rr][i].x + h[rr][i].y) + (h[rr][i].z + h[rr][i].w);
            const float mean = wsum(sum) / (float)C;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < RM::NP; ++i) {
                const float d0 = h[rr][i].x - mean, d1 = h[rr][i].y - mean, d2 = h[rr][i].z - mean, d3 = h[rr][i].w - mean;
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            const float rstd = 1.0f / sqrtf(wsum(q) / (float)C + a.eps);
#pragma unroll
            for (int i = 0; i < RM::NP; ++i) {
                float4 y;
                y.x = ((h[rr][i].x - mean) * rstd * mr.g4[i].x + mr.b4[i].x) * (1.0f + mr.sc[rr][i].x) + mr.sh[rr][i].x;
                y.y = ((h[rr][i].y - mean) * rstd * mr.g4[i].y + mr.b4[i].y) * (1.0f + mr.sc[rr][i].y) + mr.sh[rr][i].y;
                y.z = ((h[rr][i].z - mean) * rstd * mr.g4[i].z + mr.b4[i].z) * (1.0f + mr.sc[rr][i].z) + mr.sh[rr][i].z;
                y.w = ((h[rr][i].w - mean) * rstd * mr.g4[i].w + mr.b4[i].w) * (1.0f + mr.sc[rr][i].w) + mr.sh[rr][i].w;
                const float m = q8_absmax16(y);
                const float inv = m > 0.f ? 127.0f / m : 0.f;
                const int piece = lane + 64 * i, lr = RPW * wave + rr;
                *(unsigned*)(Aq + lr * lda + 4 * piece) = q8_pack4(y, inv);
                if ((lane & 15) == 0) dA[(piece >> 4) * 16 + lr] = m / 127.0f;
            }
        }
    }
    __syncthreads();

    // ---- fc1: this workgroup's column slice, full K; SiLU output as f32 into hbuf[16][CS]
#pragma unroll
    for (int i = 0; i < T1; ++i) {
        const int t = wave + kFlowWaves * i;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const i32x4 a4 = *(const i32x4*)(Aq + ccol * lda + ks * 64 + 16 * g4);
            const i32x4 z = {0, 0, 0, 0};
            const i32x4 ci = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, __builtin_bit_cast(i32x4, wf1[i][ks]), z, 0, 0, 0);
            const float4 d4 = *(const float4*)(dA + ks * 16 + crow);
            acc[0] += (float)ci[0] * d4.x; acc[1] += (float)ci[1] * d4.y; acc[2] += (float)ci[2] * d4.z; acc[3] += (float)ci[3] * d4.w;
        }
        if (t < NT1) {
            const int cl = t * 16 + ccol;
#pragma unroll
            for (int r = 0; r < 4; ++r) hbuf[(crow + r) * CS + cl] = silu_dev(acc[r] * sc1[i] + bias1[i]);
        }
    }
    __syncthreads();
    // quantise the SiLU slice into Sq / dS
    for (int id = tid; id < 16 * (CS / 4); id += kFlowThreads) {
        const int row = id / (CS / 4), piece = id % (CS / 4);
        const float4 y = *(const float4*)(hbuf + row * CS + 4 * piece);
        const float m = q8_absmax16(y);
        const float inv = m > 0.f ? 127.0f / m : 0.f;
        *(unsigned*)(Sq + row * lda2 + 4 * piece) = q8_pack4(y, inv);
        if ((piece & 15) == 0) dS[(piece >> 4) * 16 + row] = m / 127.0f;
    }
    __syncthreads();

    // ---- fc2: every output column, this workgroup's K-slice -> partial
    float* pout = a.part_next + ((long long)g * a.B + row0) * C;
#pragma unroll
    for (int i = 0; i < T2; ++i) {
        const int t = wave + kFlowWaves * i;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            const i32x4 a4 = *(const i32x4*)(Sq + ccol * lda2 + ks * 64 + 16 * g4);
            const i32x4 z = {0, 0, 0, 0};
            const i32x4 ci = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, __builtin_bit_cast(i32x4, wf2[i][ks]), z, 0, 0, 0);
            const float4 d4 = *(const float4*)(dS + ks * 16 + crow);
            acc[0] += (float)ci[0] * d4.x; acc[1] += (float)ci[1] * d4.y; acc[2] += (float)ci[2] * d4.z; acc[3] += (float)ci[3] * d4.w;
        }
        if (t < NT) {
            const int c = t * 16 + ccol;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (crow + r < nrows) pout[(long long)(crow + r) * C + c] = acc[r] * sc2[i];
        }
    }
}

struct FlowFinalArgs {
    const float* h_prev = nullptr; const float* part_prev = nullptr;   // of the last block
    const float* mods = nullptr; const float* prev_b2 = nullptr;
    const float* x_in = nullptr; long long x_stride = 0;
    const int* x_step = nullptr; long long x_step_stride = 0; int x_step_bias = 0, step_ring = 0;
    float* x_out = nullptr;                                            // [B][L]
    float* latents = nullptr; const int* lat_step = nullptr; int max_frames = 0;   // optional: also file x_out under frame *lat_step + x_step_bias
    const void* w_out = nullptr; const float* b_out = nullptr;
    const void* w_out_lo = nullptr;   // bf16 weights: second term of the output projection (null: one term)
    int B = 0, blocks = 0, mod_width = 0, rpg = 16;
    float eps = 1e-6f, dt = 0.f;
};

// Last residual update, final adaLN, output projection (K split over the waves, summed through LDS in wave
// order), x_out = x_in + dt * v.
template <typename WT, int C, int G, int L, int RPW, bool ONE = false>
__global__ __launch_bounds__(kFlowThreads) void flow_final_kernel(FlowFinalArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using T = FlowTraits<WT>;
    using RM = RowMap<C>;
    constexpr int KSTEP = T::kstep, PAD = T::pad, PL = T::planes;
    constexpr int lda = C + PAD, KS = C / KSTEP, NTL = L / 16;
    constexpr int KPARTS = kFlowWaves / NTL, KSP = KS / KPARTS;   // waves per output tile, k-steps per wave
    static_assert(kFlowWaves % NTL == 0 && KS % KPARTS == 0, "flow head: output projection does not tile");
    WT* A0 = (WT*)smem_raw;
    WT* A1 = A0 + (PL - 1) * 16 * lda;
    float* red = (float*)(A0 + PL * 16 * lda);   // [waves][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * a.rpg;
    const int nrows = (a.B - row0) < a.rpg ? (a.B - row0) : a.rpg;
    const int t = wave % NTL, kp = wave / NTL;
    // row operands first, the projection fragments behind them (vmcnt retires in issue order)
    const bool live = RPW * wave < nrows;
    const float* mfin = a.mods + a.blocks * 3 * C;
    ModRegs<C, RPW> mr;
    ResRegs<C, G, RPW> rres;
    constexpr bool EARLY = C <= 512;   // wider heads do not have the registers to hold everything at once
    if (live) {
        if constexpr (EARLY) load_mod<C, false, RPW>(mr, nullptr, nullptr, mfin, mfin + C, a.mod_width, row0, nrows, wave, lane);
        residual_load<C, G, RPW>(rres, a.h_prev, a.part_prev, (long long)a.B * C, a.mods + (a.blocks - 1) * 3 * C + 2 * C, a.mod_width, a.prev_b2,
                                 row0, nrows, wave, lane);
    }
    uint4 wo[KSP], wol[KSP];
    const bool out2 = !std::is_same<WT, float>::value && a.w_out_lo != nullptr;   // uniform
#pragma unroll
    for (int ks = 0; ks < KSP; ++ks) {
        wo[ks] = ((const uint4*)a.w_out)[((long long)t * KS + kp * KSP + ks) * 64 + lane];
        wol[ks] = ((const uint4*)(out2 ? a.w_out_lo : a.w_out))[((long long)t * KS + kp * KSP + ks) * 64 + lane];
    }
    if (live) {
        float4 h[RPW][RM::NP];
        residual_finish<C, G, RPW>(h, rres);
        if constexpr (!EARLY) load_mod<C, false, RPW>(mr, nullptr, nullptr, mfin, mfin + C, a.mod_width, row0, nrows, wave, lane);
        norm_apply<WT, C, RPW>(h, mr, A0, A1, lda, a.eps, wave, lane);
    }
    __syncthreads();
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSP; ++ks) {
        if (out2) mma_frag_w2<WT>(A0, lda, (kp * KSP + ks) * KSTEP, wol[ks], acc, lane);
        mma_frag<WT, ONE>(A0, A1, lda, (kp * KSP + ks) * KSTEP, wo[ks], acc, lane);
    }
    one_row_finish<ONE>(acc);
    *(f32x4*)(red + wave * 256 + lane * 4) = acc;
    __syncthreads();
    const float* xin = a.x_in + (a.x_step ? step_idx(a.x_step, a.x_step_bias, a.step_ring) * a.x_step_stride : 0);
    for (int i = tid; i < 16 * L; i += kFlowThreads) {
        const int r = i / L, c = i % L;
        if (r >= nrows) continue;
        const int tt = c / 16, ln = (r / 4) * 16 + (c % 16), ri = r % 4;   // accumulator of tile tt: lane ln, element ri
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < KPARTS; ++k) v += red[(k * NTL + tt) * 256 + ln * 4 + ri];
        const long long row = row0 + r;
        const float xn = xin[row * a.x_stride + c] + (v + a.b_out[c]) * a.dt;
        a.x_out[row * L + c] = xn;
        if (a.latents) a.latents[(row * a.max_frames + step_idx(a.lat_step, a.x_step_bias, a.step_ring)) * L + c] = xn;
    }
}

// Reorders W[N][K] (row major) into the fragment-major layout the kernels stream: tile (nt, ks) holds, for
// lane l, the `per` consecutive k-values ks*kstep + per*(l>>4) .. of output column nt*16 + (l&15).
std::vector<unsigned char> pack_flow_weight(const float* W, int N, int K, int wtype) {
    const bool bf = wtype == W_BF16;
    const int kstep = bf ? 32 : 16, per = bf ? 8 : 4, esz = bf ? 2 : 4;
    if (N % 16 || K % kstep) throw std::runtime_error("flow head: weight shape not tileable");
    std::vector<unsigned char> out((size_t)N * K * esz);
    const int KS = K / kstep;
    for (int nt = 0; nt < N / 16; ++nt)
        for (int ks = 0; ks < KS; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < per; ++j) {
                    const float v = W[(size_t)(nt * 16 + (l & 15)) * K + ks * kstep + per * (l >> 4) + j];
                    const size_t idx = (((size_t)nt * KS + ks) * 64 + l) * per + j;
                    if (bf) {
                        uint32_t u;
                        std::memcpy(&u, &v, 4);
                        u += 0x7fffu + ((u >> 16) & 1u);
                        uint16_t hv = (uint16_t)(u >> 16);
                        std::memcpy(out.data() + idx * 2, &hv, 2);
                    } else {
                        std::memcpy(out.data() + idx * 4, &v, 4);
                    }
                }
    return out;
}

constexpr int kMaxFlowSplit = 8;

// Decomposition of one residual-block launch: G workgroups per row group (column slice of fc1 / K-slice of fc2)
// and rpg utterance rows per row group.  Smaller row groups and more slices mean more workgroups with fewer
// bytes each (the launch is bounded by what one CU can pull in), until the partial-sum exchange (G x rows x C
// floats per workgroup) outweighs the weight slice.  FLOW_HEAD_SPLIT / FLOW_HEAD_ROWS override for experiments.
static void flow_head_shape(int C, int B, bool bf, int& G, int& rpg) {
    G = 4; rpg = 16;
    if (bf && C == 512 && B > 4) rpg = 4;   // measured at B = 64: 4 x 16 -> 627 us per chain, 4 x 4 -> 578 us, 8 x any -> 670+ us
    if (!bf && C == 512 && B <= 8) G = 8;   // a streaming call's single row: 877 us per chain with 4 slices of float weights, 375 us with 8 and no MFMA
    if (const char* e = std::getenv("FLOW_HEAD_SPLIT")) G = std::atoi(e) == 8 && C == 512 ? 8 : 4;
    if (const char* e = std::getenv("FLOW_HEAD_ROWS")) { const int r = std::atoi(e); if (r == 1 || r == 2 || r == 4 || r == 8 || r == 16) rpg = r; }
}

size_t flow_head_workspace_bytes(int B, int C, int wtype, int blocks) {
    const size_t chain = ((size_t)2 * B * C + (size_t)2 * kMaxFlowSplit * B * C + (size_t)2 * B * 32) * sizeof(float);   // h x2, partials x2, x x2
    const size_t ring = C == 512 ? flow_ring_workspace_bytes(B, blocks) : 0;   // the persistent form keeps one set per block
    return chain > ring ? chain : ring;
}

// hipFuncSetAttribute is per device and per kernel; remember where it has been applied.
static void set_lds_limit(const void* kernel, size_t smem) {
    if (smem <= 64 * 1024) return;
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    if (done.insert({kernel, dev}).second)
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// FUSE: the tail of Euler step j-1 runs inside the first block launch of step j (one launch less per step)
template <typename WT, int C, bool PF, int G = 4, int RPW = 2, bool FUSE = false, bool ONE = false>
static void run_chain(const FlowHeadArgs& a, hipStream_t s, int rpg) {
    constexpr int L = 32;
    using T = FlowTraits<WT>;
    const int B = a.B, RG = (B + rpg - 1) / rpg;
    const size_t smem_block = (size_t)16 * C * 4 + (size_t)T::planes * 16 * ((C + T::pad) + (C / G + T::pad) + (L + T::pad)) * sizeof(WT);
    const size_t smem_final = (size_t)T::planes * 16 * (C + T::pad) * sizeof(WT) + (size_t)kFlowWaves * 256 * 4;
    auto kb = flow_block_kernel<WT, C, G, L, PF, RPW, false, ONE>;
    auto kbf = flow_block_kernel<WT, C, G, L, PF, RPW, FUSE, ONE>;
    auto kf = flow_final_kernel<WT, C, G, L, RPW, ONE>;
    if (ONE && a.B != 1) throw std::runtime_error("flow head: the one-row form serves a single row");
    if (rpg > RPW * kFlowWaves) throw std::runtime_error("flow head: row group larger than the waves can cover");
    set_lds_limit((const void*)kb, smem_block);
    set_lds_limit((const void*)kbf, smem_block);
    set_lds_limit((const void*)kf, smem_final);
    const bool q8 = a.block_wtype == W_I8;
    size_t smem_q8 = 0;
    void (*kq)(FlowBlockArgs) = nullptr;
    if constexpr (std::is_same<WT, __bf16>::value && C % 256 == 0) {
        if (q8) {
            kq = flow_block_q8_kernel<C, G, RPW>;
            smem_q8 = (size_t)16 * C * 4 + (size_t)16 * (C + 16) + (size_t)(C / 64) * 16 * 4 + (size_t)16 * (C / G + 16) + (size_t)(C / 64 / G) * 16 * 4 +
                      (size_t)2 * 16 * (L + 8) * 2;
            set_lds_limit((const void*)kq, smem_q8);
        }
    }
    if (q8 && !kq) throw std::runtime_error("flow head: int8 blocks need bf16 in/out weights and flow_dim % 256 == 0");
    float* hb[2] = {a.workspace, a.workspace + (size_t)B * C};
    float* pb[2] = {a.workspace + (size_t)2 * B * C, a.workspace + (size_t)2 * B * C + (size_t)kMaxFlowSplit * B * C};
    const FlowHostTabs& tabs = *a.host_tabs;
    int cur = 0;
    if (a.probe_a) (void)hipEventRecord(a.probe_a, s);   // the probe brackets the whole chain: steps * (blocks + 1) launches
    const bool fuse = FUSE && !q8;
    // bf16 matrices whose in / out projections carry second terms (not the int8 blocks' kernel, not the float chain)
    const bool io2 = std::is_same<WT, __bf16>::value && !q8 && a.wtype == W_BF16 && tabs.r_in_lo && tabs.r_out_lo;
    float* xb[2] = {pb[1] + (size_t)kMaxFlowSplit * B * C, pb[1] + (size_t)kMaxFlowSplit * B * C + (size_t)B * L};   // x after odd / even steps
    for (int j = 0; j < a.steps; ++j) {
        const float* mods_j = a.mods + (size_t)j * B * a.mod_width;
        // x before step j: the caller's start point, then (unfused) the output buffer, (fused) the ping-pong pair
        const float* x_in = j == 0 ? a.x0 : (fuse ? xb[j & 1] : a.x_out);
        const long long x_stride = j == 0 ? a.x0_stride : (long long)L;
        for (int blk = 0; blk < a.blocks; ++blk) {
            FlowBlockArgs fb;
            fb.x_in = x_in; fb.x_stride = x_stride;
            if (j == 0) { fb.x_step = a.x0_step; fb.x_step_stride = a.x0_step_stride; }
            fb.x_step_bias = a.x0_step_bias; fb.step_ring = a.step_ring;
            fb.h_prev = hb[cur ^ 1]; fb.h_next = hb[cur]; fb.part_prev = pb[cur ^ 1]; fb.part_next = pb[cur];
            fb.mods = mods_j; fb.prev_b2 = blk ? tabs.b_fc2[blk - 1] : nullptr;
            fb.first = blk == 0; fb.blk = blk;
            fb.w_in = a.w_in; fb.b_in = a.b_in; fb.w1 = tabs.w_fc1[blk]; fb.b1 = tabs.b_fc1[blk]; fb.w2 = tabs.w_fc2[blk];
            if (io2) { fb.w_in_lo = tabs.r_in_lo; fb.w_out_lo = tabs.r_out_lo; }
            fb.ln_g = tabs.ln_g[blk]; fb.ln_b = tabs.ln_b[blk];
            if (q8) { fb.s1 = tabs.s_fc1[blk]; fb.s2 = tabs.s_fc2[blk]; }
            fb.B = B; fb.mod_width = a.mod_width; fb.eps = a.eps; fb.rpg = rpg;
            if (a.cycle_probe && j == 1 && blk == 2) fb.cycles = a.cycle_probe;
            const bool fin = fuse && blk == 0 && j > 0;
            if (fin) {   // x_j = x_{j-1} + dt * v_{j-1} from the previous step's last block, then this step's first block
                fb.fin_h = hb[cur ^ 1]; fb.fin_part = pb[cur ^ 1]; fb.fin_mods = a.mods + (size_t)(j - 1) * B * a.mod_width;
                fb.fin_b2 = tabs.b_fc2[a.blocks - 1];
                fb.fin_x = j == 1 ? a.x0 : xb[(j - 1) & 1]; fb.fin_x_stride = j == 1 ? a.x0_stride : (long long)L;
                if (j == 1) { fb.fin_x_step = a.x0_step; fb.fin_x_step_stride = a.x0_step_stride; }
                fb.fin_x_out = xb[j & 1];
                fb.w_out = a.w_out; fb.b_out = a.b_out; fb.blocks = a.blocks; fb.dt = 1.0f / (float)a.steps;
            }
            if (q8) hipLaunchKernelGGL(kq, dim3(G, RG), dim3(kFlowThreads), smem_q8, s, fb);
            else if (fin) hipLaunchKernelGGL(kbf, dim3(G, RG), dim3(kFlowThreads), smem_block, s, fb);
            else hipLaunchKernelGGL(kb, dim3(G, RG), dim3(kFlowThreads), smem_block, s, fb);
            cur ^= 1;
        }
        if (fuse && j + 1 < a.steps) continue;   // this step's tail runs in the next step's first launch
        FlowFinalArgs ff;
        ff.h_prev = hb[cur ^ 1]; ff.part_prev = pb[cur ^ 1]; ff.mods = mods_j; ff.prev_b2 = tabs.b_fc2[a.blocks - 1];
        ff.x_in = x_in; ff.x_stride = x_stride; ff.x_out = a.x_out;
        if (j == 0) { ff.x_step = a.x0_step; ff.x_step_stride = a.x0_step_stride; } ff.w_out = a.w_out; ff.b_out = a.b_out;
        if (io2) ff.w_out_lo = tabs.r_out_lo;
        ff.x_step_bias = a.x0_step_bias; ff.step_ring = a.step_ring;
        if (j + 1 == a.steps && a.latents) { ff.latents = a.latents; ff.lat_step = a.x0_step; ff.max_frames = a.max_frames; }
        ff.B = B; ff.blocks = a.blocks; ff.mod_width = a.mod_width; ff.eps = a.eps; ff.dt = 1.0f / (float)a.steps; ff.rpg = rpg;
        hipLaunchKernelGGL(kf, dim3(RG), dim3(kFlowThreads), smem_final, s, ff);
    }
    if (a.probe_b) (void)hipEventRecord(a.probe_b, s);
}

// Raises the dynamic-LDS limit of every kernel variant on the current device; call once per device outside
// any stream capture.
template <typename WT, int C, bool PF, int G = 4, int RPW = 2>
static void prepare_chain() {
    constexpr int L = 32;
    set_lds_limit((const void*)flow_block_kernel<WT, C, G, L, PF, RPW>, 128 * 1024);
    set_lds_limit((const void*)flow_final_kernel<WT, C, G, L, RPW>, 128 * 1024);
}
void flow_head_prepare(int C, int wtype) {
    const bool bf = wtype == W_BF16;
    if (C == 512) flow_ring_prepare();
    switch (C) {
        case 128: bf ? prepare_chain<__bf16, 128, true>() : prepare_chain<float, 128, false>(); break;
        case 256: bf ? prepare_chain<__bf16, 256, true>() : prepare_chain<float, 256, false>(); break;
        case 512:
            if (bf) {
                prepare_chain<__bf16, 512, true>(); prepare_chain<__bf16, 512, true, 4, 1>(); prepare_chain<__bf16, 512, true, 8, 1>();
                set_lds_limit((const void*)flow_block_kernel<__bf16, 512, 4, 32, true, 1, true>, 128 * 1024);
            }
            else {
                prepare_chain<float, 512, false>(); prepare_chain<float, 512, false, 8, 1>();
                set_lds_limit((const void*)flow_block_kernel<float, 512, 8, 32, false, 1, false, true>, 128 * 1024);
                set_lds_limit((const void*)flow_block_kernel<float, 512, 8, 32, false, 1, true, true>, 128 * 1024);
                set_lds_limit((const void*)flow_final_kernel<float, 512, 8, 32, 1, true>, 128 * 1024);
            }
            break;
        case 1024: bf ? prepare_chain<__bf16, 1024, false>() : prepare_chain<float, 1024, false>(); break;
        default: throw std::runtime_error("flow head: flow_dim must be 128, 256, 512 or 1024");
    }
}

// Launches one call of launch_flow_head makes: steps * blocks block launches plus one tail launch per step, or one
// tail launch in all where the tails ride in the next step's first block launch.
int flow_head_launch_count(const FlowHeadArgs& a) {
    if (flow_ring_supported(a)) return 1;
    const bool bf = a.wtype == W_BF16, q8 = a.block_wtype == W_I8;
    int G = 4, rpg = 16;
    flow_head_shape(a.C, a.B, bf, G, rpg);
    const bool fused = a.C == 512 && !std::getenv("FLOW_HEAD_NO_FUSE") &&
                       (bf ? (!q8 && G != 8 && rpg <= 8) : (G == 8 && a.B == 1 && !std::getenv("FLOW_HEAD_NO_ONE_ROW")));
    return a.steps * a.blocks + (fused ? 1 : a.steps);
}

void launch_flow_head(const FlowHeadArgs& a, hipStream_t s) {
    if (a.B <= 0) return;
    if (a.L != 32 || a.steps <= 0 || a.blocks <= 0) throw std::runtime_error("flow head: latent_dim must be 32");
    if (!a.workspace || !a.host_tabs) throw std::runtime_error("flow head: workspace and host pointer tables are required");
    if (flow_ring_supported(a)) { launch_flow_ring(a, s); return; }
    const bool bf = a.wtype == W_BF16;
    const bool q8 = a.block_wtype == W_I8;
    int G = 4, rpg = 16;
    flow_head_shape(a.C, a.B, bf, G, rpg);
    if (rpg < 4 && !(bf && a.C == 512)) rpg = 4;   // row groups under 4 rows exist only for the one-row-per-wave bf16 variants
    switch (a.C) {
        case 128: bf ? run_chain<__bf16, 128, true>(a, s, rpg) : run_chain<float, 128, false>(a, s, rpg); break;
        case 256: bf ? run_chain<__bf16, 256, true>(a, s, rpg) : run_chain<float, 256, false>(a, s, rpg); break;
        case 512:
            if (bf && !q8 && G == 8 && rpg <= 8) run_chain<__bf16, 512, true, 8, 1>(a, s, rpg);
            else if (bf && rpg <= 8 && !std::getenv("FLOW_HEAD_NO_FUSE")) run_chain<__bf16, 512, true, 4, 1, true>(a, s, rpg);
            else if (bf && rpg <= 8) run_chain<__bf16, 512, true, 4, 1>(a, s, rpg);
            else if (bf) run_chain<__bf16, 512, true>(a, s, rpg);
            else if (!bf && G == 8 && a.B == 1 && !std::getenv("FLOW_HEAD_NO_ONE_ROW") && !std::getenv("FLOW_HEAD_NO_FUSE"))
                run_chain<float, 512, false, 8, 1, true, true>(a, s, 4);
            else if (!bf && G == 8 && a.B == 1 && !std::getenv("FLOW_HEAD_NO_ONE_ROW")) run_chain<float, 512, false, 8, 1, false, true>(a, s, 4);
            else if (!bf && G == 8) run_chain<float, 512, false, 8, 1>(a, s, 8);   // one row per wave: two would spill (8 partials per row)
            else run_chain<float, 512, false>(a, s, rpg);
            break;
        case 1024: bf ? run_chain<__bf16, 1024, false>(a, s, rpg) : run_chain<float, 1024, false>(a, s, rpg); break;
        default: throw std::runtime_error("flow head: flow_dim must be 128, 256, 512 or 1024");
    }
}

}  // namespace ptts
