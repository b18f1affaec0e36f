// Kernels of the batched autoregressive LM step (one new token per utterance, M = batch rows <= a few
// hundred).  At this shape every matrix product is a weight-streaming problem, so the three kernels are
// built around that:
//
//  skinny_gemm   M x N x K with M tiled by 64: a workgroup of 4 waves owns 64 output columns and a K-slice;
//                wave w owns column tile w for all row tiles.  Weights are fragment-major (one wave-level
//                load = 1 KiB contiguous = one MFMA B operand per lane) and go L2 -> VGPR one K-chunk ahead
//                of use; activations (always f32 in memory) go through a double-buffered LDS tile shared
//                by the 4 waves.  Long K is split over workgroups into f32 partial slabs.
//  add_norm      the consumer of those slabs: per row, sums the K-slices in a fixed order (bitwise
//                reproducible), applies layer scale and the residual, then LayerNorms the updated row for
//                the next matrix product (and, for the last layer, emits conditioning + EOS logit).
//  attn_decode   RoPE, KV-cache append and single-query attention for one (row, head) per wave.
//
// f32 weights use v_mfma_f32_16x16x4_f32 (exact f32 FMA chain), bf16 weights v_mfma_f32_16x16x32_bf16.
// With bf16 weights an f32 activation a is staged as two bf16 terms, hi = bf16(a) and lo = bf16(a - hi),
// and each weight fragment is multiplied with both: the product then carries the activation to ~16
// mantissa bits, so the only rounding left relative to f32 arithmetic is the weights' own bf16 storage.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <stdexcept>
#include <type_traits>
#include <vector>

#include "kernels.h"

namespace ptts {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <typename CT> struct SkinnyTraits { static constexpr int kstep = 16, per = 4, kch = 128, pad = 4; };
template <> struct SkinnyTraits<__bf16> { static constexpr int kstep = 32, per = 8, kch = 128, pad = 8; };

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

// v + (v of the lane the DPP control selects): 0xB1 / 0x4E swap within pairs / quads, 0x141 / 0x140 mirror a half row / row,
// 0x120 + n rotates the 16-lane row right by n
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// a -> (hi, lo) with hi + lo == a to ~16 mantissa bits
__device__ __forceinline__ void split_bf16(const float4 v, bf16x4& hi, bf16x4& lo) {
    hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
    lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
    lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
}

__device__ __forceinline__ float act_dev(float v, int act) {
    switch (act) {
        case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        case ACT_SILU: return v / (1.0f + expf(-v));
        case ACT_ELU: return v > 0.f ? v : expm1f(v);
        default: return v;
    }
}

// W2 (bf16 weights only): the matrix comes as two bf16 planes, W = Wp + Wp2 with Wp2 = bf16(W - Wp), and every fragment takes
// a third MFMA (hi_a * lo_w): the product then carries both operands to ~16 mantissa bits.
template <typename CT, int MT, int KCH, bool W2 = false>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(SkinnyArgs a) {
    using T = SkinnyTraits<CT>;
    static_assert(!W2 || !std::is_same<CT, float>::value, "skinny gemm: a second weight plane exists for bf16 matrices only");
    constexpr int ROWS = 16 * MT;           // rows per workgroup
    constexpr bool kBf16 = !std::is_same<CT, float>::value;
    constexpr int KSTEP = T::kstep, LDA = KCH + T::pad;
    constexpr int NKS = KCH / KSTEP;        // k-steps per chunk
    constexpr int PIECES = KCH / 4;         // float4 pieces per row of a chunk (32)
    constexpr int APT = ROWS * PIECES / 256;  // A pieces per thread per chunk (8 at 64 rows)
    constexpr int PLANES = kBf16 ? 2 : 1;   // bf16: hi and lo planes
    __shared__ __attribute__((aligned(16))) CT As[2][PLANES][ROWS * LDA];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x * 4 + wave;
    const bool tile_live = nt * 16 < a.N;
    const int m0 = blockIdx.z * ROWS;
    const int Kc = a.K / a.splits;
    const int k_begin = blockIdx.y * Kc;
    const int nchunks = (Kc + KCH - 1) / KCH;
    const int KS_total = a.K / KSTEP;
    const uint4* wbase = (const uint4*)a.Wp + (long long)nt * KS_total * 64 + lane;
    const uint4* wbase2 = (const uint4*)(W2 ? a.Wp2 : a.Wp) + (long long)nt * KS_total * 64 + lane;

    float4 areg[APT];
    uint4 bnext[NKS], bcur[NKS];
    uint4 bnext2[W2 ? NKS : 1], bcur2[W2 ? NKS : 1];
    auto load_chunk = [&](int c) {
        const int kc0 = c * KCH;  // offset inside the slice
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int id = tid + 256 * i, row = id / PIECES, piece = id % PIECES;
            const int k = kc0 + piece * 4;
            const bool ok = (m0 + row) < a.M && k < Kc;
            areg[i] = ok ? *(const float4*)(a.A + (long long)(m0 + row) * a.lda + k_begin + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NKS; ++i) {
            const int k = kc0 + i * KSTEP;
            bnext[i] = (tile_live && k < Kc) ? wbase[(long long)((k_begin + k) / KSTEP) * 64] : make_uint4(0u, 0u, 0u, 0u);
            if constexpr (W2) bnext2[i] = (tile_live && k < Kc) ? wbase2[(long long)((k_begin + k) / KSTEP) * 64] : make_uint4(0u, 0u, 0u, 0u);
        }
    };

    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, g = lane >> 4;
    // the epilogue's bias is requested with the first operands, not after the last product
    const float bias_col = (a.bias && !a.P && tile_live) ? a.bias[nt * 16 + r] : 0.f;
    const float cs_col = (a.colscale && !a.P && tile_live) ? a.colscale[nt * 16 + r] : 1.f;

    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        CT* buf = As[c & 1][0];
        CT* buf_lo = As[c & 1][PLANES - 1];
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int id = tid + 256 * i, row = id / PIECES, piece = id % PIECES;
            if constexpr (kBf16) {
                bf16x4 hi, lo;
                split_bf16(areg[i], hi, lo);
                *(bf16x4*)(buf + row * LDA + piece * 4) = hi;
                *(bf16x4*)(buf_lo + row * LDA + piece * 4) = lo;
            } else {
                *(float4*)(buf + row * LDA + piece * 4) = areg[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NKS; ++i) bcur[i] = bnext[i];
        if constexpr (W2) {
#pragma unroll
            for (int i = 0; i < NKS; ++i) bcur2[i] = bnext2[i];
        }
        __syncthreads();
        if (c + 1 < nchunks) load_chunk(c + 1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if constexpr (!kBf16) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float4 a4 = *(const float4*)(buf + (mt * 16 + r) * LDA + ks * 16 + 4 * g);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, __uint_as_float(bcur[ks].x), acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, __uint_as_float(bcur[ks].y), acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, __uint_as_float(bcur[ks].z), acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, __uint_as_float(bcur[ks].w), acc[mt], 0, 0, 0);
                }
            } else {
                const bf16x8 b8 = __builtin_bit_cast(bf16x8, bcur[ks]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x8 lo8 = *(const bf16x8*)(buf_lo + (mt * 16 + r) * LDA + ks * 32 + 8 * g);
                    const bf16x8 hi8 = *(const bf16x8*)(buf + (mt * 16 + r) * LDA + ks * 32 + 8 * g);
                    if constexpr (W2) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi8, __builtin_bit_cast(bf16x8, bcur2[ks]), acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo8, b8, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi8, b8, acc[mt], 0, 0, 0);
                }
            }
        }
    }
    if (!tile_live) return;
    const int col = nt * 16 + r;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + mt * 16 + g * 4 + i;
            if (row >= a.M) continue;
            float v = acc[mt][i];
            if (a.P) {
                a.P[((long long)blockIdx.y * a.M + row) * a.N + col] = v;
            } else {
                v = act_dev(v + bias_col, a.act);
                if (a.colscale) v *= cs_col;
                if (a.R) v += a.R[(long long)row * a.ldr + col];
                a.C[(long long)row * a.ldc + col] = v;
            }
        }
    }
}

// ---- int8 weights x int8 activations (v_mfma_i32_16x16x64_i8) --------------------------------------------------
// Activations are quantised while they are staged: per row and per block of 64 consecutive k,
// s = max|a|, q = rint(a * 127 / s), d = s / 127; the integer block products are exact, and
// y[n] = (sum over blocks of (float)isum * d) * wscale[n] + bias[n].  A block is held by 16 consecutive lanes
// (one float4 each), so its maximum is four shuffles.
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ unsigned quantize4(const float4 v, float inv) {
    const int q0 = (int)rintf(v.x * inv), q1 = (int)rintf(v.y * inv), q2 = (int)rintf(v.z * inv), q3 = (int)rintf(v.w * inv);
    return (unsigned)(q0 & 0xff) | ((unsigned)(q1 & 0xff) << 8) | ((unsigned)(q2 & 0xff) << 16) | ((unsigned)(q3 & 0xff) << 24);
}

__device__ __forceinline__ float block_absmax16(const float4 v) {
    float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    // 16-lane (one DPP row) maximum without touching LDS: swap within pairs, swap pairs within quads, mirror each
    // half row, mirror the row; after each step every lane holds the maximum of a group twice as large
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xF, 0xF, true)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x141, 0xF, 0xF, true)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x140, 0xF, 0xF, true)));
    return m;
}

template <int MT>
__global__ __launch_bounds__(256) void skinny_q8_kernel(SkinnyArgs a) {
    constexpr int ROWS = 16 * MT;
    constexpr int KSTEP = 64, KCH = 128, LDA = KCH + 16, NKS = KCH / KSTEP, PIECES = KCH / 4, APT = ROWS * PIECES / 256;
    __shared__ __attribute__((aligned(16))) signed char As[2][ROWS * LDA];
    __shared__ __attribute__((aligned(16))) float Dq[2][NKS][ROWS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x * 4 + wave;
    const bool tile_live = nt * 16 < a.N;
    const int m0 = blockIdx.z * ROWS;
    const int Kc = a.K / a.splits;
    const int k_begin = blockIdx.y * Kc;
    const int nchunks = (Kc + KCH - 1) / KCH;
    const int KS_total = a.K / KSTEP;
    const uint4* wbase = (const uint4*)a.Wp + (long long)nt * KS_total * 64 + lane;

    float4 areg[APT];
    uint4 bnext[NKS], bcur[NKS];
    auto load_chunk = [&](int c) {
        const int kc0 = c * KCH;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int id = tid + 256 * i, row = id / PIECES, piece = id % PIECES;
            const int k = kc0 + piece * 4;
            const bool ok = (m0 + row) < a.M && k < Kc;
            areg[i] = ok ? *(const float4*)(a.A + (long long)(m0 + row) * a.lda + k_begin + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NKS; ++i) {
            const int k = kc0 + i * KSTEP;
            bnext[i] = (tile_live && k < Kc) ? wbase[(long long)((k_begin + k) / KSTEP) * 64] : make_uint4(0u, 0u, 0u, 0u);
        }
    };

    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, g = lane >> 4;

    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        signed char* buf = As[c & 1];
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int id = tid + 256 * i, row = id / PIECES, piece = id % PIECES;
            const float m = block_absmax16(areg[i]);
            const float inv = m > 0.f ? 127.0f / m : 0.f;
            *(unsigned*)(buf + row * LDA + piece * 4) = quantize4(areg[i], inv);
            if ((piece & 15) == 0) Dq[c & 1][piece >> 4][row] = m / 127.0f;
        }
#pragma unroll
        for (int i = 0; i < NKS; ++i) bcur[i] = bnext[i];
        __syncthreads();
        if (c + 1 < nchunks) load_chunk(c + 1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const i32x4 b4 = __builtin_bit_cast(i32x4, bcur[ks]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const i32x4 a4 = *(const i32x4*)(buf + (mt * 16 + r) * LDA + ks * 64 + 16 * g);
                const i32x4 z = {0, 0, 0, 0};
                const i32x4 ci = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, z, 0, 0, 0);
                const float4 d4 = *(const float4*)(&Dq[c & 1][ks][mt * 16 + 4 * g]);
                // explicit fused multiply-adds: both int8 product kernels must round alike whatever the compiler pairs up
                acc[mt][0] = __builtin_fmaf((float)ci[0], d4.x, acc[mt][0]); acc[mt][1] = __builtin_fmaf((float)ci[1], d4.y, acc[mt][1]);
                acc[mt][2] = __builtin_fmaf((float)ci[2], d4.z, acc[mt][2]); acc[mt][3] = __builtin_fmaf((float)ci[3], d4.w, acc[mt][3]);
            }
        }
    }
    if (!tile_live) return;
    const int col = nt * 16 + r;
    const float ws = a.wscale[col];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + mt * 16 + g * 4 + i;
            if (row >= a.M) continue;
            float v = __fmul_rn(acc[mt][i], ws);
            if (a.P) {
                a.P[((long long)blockIdx.y * a.M + row) * a.N + col] = v;
            } else {
                if (a.bias) v = __fadd_rn(v, a.bias[col]);
                a.C[(long long)row * a.ldc + col] = act_dev(v, a.act);
            }
        }
    }
}

// ---- int8 weights x dynamically quantised uint8 activations with a zero point (q8_act = 1) -----------------------
// The activation scheme of ONNX DynamicQuantizeLinear + MatMulInteger, which is what a dynamically quantised ONNX graph
// (the reference's *_int8.onnx files, src/pocket_tts.cpp:90-96) applies to every matrix input.  Per row (the reference runs
// one utterance per call, so a decode step's activation tensor IS one row):
//     lo = min(0, min a), hi = max(0, max a), s = (hi - lo) / 255, zp = rint(clamp(-lo / s, 0, 255)),
//     q = clamp(rint(a / s) + zp, 0, 255)                                    (rint: round half to even; s = 0 -> q = zp = 0)
//     y[n] = (float)(sum_k (q[k] - zp) * w[n][k]) * (s * wscale[n]) + bias[n]
// The matrix instruction multiplies signed bytes, so the row is staged as q - 128 and the integer sum is corrected by
// (128 - zp) * (sum_k w[n][k]), the weight row sums (SkinnyArgs::wsum) being formed once when the matrix is loaded: the
// whole sum stays in int32 and is exact.  A workgroup first forms lo / hi of its rows over the full K (every K-slice
// workgroup repeats that), then streams its K-slice as skinny_q8_kernel does.  With K-slices wsum is [slices][N], each
// slice corrects its own part of the sum and writes an already scaled slab (the consumer adds the slabs in f32).
template <int MT>
__global__ __launch_bounds__(256) void skinny_q8o_kernel(SkinnyArgs a) {
    constexpr int ROWS = 16 * MT;
    constexpr int KSTEP = 64, KCH = 128, LDA = KCH + 16, NKS = KCH / KSTEP, PIECES = KCH / 4, APT = ROWS * PIECES / 256;
    __shared__ __attribute__((aligned(16))) signed char As[2][ROWS * LDA];
    __shared__ float row_s[ROWS];
    __shared__ int row_zp[ROWS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x * 4 + wave;
    const bool tile_live = nt * 16 < a.N;
    const int m0 = blockIdx.z * ROWS;
    const int Kc = a.K / a.splits;
    const int k_begin = blockIdx.y * Kc;
    const int nchunks = (Kc + KCH - 1) / KCH;
    const int KS_total = a.K / KSTEP;
    const uint4* wbase = (const uint4*)a.Wp + (long long)nt * KS_total * 64 + lane;

    // ---- row statistics over the full K: wave w takes rows w, w + 4, ...  (from the producers' partial ranges when given)
    for (int rr = wave; rr < ROWS; rr += 4) {
        float lo = 0.f, hi = 0.f;
        if (m0 + rr < a.M) {
            if (a.a_mm) {
                for (int p = lane; p < a.mm_parts; p += 64) {
                    const float2 mm = *(const float2*)(a.a_mm + ((long long)p * a.M + m0 + rr) * 2);
                    lo = fminf(lo, mm.x);
                    hi = fmaxf(hi, mm.y);
                }
            } else {
                const float* ar = a.A + (long long)(m0 + rr) * a.lda;
                for (int k = 4 * lane; k < a.K; k += 256) {
                    const float4 v = *(const float4*)(ar + k);
                    lo = fminf(lo, fminf(fminf(v.x, v.y), fminf(v.z, v.w)));
                    hi = fmaxf(hi, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
                }
            }
        }
#pragma unroll
        for (int of = 32; of > 0; of >>= 1) { lo = fminf(lo, __shfl_xor(lo, of, 64)); hi = fmaxf(hi, __shfl_xor(hi, of, 64)); }
        if (lane == 0) {
            const float sc = (hi - lo) / 255.0f;
            const bool flat = !(sc > 0.f);
            row_s[rr] = flat ? 0.f : sc;
            row_zp[rr] = flat ? 0 : (int)rintf(fminf(fmaxf(-lo / sc, 0.f), 255.f));
        }
    }
    __syncthreads();

    float4 areg[APT];
    uint4 bnext[NKS], bcur[NKS];
    auto load_chunk = [&](int c) {
        const int kc0 = c * KCH;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int id = tid + 256 * i, row = id / PIECES, piece = id % PIECES;
            const int k = kc0 + piece * 4;
            const bool ok = (m0 + row) < a.M && k < Kc;
            areg[i] = ok ? *(const float4*)(a.A + (long long)(m0 + row) * a.lda + k_begin + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NKS; ++i) {
            const int k = kc0 + i * KSTEP;
            bnext[i] = (tile_live && k < Kc) ? wbase[(long long)((k_begin + k) / KSTEP) * 64] : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto quant1 = [](float v, float sc, int zp) -> int {   // x / s, as the operator is specified (a true division)
        int q = (int)rintf(v / sc) + zp;
        q = q < 0 ? 0 : (q > 255 ? 255 : q);
        return (q - 128) & 0xff;
    };

    i32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = i32x4{0, 0, 0, 0};
    const int r = lane & 15, g = lane >> 4;

    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        signed char* buf = As[c & 1];
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int id = tid + 256 * i, row = id / PIECES, piece = id % PIECES;
            const float sc = row_s[row];
            const int zp = row_zp[row];
            unsigned pk;
            if (sc > 0.f) {
                pk = (unsigned)quant1(areg[i].x, sc, zp) | ((unsigned)quant1(areg[i].y, sc, zp) << 8) | ((unsigned)quant1(areg[i].z, sc, zp) << 16) |
                     ((unsigned)quant1(areg[i].w, sc, zp) << 24);
            } else {
                pk = 0x80808080u;   // q = zp = 0 -> staged as -128; the correction term cancels it exactly
            }
            // rows past M and columns past the slice were loaded as 0.0: they quantise to zp, i.e. contribute (zp - zp) = 0
            *(unsigned*)(buf + row * LDA + piece * 4) = pk;
        }
#pragma unroll
        for (int i = 0; i < NKS; ++i) bcur[i] = bnext[i];
        __syncthreads();
        if (c + 1 < nchunks) load_chunk(c + 1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const i32x4 b4 = __builtin_bit_cast(i32x4, bcur[ks]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const i32x4 a4 = *(const i32x4*)(buf + (mt * 16 + r) * LDA + ks * 64 + 16 * g);
                acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, acc[mt], 0, 0, 0);
            }
        }
    }
    if (!tile_live) return;   // (launches that ask for c_mm have N % 64 == 0: no dead tile in front of the barriers below)
    const int col = nt * 16 + r;
    const float ws = a.wscale[col];
    const int wsum = a.wsum[(long long)blockIdx.y * a.N + col];
    float outv[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) outv[mt][i] = 0.f;   // rows past M: 0 is inside every range
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = mt * 16 + g * 4 + i, row = m0 + rl;
            if (row >= a.M) continue;
            const int total = acc[mt][i] + (128 - row_zp[rl]) * wsum;
            float v = __fmul_rn((float)total, __fmul_rn(row_s[rl], ws));
            if (a.P) {
                a.P[((long long)blockIdx.y * a.M + row) * a.N + col] = v;
            } else {
                if (a.bias) v = __fadd_rn(v, a.bias[col]);
                v = act_dev(v, a.act);
                a.C[(long long)row * a.ldc + col] = v;
                outv[mt][i] = v;
            }
        }
    }
    if (!a.c_mm) return;
    // ranges of the result rows over this workgroup's 64 columns: 16 columns per wave by DPP-row shuffles, 4 waves through LDS
    // (the staging buffer is free: every wave is past its last MFMA only after the barrier below)
    __syncthreads();
    float* red = (float*)As;   // [4 waves][ROWS][2]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float lo = outv[mt][i], hi = outv[mt][i];
#pragma unroll
            for (int of = 1; of < 16; of <<= 1) { lo = fminf(lo, __shfl_xor(lo, of, 64)); hi = fmaxf(hi, __shfl_xor(hi, of, 64)); }
            if (r == 0) { red[(wave * ROWS + mt * 16 + g * 4 + i) * 2] = lo; red[(wave * ROWS + mt * 16 + g * 4 + i) * 2 + 1] = hi; }
        }
    }
    __syncthreads();
    if (tid < ROWS && m0 + tid < a.M) {
        float lo = red[tid * 2], hi = red[tid * 2 + 1];
#pragma unroll
        for (int w = 1; w < 4; ++w) { lo = fminf(lo, red[(w * ROWS + tid) * 2]); hi = fmaxf(hi, red[(w * ROWS + tid) * 2 + 1]); }
        *(float2*)(a.c_mm + ((long long)blockIdx.x * a.M + m0 + tid) * 2) = make_float2(lo, hi);
    }
}

// The same product with activations that arrive quantised (SkinnyArgs::Aq / Ad): nothing is staged.  A wave's A
// fragment of k-step ks is 16 bytes per lane straight from the int8 rows (lane (r, g): row m0 + r, k = 64 ks + 16 g ..),
// its scales one float4 of Ad[ks][m0 + 4 g ..]; with the weight fragment that is three independent loads per k-step and no
// LDS, no barrier.  Four k-steps are in flight while the previous four are multiplied.  16 rows per workgroup.
__global__ __launch_bounds__(256) void skinny_q8p_kernel(SkinnyArgs a) {
    __shared__ float smax[4][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x * 4 + wave;
    const bool tile_live = nt * 16 < a.N;
    const int ntc = tile_live ? nt : 0;
    const int m0 = blockIdx.z * 16;
    const int nks = a.K / a.splits / 64, ks0 = blockIdx.y * nks, KS_total = a.K / 64;
    const int r = lane & 15, g = lane >> 4;
    const bool row_ok = m0 + r < a.M;
    const uint4* wp = (const uint4*)a.Wp + ((long long)ntc * KS_total + ks0) * 64 + lane;
    const signed char* ap = a.Aq + (long long)(row_ok ? m0 + r : 0) * a.ldaq + (long long)ks0 * 64 + 16 * g;
    const float* dp = a.Ad + (long long)ks0 * a.Mq + m0 + 4 * g;
    uint4 an[4], bn[4];
    float4 dn[4];
    auto request = [&](int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ks = c + i < nks ? c + i : nks - 1;   // past the end: a repeat of the last step, not used
            an[i] = *(const uint4*)(ap + (long long)ks * 64);
            bn[i] = wp[(long long)ks * 64];
            dn[i] = *(const float4*)(dp + (long long)ks * a.Mq);
        }
    };
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    request(0);
    for (int c = 0; c < nks; c += 4) {
        uint4 ac[4], bc[4];
        float4 dc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { ac[i] = an[i]; bc[i] = bn[i]; dc[i] = dn[i]; }
        if (c + 4 < nks) request(c + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (c + i >= nks) break;
            const i32x4 z = {0, 0, 0, 0};
            const i32x4 a4 = row_ok ? __builtin_bit_cast(i32x4, ac[i]) : z;
            const i32x4 ci = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, __builtin_bit_cast(i32x4, bc[i]), z, 0, 0, 0);
            acc[0] = __builtin_fmaf((float)ci[0], dc[i].x, acc[0]); acc[1] = __builtin_fmaf((float)ci[1], dc[i].y, acc[1]);
            acc[2] = __builtin_fmaf((float)ci[2], dc[i].z, acc[2]); acc[3] = __builtin_fmaf((float)ci[3], dc[i].w, acc[3]);
        }
    }
    const int col = ntc * 16 + r;
    const float ws = a.wscale[col];
    if (a.Cq) {
        // result as int8 blocks of 64 columns = this workgroup's four tiles: row maxima over a tile's 16 columns by DPP
        // (the 16 lanes of one g), over the four waves through LDS
        const float bias = a.bias ? a.bias[col] : 0.f;
        float v[4], mx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = __fmul_rn(acc[i], ws);
            if (a.bias) t = __fadd_rn(t, bias);      // the same two roundings as the f32 epilogues
            v[i] = act_dev(t, a.act);
            float m = fabsf(v[i]);
            m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true)));
            m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xF, 0xF, true)));
            m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x141, 0xF, 0xF, true)));
            m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x140, 0xF, 0xF, true)));
            mx[i] = m;
        }
        if (r == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) smax[wave][4 * g + i] = mx[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = 4 * g + i, row = m0 + rr;
            const float m = fmaxf(fmaxf(smax[0][rr], smax[1][rr]), fmaxf(smax[2][rr], smax[3][rr]));
            if (row >= a.M) continue;
            const float inv = m > 0.f ? 127.0f / m : 0.f;
            a.Cq[(long long)row * a.N + col] = (signed char)(int)rintf(v[i] * inv);
            if (wave == 0 && r == 0) a.Cd[(long long)blockIdx.x * a.Mq + row] = m / 127.0f;
        }
        return;
    }
    if (!tile_live) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + g * 4 + i;
        if (row >= a.M) continue;
        float v = __fmul_rn(acc[i], ws);
        if (a.P) {
            a.P[((long long)blockIdx.y * a.M + row) * a.N + col] = v;
        } else {
            if (a.bias) v = __fadd_rn(v, a.bias[col]);
            a.C[(long long)row * a.ldc + col] = act_dev(v, a.act);
        }
    }
}

// int8 fragment-major packing: tile (nt, ks) holds, for lane l, the 16 consecutive k-values ks*64 + 16*(l>>4) ..
// of output column nt*16 + (l&15).
std::vector<signed char> quantize_q8_rows(const float* W, int N, int K, std::vector<float>& scale) {
    if (N % 16 || K % 64) throw std::runtime_error("int8 weight: N must be a multiple of 16 and K of 64");
    scale.assign(N, 0.f);
    std::vector<signed char> q((size_t)N * K);
    for (int n = 0; n < N; ++n) {
        float mx = 0.f;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, fabsf(W[(size_t)n * K + k]));
        const float inv = mx > 0.f ? 127.0f / mx : 0.f;
        scale[n] = mx / 127.0f;
        for (int k = 0; k < K; ++k) {
            float rv = rintf(W[(size_t)n * K + k] * inv);
            rv = fminf(fmaxf(rv, -127.f), 127.f);
            q[(size_t)n * K + k] = (signed char)rv;
        }
    }
    return q;
}

std::vector<unsigned char> pack_q8_weight(const float* W, int N, int K, std::vector<float>& scale) {
    const std::vector<signed char> q = quantize_q8_rows(W, N, K, scale);
    return pack_q8_layout(q.data(), N, K);
}

// sums[s][n] = sum over K-slice s of q[n][k]  (the zero-point correction of the dynamic uint8 activation scheme)
std::vector<int> q8_row_sums(const signed char* q, int N, int K, int splits) {
    if (splits < 1 || K % splits) throw std::runtime_error("int8 weight: K-slices must divide K");
    std::vector<int> out((size_t)splits * N, 0);
    const int Kc = K / splits;
    for (int n = 0; n < N; ++n)
        for (int sl = 0; sl < splits; ++sl) {
            int acc = 0;
            for (int k = 0; k < Kc; ++k) acc += q[(size_t)n * K + (size_t)sl * Kc + k];
            out[(size_t)sl * N + n] = acc;
        }
    return out;
}

std::vector<unsigned char> pack_q8_layout(const signed char* q, int N, int K) {
    if (N % 16 || K % 64) throw std::runtime_error("int8 weight: N must be a multiple of 16 and K of 64");
    std::vector<unsigned char> out((size_t)N * K);
    const int KS = K / 64;
    for (int nt = 0; nt < N / 16; ++nt)
        for (int ks = 0; ks < KS; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 16; ++j)
                    out[(((size_t)nt * KS + ks) * 64 + l) * 16 + j] = (unsigned char)q[(size_t)(nt * 16 + (l & 15)) * K + ks * 64 + 16 * (l >> 4) + j];
    return out;
}


// ---- rows-stationary product: a few hundred rows against thousands of columns at K = 512 ------------------------
// (the adaLN vectors of a frame: 640 x 10240 x 512).  In skinny_gemm_kernel every workgroup of 64 columns stages its
// rows again (3200 workgroups x 64 KB through LDS at that shape) and that staging, not the matrix work, is its time.
// Here a wave keeps the two-term bf16 fragments of its 32 rows in registers for the whole launch (256 registers) and
// the weights stream past them: a workgroup is 4 waves = 128 rows
// and walks its share of the 16-column tiles; each wave fetches a quarter of the next tiles' fragments three tiles
// ahead (six were measured no faster: the weight stream is not what a tile waits for), the tile meets the four waves through a two-slot LDS buffer (one barrier per tile), 64 MFMAs per wave and
// tile.  Workgroup L runs on XCD L % 8 and takes column partition (L % 8) + 8 k: an XCD's L2 sees an eighth of the
// weights and every row.  Products are accumulated in skinny_gemm_kernel's order (k-steps ascending, low term
// before high term), so the result is the same bit for bit.
constexpr int kWideK = 512, kWideKS = kWideK / 32, kWideAhead = 3, kWideLd = 64 + 8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// MT: 16-row tiles per wave.  2: 32 rows per wave, 128 rows per workgroup.  1: 16 rows per wave, for row counts that fill
// 64-row slabs only.  One workgroup per CU either way: two of the second form per CU (256 registers, a few spilled) were
// measured slower (25.5 against 23 us); two tiles per barrier were measured too (24 us, no gain).  Whatever the form,
// the launch takes about 36 cycles per MFMA where 16 is the issue rate.
// FRAG: the rows arrive as two-term bf16 fragments (SkinnyArgs::a_frag): one 16-byte load per fragment, no conversion,
// no staging.
// W2: the matrix comes as two bf16 planes (SkinnyArgs::Wp2); a tile then takes 96 MFMAs per wave instead of 64 and both
// planes travel through LDS (the tile's fragments are read from LDS in two halves of 8 k-steps, which keeps the kernel
// inside 512 registers).
template <int MT, bool FRAG, bool W2>
__global__ __launch_bounds__(256, 1) void wide_rows_kernel(SkinnyArgs a, int pk) {
    constexpr int kWideRows = 64 * MT;
    constexpr int NPL = W2 ? 2 : 1;                          // weight planes
    constexpr int kSlot = kWideKS * 64;                      // 16-byte units of one tile of one plane
    // [plane][slot][k-step][lane]; the per-wave row staging areas of the prologue ([wave][term][16 * kWideLd] bf16) lie over
    // the same bytes and are done with before the first tile is written (one barrier)
    __shared__ __attribute__((aligned(16))) u32x4 Wl[NPL * 2 * kSlot];
    static_assert(sizeof(u32x4) * 2 * kSlot >= 4 * 2 * 16 * kWideLd * sizeof(__bf16), "wide rows: the staging areas must fit under the tile slots");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int slab = j / pk, P = 8 * pk, part = xcd + 8 * (j % pk);
    const int NT = a.N / 16;
    const int t0 = (int)((long long)part * NT / P), n = (int)((long long)(part + 1) * NT / P) - t0;
    const int m0 = slab * kWideRows + wave * 16 * MT;
    const u32x4* wp = (const u32x4*)a.Wp + (long long)t0 * kWideKS * 64 + (4 * wave) * 64 + lane;
    const u32x4* wp2 = (const u32x4*)(W2 ? a.Wp2 : a.Wp) + (long long)t0 * kWideKS * 64 + (4 * wave) * 64 + lane;

    if (n <= 0) return;
    // A staging set = this wave's quarter of a tile's fragments and the tile's bias element, requested three tiles ahead.
    // Tiles past the partition's end re-read its last tile: every load is unconditional and the main loop below is
    // straight-line code, so each wait counts exactly the requests issued after the one it needs (a conditional load or
    // store anywhere in the loop makes every wait a full drain, stores included).
    u32x4 st0[4 * NPL], st1[4 * NPL], st2[4 * NPL];
    float sb0, sb1, sb2;
    auto issue = [&](u32x4 (&dst)[4 * NPL], float& bias, int i) {
        const int ii = i < n ? i : n - 1;
        const u32x4* w = wp + (long long)ii * kWideKS * 64;
        bias = a.bias[(t0 + ii) * 16 + r];
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = w[q * 64];
        if constexpr (W2) {
            const u32x4* w2 = wp2 + (long long)ii * kWideKS * 64;
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[4 + q] = w2[q * 64];
        }
    };
    issue(st0, sb0, 0);
    issue(st1, sb1, 1);
    issue(st2, sb2, 2);

    bf16x8 ah[MT][kWideKS], al[MT][kWideKS];
    // Row fragments.  Read in fragment order, a wave instruction would touch 64 different 128-byte lines for 16 bytes each
    // (a lane per row); instead each wave reads its rows in row-major runs (16 lanes = 256 contiguous bytes), splits
    // them into the two bf16 terms, and turns 16 rows x 64 k at a time into fragment order through a staging area of its
    // own in LDS (a wave's LDS operations execute in order, so no workgroup barrier is involved).
    if constexpr (FRAG) {
        const u32x4* af = (const u32x4*)a.A + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ks = 0; ks < kWideKS; ++ks) {
                const long long unit = ((long long)(m0 / 16 + mt) * kWideKS + ks) * 2;
                ah[mt][ks] = __builtin_bit_cast(bf16x8, af[unit * 64]);
                al[mt][ks] = __builtin_bit_cast(bf16x8, af[(unit + 1) * 64]);
            }
    } else
    {
        __bf16* sh = (__bf16*)Wl + (wave * 2) * 16 * kWideLd;
        __bf16* sl = sh + 16 * kWideLd;
        const int lrow = lane >> 4, lk = (lane & 15) * 4;
        // every load of the wave's rows is requested before the first conversion (MT x 32 float4 per lane)
        float4 v[MT][kWideK / 64][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float* ap = a.A + (long long)(m0 + mt * 16 + lrow) * a.lda + lk;
#pragma unroll
            for (int kc = 0; kc < kWideK / 64; ++kc)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[mt][kc][i] = *(const float4*)(ap + (long long)(4 * i) * a.lda + kc * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int kc = 0; kc < kWideK / 64; ++kc) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    bf16x4 hi, lo;
                    split_bf16(v[mt][kc][i], hi, lo);
                    *(bf16x4*)(sh + (lrow + 4 * i) * kWideLd + lk) = hi;
                    *(bf16x4*)(sl + (lrow + 4 * i) * kWideLd + lk) = lo;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    ah[mt][kc * 2 + j] = *(const bf16x8*)(sh + r * kWideLd + j * 32 + 8 * g);
                    al[mt][kc * 2 + j] = *(const bf16x8*)(sl + r * kWideLd + j * 32 + 8 * g);
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();   // the staging areas lie under the tile slots
    }

    float* crow = a.C + (long long)(m0 + g * 4) * a.ldc + t0 * 16 + r;
    auto step = [&](u32x4 (&cur)[4 * NPL], float& bias, int i) {
        u32x4* slot = Wl + (i & 1) * kSlot;
        u32x4* slot2 = Wl + (2 + (i & 1)) * kSlot;
#pragma unroll
        for (int q = 0; q < 4; ++q) slot[(4 * wave + q) * 64 + lane] = cur[q];
        if constexpr (W2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) slot2[(4 * wave + q) * 64 + lane] = cur[4 + q];
        }
        const float bias_col = bias;
        __syncthreads();
        issue(cur, bias, i + kWideAhead);
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (W2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                constexpr int HK = kWideKS / 2;
                u32x4 bw[HK], bw2[HK];
#pragma unroll
                for (int ks = 0; ks < HK; ++ks) { bw[ks] = slot[(h * HK + ks) * 64 + lane]; bw2[ks] = slot2[(h * HK + ks) * 64 + lane]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < HK; ++ks) {
                    const bf16x8 b8 = __builtin_bit_cast(bf16x8, bw[ks]), b8l = __builtin_bit_cast(bf16x8, bw2[ks]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt][h * HK + ks], b8l, acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mt][h * HK + ks], b8, acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt][h * HK + ks], b8, acc[mt], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // every fragment of the tile is requested from LDS before the first MFMA (one LDS latency per tile, not one per k-step)
            u32x4 bw[kWideKS];
#pragma unroll
            for (int ks = 0; ks < kWideKS; ++ks) bw[ks] = slot[ks * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < kWideKS; ++ks) {
                const bf16x8 b8 = __builtin_bit_cast(bf16x8, bw[ks]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mt][ks], b8, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt][ks], b8, acc[mt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float* c = crow + i * 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) c[(long long)(mt * 16 + e) * a.ldc] = acc[mt][e] + bias_col;
        }
    };
    int i = 0;
    for (; i + kWideAhead <= n; i += kWideAhead) {
        step(st0, sb0, i);
        step(st1, sb1, i + 1);
        step(st2, sb2, i + 2);
    }
    if (i < n) step(st0, sb0, i);          // n is the same for the whole workgroup: these barriers are uniform
    if (i + 1 < n) step(st1, sb1, i + 1);
}

// Whether launch_skinny_gemm hands this product to wide_rows_kernel: bf16 weights, K = 512, plain C = acc + bias, whole
// 64-row slabs (no row guards in the kernel), at least 256 rows and enough column tiles for every partition.
// POCKET_TTS_WIDE_ROWS=0 keeps skinny_gemm_kernel.
bool skinny_wide_rows_applies(const SkinnyArgs& a) {
    const char* e = std::getenv("POCKET_TTS_WIDE_ROWS");   // read per launch (launches are captured once per graph): tests flip it
    const bool on = !e || std::atoi(e) != 0;
    return on && a.wtype == W_BF16 && a.K == kWideK && a.splits == 1 && !a.P && a.C && a.bias && a.act == ACT_NONE && !a.colscale && !a.R &&
           !a.Aq && !a.Cq && a.M >= 256 && a.M % 64 == 0 && a.N / 16 >= 256;
}

void launch_skinny_gemm(const SkinnyArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return;
    const bool bf = a.wtype == W_BF16;
    const int kstep = a.wtype == W_I8 ? 64 : bf ? 32 : 16;
    if (a.wtype == W_I8 && !a.wscale) throw std::runtime_error("skinny gemm: int8 weights need their scales");
    if (a.N % 16 || a.K % kstep || a.splits <= 0 || a.K % a.splits || (a.K / a.splits) % kstep || (!a.Aq && ((a.lda % 4) || ((uintptr_t)a.A & 15))))
        throw std::runtime_error("skinny gemm: shape not tileable (N % 16, K % k-step, K % splits, lda alignment)");
    if (!a.P && a.splits != 1) throw std::runtime_error("skinny gemm: split K needs a partial buffer");
    if (a.a_frag && !skinny_wide_rows_applies(a)) throw std::runtime_error("skinny gemm: fragment-order rows are read by the rows-stationary kernel only");
    if (skinny_wide_rows_applies(a)) {
        // 32 rows per wave where the rows fill 128-row slabs, 16 otherwise (POCKET_TTS_WIDE_ROWS=1 forces the second form)
        const char* e = std::getenv("POCKET_TTS_WIDE_ROWS");
        const int mt = (a.M % 128 == 0 && !(e && std::atoi(e) == 1)) ? 2 : 1;
        const int slabs = a.M / (64 * mt);
        // about one workgroup per CU, at least 4 tiles each
        const int pk = std::max(1, std::min(a.N / 16 / 8 / 4, (256 + 4 * slabs) / (8 * slabs)));
        const dim3 wgrid(8 * pk * slabs);
        if (a.Wp2) {
            if (mt == 2 && a.a_frag) hipLaunchKernelGGL((wide_rows_kernel<2, true, true>), wgrid, dim3(256), 0, s, a, pk);
            else if (mt == 2) hipLaunchKernelGGL((wide_rows_kernel<2, false, true>), wgrid, dim3(256), 0, s, a, pk);
            else if (a.a_frag) hipLaunchKernelGGL((wide_rows_kernel<1, true, true>), wgrid, dim3(256), 0, s, a, pk);
            else hipLaunchKernelGGL((wide_rows_kernel<1, false, true>), wgrid, dim3(256), 0, s, a, pk);
        }
        else if (mt == 2 && a.a_frag) hipLaunchKernelGGL((wide_rows_kernel<2, true, false>), wgrid, dim3(256), 0, s, a, pk);
        else if (mt == 2) hipLaunchKernelGGL((wide_rows_kernel<2, false, false>), wgrid, dim3(256), 0, s, a, pk);
        else if (a.a_frag) hipLaunchKernelGGL((wide_rows_kernel<1, true, false>), wgrid, dim3(256), 0, s, a, pk);
        else hipLaunchKernelGGL((wide_rows_kernel<1, false, false>), wgrid, dim3(256), 0, s, a, pk);
        return;
    }
    dim3 grid((a.N + 63) / 64, a.splits, (a.M + 63) / 64), block(256);
    // rows per workgroup: 64, 32 or 16.  Fewer rows = more workgroups streaming the same weights (served by the
    // XCD's L2 after the first: same-column workgroups differ by a multiple of 8 in linear id when grid.x * grid.y
    // is one) and a shorter LDS / MFMA phase per K chunk.
    // Measured at M = 64 (QKV / FFN / output products of the LM step): the time of a launch follows the LDS + MFMA
    // work per workgroup, not the bytes, until there are about 500 workgroups; 16-row workgroups are 2x faster
    // than 64-row ones on the unsplit products.
    int mt = a.mtiles;
    if (mt != 1 && mt != 2 && mt != 4) {
        const long long cols = (long long)grid.x * grid.y;
        mt = 4;
        while (mt > 1 && cols * ((a.M + 16 * mt - 1) / (16 * mt)) < 512) mt >>= 1;
        if (a.M <= 16 * (mt / 2)) mt = a.M <= 16 ? 1 : 2;     // never stage rows that do not exist
    }
    if (a.Aq && a.q8_act == 1) throw std::runtime_error("skinny gemm: pre-quantised rows exist in the block scheme only");
    if (a.Aq) {
        if (a.wtype != W_I8 || !a.Ad || a.Mq < ((a.M + 15) & ~15) || (a.Mq & 3) || (a.ldaq & 15) || ((uintptr_t)a.Aq & 15) || ((uintptr_t)a.Ad & 15))
            throw std::runtime_error("skinny gemm: quantised activations need int8 weights, scales [K/64][Mq] and 16-byte aligned rows");
        if (a.Cq && (!a.Cd || a.N % 64 || a.splits != 1)) throw std::runtime_error("skinny gemm: a quantised result needs N % 64 == 0 and no K split");
        grid.z = (a.M + 15) / 16;
        hipLaunchKernelGGL(skinny_q8p_kernel, grid, block, 0, s, a);
        return;
    }
    if (a.Cq) throw std::runtime_error("skinny gemm: a quantised result is produced from quantised activations only");
    if ((a.colscale || a.R) && (a.P || a.wtype == W_I8)) throw std::runtime_error("skinny gemm: column scale and residual belong to the f32 / bf16 C form");
    grid.z = (a.M + 16 * mt - 1) / (16 * mt);
    if (a.wtype == W_I8 && a.q8_act == 1) {
        if (!a.wsum) throw std::runtime_error("skinny gemm: the dynamic uint8 activation scheme needs the weight row sums");
        if (a.lda % 4 || a.K % 4) throw std::runtime_error("skinny gemm: rows must be 16-byte aligned for the dynamic uint8 scheme");
        if (a.a_mm && a.mm_parts < 1) throw std::runtime_error("skinny gemm: row ranges need their part count");
        if (a.c_mm && (a.P || a.N % 64)) throw std::runtime_error("skinny gemm: result ranges go with the C form and N % 64 == 0");
        if (mt == 1) hipLaunchKernelGGL(skinny_q8o_kernel<1>, grid, block, 0, s, a);
        else if (mt == 2) hipLaunchKernelGGL(skinny_q8o_kernel<2>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(skinny_q8o_kernel<4>, grid, block, 0, s, a);
        return;
    }
    if (a.wtype == W_I8) {
        if (mt == 1) hipLaunchKernelGGL(skinny_q8_kernel<1>, grid, block, 0, s, a);
        else if (mt == 2) hipLaunchKernelGGL(skinny_q8_kernel<2>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(skinny_q8_kernel<4>, grid, block, 0, s, a);
        return;
    }
    const int Kc = a.K / a.splits;
    int kch = a.kchunk;
    if (kch != 128 && kch != 256 && kch != 512) kch = (mt == 1 && Kc >= 1024) ? 512 : (mt <= 2 && Kc >= 256) ? 256 : 128;
    if (Kc % kch) kch = 128;
    if (a.Wp2 && !bf) throw std::runtime_error("skinny gemm: a second weight plane goes with bf16 weights");
    if (bf && a.Wp2) {   // two-term matrix: half the K chunk of the one-term form at most (twice the fragment registers)
        if (kch == 512) kch = 256;
        if (mt == 1 && kch == 256) hipLaunchKernelGGL((skinny_gemm_kernel<__bf16, 1, 256, true>), grid, block, 0, s, a);
        else if (mt == 1) hipLaunchKernelGGL((skinny_gemm_kernel<__bf16, 1, 128, true>), grid, block, 0, s, a);
        else if (mt == 2 && kch == 256) hipLaunchKernelGGL((skinny_gemm_kernel<__bf16, 2, 256, true>), grid, block, 0, s, a);
        else if (mt == 2) hipLaunchKernelGGL((skinny_gemm_kernel<__bf16, 2, 128, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((skinny_gemm_kernel<__bf16, 4, 128, true>), grid, block, 0, s, a);
    } else if (bf) {
        if (mt == 1 && kch == 512) hipLaunchKernelGGL((skinny_gemm_kernel<__bf16, 1, 512>), grid, block, 0, s, a);
        else if (mt == 1 && kch == 256) hipLaunchKernelGGL((skinny_gemm_kernel<__bf16, 1, 256>), grid, block, 0, s, a);
        else if (mt == 1) hipLaunchKernelGGL((skinny_gemm_kernel<__bf16, 1, 128>), grid, block, 0, s, a);
        else if (mt == 2 && kch == 256) hipLaunchKernelGGL((skinny_gemm_kernel<__bf16, 2, 256>), grid, block, 0, s, a);
        else if (mt == 2) hipLaunchKernelGGL((skinny_gemm_kernel<__bf16, 2, 128>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((skinny_gemm_kernel<__bf16, 4, 128>), grid, block, 0, s, a);
    } else {
        if (mt == 1) hipLaunchKernelGGL((skinny_gemm_kernel<float, 1, 128>), grid, block, 0, s, a);
        else if (mt == 2) hipLaunchKernelGGL((skinny_gemm_kernel<float, 2, 128>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((skinny_gemm_kernel<float, 4, 128>), grid, block, 0, s, a);
    }
}

// ------------------------------------------------------------------ add & norm, one 256-thread block per row
// Thread t owns channels 4t..4t+3 (D <= 1024).  All K-slice loads of a thread are independent 16-byte
// loads issued together; the slices are then added in slice order, so the sum is bitwise reproducible.

// sum over the 64 lanes, the same value in every lane: four DPP steps give each 16-lane row its sum, the four row
// sums are read as scalars
__device__ __forceinline__ float wsum64(float v) {
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x140>(v);
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ float block_sum4(float v, float* red, int tid) {
    v = wsum64(v);
    __syncthreads();  // red[] may still be read from a previous call
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

template <int MAXS>
__global__ __launch_bounds__(256) void add_norm_kernel(AddNormArgs a) {
    __shared__ float red[4];
    const int tid = threadIdx.x, m = blockIdx.x;
    const int c = 4 * tid;
    const bool on = c < a.D;
    const int cc = on ? c : 0;                       // lanes past the row read column 0 and store nothing
    float* __restrict__ xr = a.X + (long long)m * a.ldx;
    const float* __restrict__ P = a.P;
    // every operand of the launch is requested here, before anything is waited for
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v;
    if (a.seq) {
        // input projection of the LM (no bias): this thread's four output columns, K = L
        const float* sq = a.seq + (long long)m * a.L;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.w_in_type == W_BF16) {
            const __bf16* wr = (const __bf16*)a.w_in + (long long)cc * a.L;
            for (int k = 0; k < a.L; k += 8) {
                const float4 sa = *(const float4*)(sq + k), sb = *(const float4*)(sq + k + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16x8 w8 = *(const bf16x8*)(wr + (long long)i * a.L + k);
                    o[i] += ((float)w8[0] * sa.x + (float)w8[1] * sa.y) + ((float)w8[2] * sa.z + (float)w8[3] * sa.w) +
                            ((float)w8[4] * sb.x + (float)w8[5] * sb.y) + ((float)w8[6] * sb.z + (float)w8[7] * sb.w);
                }
            }
        } else {
            const float* wr = (const float*)a.w_in + (long long)cc * a.L;
            for (int k = 0; k < a.L; k += 4) {
                const float4 sa = *(const float4*)(sq + k);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 w4f = *(const float4*)(wr + (long long)i * a.L + k);
                    o[i] += (w4f.x * sa.x + w4f.y * sa.y) + (w4f.z * sa.z + w4f.w * sa.w);
                }
            }
        }
        v = make_float4(o[0], o[1], o[2], o[3]);
        if (on) *(float4*)(xr + c) = v;
    } else {
        v = *(const float4*)(xr + cc);
    }
    float4 part[MAXS];
#pragma unroll
    for (int k = 0; k < MAXS; ++k) part[k] = (P && k < a.splits) ? *(const float4*)(P + ((long long)k * a.M + m) * a.D + cc) : zero;
    const float4 pb4 = (P && a.pbias) ? *(const float4*)(a.pbias + cc) : zero;
    const float4 s4 = (P && a.colscale) ? *(const float4*)(a.colscale + cc) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 g4 = a.g ? *(const float4*)(a.g + cc) : zero, b4 = a.g ? *(const float4*)(a.b + cc) : zero;
    const float4 w4 = a.dot_w ? *(const float4*)(a.dot_w + cc) : zero;
    if (!on) v = zero;
    if (P) {
        float4 p = zero;
#pragma unroll
        for (int k = 0; k < MAXS; ++k) { p.x += part[k].x; p.y += part[k].y; p.z += part[k].z; p.w += part[k].w; }
        if (on) {
            p.x = (p.x + pb4.x) * s4.x; p.y = (p.y + pb4.y) * s4.y; p.z = (p.z + pb4.z) * s4.z; p.w = (p.w + pb4.w) * s4.w;
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
            *(float4*)(xr + c) = v;
        }
    }
    if (a.pos && tid == 0) a.pos[m] += a.pos_delta;
    if (!a.g) return;
    const float mean = block_sum4((v.x + v.y) + (v.z + v.w), red, tid) / (float)a.D;
    const float d0 = on ? v.x - mean : 0.f, d1 = on ? v.y - mean : 0.f, d2 = on ? v.z - mean : 0.f, d3 = on ? v.w - mean : 0.f;
    const float rstd = 1.0f / sqrtf(block_sum4((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3), red, tid) / (float)a.D + a.eps);
    float dot = 0.f;
    // (computed once: the int8 form below must round the very values the f32 form stores)
    const float y0 = d0 * rstd * g4.x + b4.x, y1 = d1 * rstd * g4.y + b4.y, y2 = d2 * rstd * g4.z + b4.z, y3 = d3 * rstd * g4.w + b4.w;
    if (on) {
        if (a.H) *(float4*)(a.H + (long long)m * a.D + c) = make_float4(y0, y1, y2, y3);
        if (a.H32) *(float4*)(a.H32 + (long long)m * a.D + c) = make_float4(y0, y1, y2, y3);
        dot = (y0 * w4.x + y1 * w4.y) + (y2 * w4.z + y3 * w4.w);
    }
    if (a.Hq) {
        // y as int8: a block of 64 channels is 16 consecutive threads (one DPP row)
        const float4 y4 = on ? make_float4(y0, y1, y2, y3) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float bm = block_absmax16(y4);
        const float inv = bm > 0.f ? 127.0f / bm : 0.f;
        if (on) {
            *(unsigned*)(a.Hq + (long long)m * a.D + c) = quantize4(y4, inv);
            if ((tid & 15) == 0) a.Hd[(long long)(c >> 6) * a.Mq + m] = bm / 127.0f;
        }
    }
    if (a.H_mm) {
        // the row's range: lanes past the row hold 0, which the consumer includes anyway
        float lo = on ? fminf(fminf(y0, y1), fminf(y2, y3)) : 0.f, hi = on ? fmaxf(fmaxf(y0, y1), fmaxf(y2, y3)) : 0.f;
#pragma unroll
        for (int of = 32; of > 0; of >>= 1) { lo = fminf(lo, __shfl_xor(lo, of, 64)); hi = fmaxf(hi, __shfl_xor(hi, of, 64)); }
        __shared__ float mm[8];
        if ((tid & 63) == 0) { mm[(tid >> 6) * 2] = lo; mm[(tid >> 6) * 2 + 1] = hi; }
        __syncthreads();
        if (tid == 0)
            *(float2*)(a.H_mm + (long long)m * 2) = make_float2(fminf(fminf(mm[0], mm[2]), fminf(mm[4], mm[6])), fmaxf(fmaxf(mm[1], mm[3]), fmaxf(mm[5], mm[7])));
    }
    if (a.dot_w) {
        dot = block_sum4(dot, red, tid);
        if (tid == 0) a.dot_out[m] = dot + a.dot_b[0];
    }
}

void launch_add_norm(const AddNormArgs& a, hipStream_t s) {
    if (a.M <= 0) return;
    if (a.D % 4 || a.D > 1024 || (a.ldx & 3)) throw std::runtime_error("add_norm: width must be a multiple of 4 and <= 1024");
    if (a.P && (a.splits < 1 || a.splits > 16)) throw std::runtime_error("add_norm: 1..16 K-slices supported");
    if (a.Hq && (!a.Hd || a.D % 64 || a.Mq < a.M || !a.g)) throw std::runtime_error("add_norm: int8 output needs scales, D % 64 == 0 and a normalisation");
    if (a.seq && (!a.w_in || a.L <= 0 || (a.L & 7) || a.P)) throw std::runtime_error("add_norm: the projection prologue needs L % 8 == 0 and no K-slices");
    dim3 grid(a.M), block(256);
    if (a.splits <= 8) hipLaunchKernelGGL(add_norm_kernel<8>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(add_norm_kernel<16>, grid, block, 0, s, a);
}

// ------------------------------------------------------------------ decode attention, one 4-wave block per (row, head)
// The earlier keys of the slot are cut into four contiguous ranges, one per wave (the step is latency
// bound, so more independent loads in flight matter more than work per wave).  Each wave runs local
// softmaxes over its range and leaves (max, sum, unnormalised output) partials in LDS; wave 0 merges them and
// adds the new token, whose rotated key and value never leave registers.  Every wave rotates q itself; wave 0
// appends k, v to the cache.
// bf16 cache (the batched frame loop): lane (g, c) = (lane / 8, lane % 8) holds channels 8c..8c+7 of keys g, g+8, ...:
// one 16-byte load per key for K and for V, a wave instruction covering eight whole 128-byte rows, the 16 loads
// of the wave's first 64 keys requested before anything waits on q.  Scores are reduced over the 8 channel lanes
// by DPP, the softmax and P.V stay in registers in that same layout, and each 16-lane row leaves its own partial
// (16 partials per workgroup), so nothing crosses a DPP row.  A wave with more than 64 keys (contexts beyond 256)
// walks further blocks of 64 through the same registers under a running softmax.  The kernel is held to 128
// VGPRs: a frame step launches four workgroups per CU and they must all be resident at once.
// fp32 cache: lane per key for the scores, then 4 key groups x 16 channel quads for P.V, one partial per wave.
// NT (bf16 cache): the cache rows are loaded non-temporally.  Each K / V row is read by exactly one workgroup per step and
// the caches of a table (hundreds of MB) are larger than what the chip keeps: marking them streaming leaves the weights,
// which every step reads again, in the last-level cache.
template <bool KV16, bool NT = false>
__global__ __launch_bounds__(256, 4) void attn_decode_kernel(AttnDecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const int p = a.pos[m];
    const int pc = p < a.tab_len ? p : a.tab_len - 1;
    const float cs = a.cos_tab[(long long)pc * 32 + (lane >> 1)], sn = a.sin_tab[(long long)pc * 32 + (lane >> 1)];
    const long long cbase = (long long)(a.slot0 + m) * a.cap * a.d + h * 64;
    int j0 = 0;
    if (a.window > 0 && p - a.window + 1 > 0) j0 = p - a.window + 1;
    const int nk = p - j0;                    // earlier keys only
    const int chunk = ((nk + 3) / 4 + 3) & ~3; // keys per wave, multiple of 4
    const int kb = w * chunk, ke = min(nk, kb + chunk), cnt = max(ke - kb, 0);
    const int kg = lane >> 4, cq = lane & 15;
    const float* row = a.qkv + (long long)m * 3 * a.d + h * 64;
    float qv = row[lane], kv = row[a.d + lane], vv = row[2 * a.d + lane];
    for (int s = 1; s < a.qkv_splits; ++s) {
        const float* rs = row + (long long)s * a.M * 3 * a.d;
        qv += rs[lane]; kv += rs[a.d + lane]; vv += rs[2 * a.d + lane];
    }
    // fast path: the K and V elements of the wave's first 64 keys are requested here; later blocks of 64 keys reuse
    // the same registers inside the loop below.  Out-of-range keys re-read the block's last key (their weight is
    // zero below).
    bf16x8 kreg[8], vreg[8];
    const bool early = KV16 && cnt > 0;
    const int kg8 = lane >> 3, c8 = lane & 7;
    auto request_block = [&](int c0) {
        const int cb = min(cnt - c0, 64);
        const int nb = __builtin_amdgcn_readfirstlane((cb + 7) >> 3);   // 8-key steps of this block
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nb) {   // wave-uniform: whole instructions of out-of-range keys are not issued
                const int jk = c0 + (kg8 + 8 * i < cb ? kg8 + 8 * i : cb - 1);
                const bf16x8* kp = (const bf16x8*)((const __bf16*)a.kcache + cbase + (long long)((j0 + kb + jk) % a.cap) * a.d + 8 * c8);
                kreg[i] = NT ? __builtin_nontemporal_load(kp) : *kp;
            }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nb) {
                const int jk = c0 + (kg8 + 8 * i < cb ? kg8 + 8 * i : cb - 1);
                const bf16x8* vp = (const bf16x8*)((const __bf16*)a.vcache + cbase + (long long)((j0 + kb + jk) % a.cap) * a.d + 8 * c8);
                vreg[i] = NT ? __builtin_nontemporal_load(vp) : *vp;
            }
    };
    if constexpr (KV16) {
        if (early) request_block(0);
    }
    const float qo = __shfl_xor(qv, 1, 64), ko = __shfl_xor(kv, 1, 64);
    // pair (2i, 2i+1): even lane x*c - y*s, odd lane x*s + y*c, with x the even and y the odd element
    const float qr = (lane & 1) ? (qo * sn + qv * cs) : (qv * cs - qo * sn);
    const float kr = (lane & 1) ? (ko * sn + kv * cs) : (kv * cs - ko * sn);
    if (w == 0) {
        const long long at = cbase + (long long)(p % a.cap) * a.d + lane;
        if constexpr (KV16) { ((__bf16*)a.kcache)[at] = (__bf16)kr; ((__bf16*)a.vcache)[at] = (__bf16)vv; }
        else { ((float*)a.kcache)[at] = kr; ((float*)a.vcache)[at] = vv; }
    }
    const int scap = max((a.cap / 4 + 8) & ~3, 64);
    float* qs = smem + (long long)w * (64 + scap);
    float* sc = qs + 64;
    float* comb = smem + 4ll * (64 + scap);   // [16][68]: max, sum, pad, pad, out[64]; partial 4w + r
    qs[lane] = qr;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    auto ld4 = [&](const void* base, long long idx) -> float4 {
        if constexpr (KV16) {
            const bf16x4 v = *(const bf16x4*)((const __bf16*)base + idx);
            return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
        } else {
            return *(const float4*)((const float*)base + idx);
        }
    };
    float mx = -INFINITY, sum = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    bool done = false;
    if constexpr (KV16) {
        if (early) {
            const float4 qa = *(const float4*)(qs + 8 * c8), qb = *(const float4*)(qs + 8 * c8 + 4);
            // running softmax of this 16-lane row over the wave's blocks of 64 keys; the sum and the output stay
            // per 8-lane group until the end.  The first block is straight-line code so that each score waits only
            // for its own key's load; later blocks (contexts beyond 256 keys) run in the loop.
            float o8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ls = 0.f, mr;
            float sv[8];
            auto scores = [&](int cb, int nb) -> float {   // fills sv, returns the row's maximum
                float bm = -INFINITY;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    sv[i] = -INFINITY;
                    if (i >= nb) continue;
                    const bf16x8 k8 = kreg[i];
                    float d = ((qa.x * (float)k8[0] + qa.y * (float)k8[1]) + (qa.z * (float)k8[2] + qa.w * (float)k8[3])) +
                              ((qb.x * (float)k8[4] + qb.y * (float)k8[5]) + (qb.z * (float)k8[6] + qb.w * (float)k8[7]));
                    d = dpp_add<0xB1>(d);    // + the other lane of the pair
                    d = dpp_add<0x4E>(d);    // + the other pair of the quad
                    d = dpp_add<0x141>(d);   // + the other quad of the 8 channel lanes
                    sv[i] = kg8 + 8 * i < cb ? d * 0.125f : -INFINITY;
                    bm = fmaxf(bm, sv[i]);
                }
                return fmaxf(bm, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, bm), 0x128, 0xF, 0xF, true)));   // row_ror:8
            };
            auto weigh = [&](int nb) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i >= nb) continue;
                    const float pw = sv[i] == -INFINITY ? 0.f : expf(sv[i] - mr);
                    ls += pw;
                    const bf16x8 v8 = vreg[i];
#pragma unroll
                    for (int c = 0; c < 8; ++c) o8[c] += pw * (float)v8[c];
                }
            };
            {
                const int cb = min(cnt, 64);
                const int nb = __builtin_amdgcn_readfirstlane((cb + 7) >> 3);
                mr = scores(cb, nb);
                weigh(nb);
            }
            for (int c0 = 64; c0 < cnt; c0 += 64) {
                const int cb = min(cnt - c0, 64);
                const int nb = __builtin_amdgcn_readfirstlane((cb + 7) >> 3);
                request_block(c0);
                const float mn = fmaxf(mr, scores(cb, nb));
                // rescale what earlier blocks left (a row without keys so far scales by zero, not by exp(nan))
                const float rs = mr == -INFINITY ? 0.f : expf(mr - mn);
                ls *= rs;
#pragma unroll
                for (int c = 0; c < 8; ++c) o8[c] *= rs;
                mr = mn;
                weigh(nb);
            }
            ls = dpp_add<0x128>(ls);
#pragma unroll
            for (int c = 0; c < 8; ++c) o8[c] = dpp_add<0x128>(o8[c]);
            float* cw = comb + (w * 4 + (lane >> 4)) * 68;
            if ((lane & 15) == 0) { cw[0] = mr; cw[1] = ls; }
            if ((lane & 15) < 8) {
                *(float4*)(cw + 4 + 8 * c8) = make_float4(o8[0], o8[1], o8[2], o8[3]);
                *(float4*)(cw + 4 + 8 * c8 + 4) = make_float4(o8[4], o8[5], o8[6], o8[7]);
            }
            done = true;
        }
    }
    if (!done) {
        float4 q[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) q[i] = *(const float4*)(qs + 4 * i);
        for (int j = lane; j < cnt; j += 64) {
            const long long kr_ = cbase + (long long)((j0 + kb + j) % a.cap) * a.d;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            if constexpr (KV16) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bf16x8 k8 = *(const bf16x8*)((const __bf16*)a.kcache + kr_ + 8 * i);
                    s0 += q[2 * i].x * (float)k8[0]; s1 += q[2 * i].y * (float)k8[1]; s2 += q[2 * i].z * (float)k8[2]; s3 += q[2 * i].w * (float)k8[3];
                    s0 += q[2 * i + 1].x * (float)k8[4]; s1 += q[2 * i + 1].y * (float)k8[5]; s2 += q[2 * i + 1].z * (float)k8[6]; s3 += q[2 * i + 1].w * (float)k8[7];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float4 k4 = *(const float4*)((const float*)a.kcache + kr_ + 4 * i);
                    s0 += q[i].x * k4.x; s1 += q[i].y * k4.y; s2 += q[i].z * k4.z; s3 += q[i].w * k4.w;
                }
            }
            const float sv = ((s0 + s1) + (s2 + s3)) * 0.125f;
            sc[j] = sv;
            mx = fmaxf(mx, sv);
        }
#pragma unroll
        for (int of = 32; of > 0; of >>= 1) mx = fmaxf(mx, __shfl_xor(mx, of, 64));
        for (int j = lane; j < cnt; j += 64) {
            const float e = expf(sc[j] - mx);
            sc[j] = e;
            sum += e;
        }
        sum = wsum64(sum);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        int j = kg;
        for (; j + 12 < cnt; j += 16) {
            const float4 v0 = ld4(a.vcache, cbase + (long long)((j0 + kb + j) % a.cap) * a.d + 4 * cq);
            const float4 v1 = ld4(a.vcache, cbase + (long long)((j0 + kb + j + 4) % a.cap) * a.d + 4 * cq);
            const float4 v2 = ld4(a.vcache, cbase + (long long)((j0 + kb + j + 8) % a.cap) * a.d + 4 * cq);
            const float4 v3 = ld4(a.vcache, cbase + (long long)((j0 + kb + j + 12) % a.cap) * a.d + 4 * cq);
            const float p0 = sc[j], p1 = sc[j + 4], p2 = sc[j + 8], p3 = sc[j + 12];
            o.x += (p0 * v0.x + p1 * v1.x) + (p2 * v2.x + p3 * v3.x);
            o.y += (p0 * v0.y + p1 * v1.y) + (p2 * v2.y + p3 * v3.y);
            o.z += (p0 * v0.z + p1 * v1.z) + (p2 * v2.z + p3 * v3.z);
            o.w += (p0 * v0.w + p1 * v1.w) + (p2 * v2.w + p3 * v3.w);
        }
        for (; j < cnt; j += 4) {
            const float4 v0 = ld4(a.vcache, cbase + (long long)((j0 + kb + j) % a.cap) * a.d + 4 * cq);
            const float p0 = sc[j];
            o.x += p0 * v0.x; o.y += p0 * v0.y; o.z += p0 * v0.z; o.w += p0 * v0.w;
        }
        o.x += __shfl_xor(o.x, 16, 64); o.y += __shfl_xor(o.y, 16, 64); o.z += __shfl_xor(o.z, 16, 64); o.w += __shfl_xor(o.w, 16, 64);
        o.x += __shfl_xor(o.x, 32, 64); o.y += __shfl_xor(o.y, 32, 64); o.z += __shfl_xor(o.z, 32, 64); o.w += __shfl_xor(o.w, 32, 64);
        float* cw = comb + (w * 4) * 68;
        if (lane == 0) { cw[0] = mx; cw[1] = sum; }
        if (lane >= 1 && lane < 4) { cw[lane * 68] = -INFINITY; cw[lane * 68 + 1] = 0.f; }   // this wave's other three partials: empty
        if (kg == 0) *(float4*)(cw + 4 + 4 * cq) = o;
    }
    __syncthreads();
    if (w != 0) return;
    // merge: lane e owns channel e again
    const float s_new = wsum64(qr * kr) * 0.125f;
    float M = s_new;
#pragma unroll
    for (int i = 0; i < 16; ++i) M = fmaxf(M, comb[i * 68]);
    float tot = expf(s_new - M), acc = tot * vv;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float mi = comb[i * 68];
        if (mi != -INFINITY) {   // uniform over the wave; an empty partial's output words are undefined
            const float f = expf(mi - M);
            tot += comb[i * 68 + 1] * f;
            acc += comb[i * 68 + 4 + lane] * f;
        }
    }
    const float out = acc / tot;
    if (a.out_q) {   // a head's 64 channels are one quantisation block of the output product
        float bm = fabsf(out);
#pragma unroll
        for (int of = 32; of > 0; of >>= 1) bm = fmaxf(bm, __shfl_xor(bm, of, 64));
        const float inv = bm > 0.f ? 127.0f / bm : 0.f;
        a.out_q[(long long)m * a.d + h * 64 + lane] = (signed char)(int)rintf(out * inv);
        if (lane == 0) a.out_d[(long long)h * a.Mq + m] = bm / 127.0f;
    } else {
        a.out[(long long)m * a.d + h * 64 + lane] = out;
    }
    if (a.out_mm) {
        float lo = out, hi = out;
#pragma unroll
        for (int of = 32; of > 0; of >>= 1) { lo = fminf(lo, __shfl_xor(lo, of, 64)); hi = fmaxf(hi, __shfl_xor(hi, of, 64)); }
        if (lane == 0) *(float2*)(a.out_mm + ((long long)h * a.M + m) * 2) = make_float2(lo, hi);
    }
}

void launch_attn_decode(const AttnDecodeArgs& a, hipStream_t s) {
    if (a.M <= 0) return;
    if (a.d != a.heads * 64) throw std::runtime_error("attention: head dimension must be 64");
    if (a.out_q && (!a.out_d || a.Mq < a.M)) throw std::runtime_error("attention: int8 output needs its scale buffer");
    const int scap = std::max((a.cap / 4 + 8) & ~3, 64);
    const size_t smem = ((size_t)4 * (64 + scap) + 16 * 68) * sizeof(float);
    if (smem > 64 * 1024) throw std::runtime_error("attention: context capacity too large for the LDS score buffer");
    dim3 grid((unsigned)(a.M * a.heads)), block(256);
    // measured on the 64 x 125 bf16 workload: 6370 -> 6555 RTFx, LM step 0.360 -> 0.345 ms (POCKET_TTS_KV_NT=0 switches it off)
    static const bool nt = [] { const char* e = std::getenv("POCKET_TTS_KV_NT"); return !e || std::atoi(e) != 0; }();
    if (a.kv16 && nt) hipLaunchKernelGGL((attn_decode_kernel<true, true>), grid, block, smem, s, a);
    else if (a.kv16) hipLaunchKernelGGL(attn_decode_kernel<true>, grid, block, smem, s, a);
    else hipLaunchKernelGGL(attn_decode_kernel<false>, grid, block, smem, s, a);
}

}  // namespace ptts
