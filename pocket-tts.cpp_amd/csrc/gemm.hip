// LDS-tiled MFMA GEMM with a fused prologue/epilogue; the workhorse behind every linear layer and,
// through overlapping A rows, every Conv1d / ConvTranspose1d of the Mimi codec (direct convolution:
// the k-tap window of channels-last rows is already a contiguous K = k*Cin operand row).
//
// gfx950 mapping: 256 threads = 4 waves in a 2x2 grid; each wave owns (BM/2)x(BN/2) of the block tile
// as 32x32 MFMA tiles.  f32 weights use v_mfma_f32_32x32x2_f32 (an exact f32 FMA chain), bf16 weights
// use v_mfma_f32_32x32x16_bf16 with activations converted to bf16 while they are staged into LDS: either
// rounded once (SPLIT = false) or as the two-term sum hi + lo (SPLIT = true, two MFMAs per tile), which
// keeps ~16 mantissa bits of the activation so that only the weights' own bf16 storage rounds.
// Global loads for tile k+1 are issued before the MFMAs of tile k and written to LDS after them.
#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>
#include <type_traits>

#include "kernels.h"

namespace ptts {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        case ACT_SILU: return v / (1.0f + expf(-v));
        case ACT_ELU: return v > 0.f ? v : expm1f(v);
        default: return v;
    }
}

constexpr int BK = 32;

template <typename WT, int BM, int BN, bool SPLIT = false>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
    constexpr bool kBf16 = !__is_same(WT, float);
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int LDSK = kBf16 ? (BK + 8) : (BK + 1);  // element stride of one LDS row
    using LT = typename std::conditional<kBf16, __bf16, float>::type;
    __shared__ __attribute__((aligned(16))) LT As[BM * LDSK];
    __shared__ __attribute__((aligned(16))) LT Al[SPLIT ? BM * LDSK : 8];  // lo terms
    __shared__ __attribute__((aligned(16))) LT Ws[BN * LDSK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int M = a.M, N = a.N, K = a.K;

    // ---- staging assignment -------------------------------------------------------------------
    constexpr int AR = BM / 32;  // A rows per thread (float4 each)
    const int a_c4 = tid & 7, a_r = tid >> 3;
    const float* arow[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        int m = m0 + a_r + 32 * i;
        arow[i] = nullptr;
        if (m < M) arow[i] = a.A + (long long)(m / a.rpb) * a.a_bs + (long long)(m % a.rpb) * a.lda;
    }
    constexpr int WR = kBf16 ? BN / 64 : BN / 32;  // W rows per thread (16 B each)
    const int w_c = kBf16 ? (tid & 3) : (tid & 7), w_r = kBf16 ? (tid >> 2) : (tid >> 3);
    constexpr int W_RSTEP = kBf16 ? 64 : 32, W_CW = kBf16 ? 8 : 4;
    const WT* wrow[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        int n = n0 + w_r + W_RSTEP * i;
        wrow[i] = (n < N) ? (const WT*)a.W + (long long)n * K : nullptr;
    }

    float4 areg[AR];
    uint4 wreg[WR];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            int k = k0 + a_c4 * 4;
            areg[i] = (arow[i] && k < K) ? *(const float4*)(arow[i] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            int k = k0 + w_c * W_CW;
            wreg[i] = (wrow[i] && k < K) ? *(const uint4*)(wrow[i] + k) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            float4 v = areg[i];
            if (a.a_act != ACT_NONE) {
                v.x = apply_act(v.x, a.a_act); v.y = apply_act(v.y, a.a_act);
                v.z = apply_act(v.z, a.a_act); v.w = apply_act(v.w, a.a_act);
            }
            int row = a_r + 32 * i;
            if constexpr (kBf16) {
                bf16x4 p;
                p[0] = (__bf16)v.x; p[1] = (__bf16)v.y; p[2] = (__bf16)v.z; p[3] = (__bf16)v.w;
                *(bf16x4*)&As[row * LDSK + a_c4 * 4] = p;
                if constexpr (SPLIT) {
                    bf16x4 q;
                    q[0] = (__bf16)(v.x - (float)p[0]); q[1] = (__bf16)(v.y - (float)p[1]);
                    q[2] = (__bf16)(v.z - (float)p[2]); q[3] = (__bf16)(v.w - (float)p[3]);
                    *(bf16x4*)&Al[row * LDSK + a_c4 * 4] = q;
                }
            } else {
                float* d = (float*)&As[row * LDSK + a_c4 * 4];
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            int row = w_r + W_RSTEP * i;
            if constexpr (kBf16) {
                *(uint4*)&Ws[row * LDSK + w_c * 8] = wreg[i];
            } else {
                float* d = (float*)&Ws[row * LDSK + w_c * 4];
                d[0] = __uint_as_float(wreg[i].x); d[1] = __uint_as_float(wreg[i].y);
                d[2] = __uint_as_float(wreg[i].z); d[3] = __uint_as_float(wreg[i].w);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    load_tiles(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        store_tiles();
        __syncthreads();
        if (k0 + BK < K) load_tiles(k0 + BK);
        if constexpr (kBf16) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 af[TM], al[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    af[i] = *(const bf16x8*)&As[(wr * TM * 32 + i * 32 + lr) * LDSK + ks * 16 + lh * 8];
                    if constexpr (SPLIT) al[i] = *(const bf16x8*)&Al[(wr * TM * 32 + i * 32 + lr) * LDSK + ks * 16 + lh * 8];
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[j] = *(const bf16x8*)&Ws[(wc * TN * 32 + j * 32 + lr) * LDSK + ks * 16 + lh * 8];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (SPLIT) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bf[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                float af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = As[(wr * TM * 32 + i * 32 + lr) * LDSK + 2 * kk + lh];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = Ws[(wc * TN * 32 + j * 32 + lr) * LDSK + 2 * kk + lh];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue -----------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int m = m0 + wr * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m >= M) continue;
            int b = m / a.rpb, t = m % a.rpb;
            bool dead = a.row_len && t >= a.row_len[b];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int n = n0 + wc * TN * 32 + j * 32 + lr;
                if (n >= N) continue;
                float v = acc[i][j][r];
                if (a.bias) v += a.bias[a.bias_mod ? (n % a.bias_mod) : n];
                v = apply_act(v, a.act_pre);
                if (a.G) v *= a.G[(long long)b * a.g_bs + (long long)t * a.ldg + n];
                float sc = a.alpha;
                if (a.colscale) sc *= a.colscale[n];
                v *= sc;
                if (a.R) v += a.R[(long long)b * a.r_bs + (long long)t * a.ldr + n];
                if (dead) v = 0.f;
                if (a.C) a.C[(long long)b * a.c_bs + (long long)t * a.ldc + n] = apply_act(v, a.c_act);
                if (a.C2) a.C2[(long long)b * a.c2_bs + (long long)t * a.ldc2 + n] = apply_act(v, a.c2_act);
            }
        }
    }
}

bool gemm_lean_eligible(const GemmArgs& a);
void launch_gemm_lean(const GemmArgs& a, hipStream_t s);

void launch_gemm(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return;
    if (a.K > 0 && a.rpb > 0) {
        GemmArgs l = a;
        // an activation before an otherwise empty epilogue tail is the same as one on the stored value
        if (l.act_pre != ACT_NONE && !l.G && !l.colscale && l.alpha == 1.0f && !l.R && l.c_act == ACT_NONE && !l.C2) {
            l.c_act = l.act_pre;
            l.act_pre = ACT_NONE;
        }
        if (gemm_lean_eligible(l)) { launch_gemm_lean(l, s); return; }
    }
    if (a.a_bf16 || a.c_bf16) throw std::runtime_error("gemm: bf16 activation rows are only handled by the lean kernel (shape or options not eligible)");
    if (a.W2) throw std::runtime_error("gemm: two-term bf16 weights are only handled by the lean kernel (shape or options not eligible)");
    if (a.K <= 0 || (a.K & 3) || (a.lda & 3) || (a.a_bs & 3) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15))
        throw std::runtime_error("gemm: K, lda and batch stride must be multiples of 4 floats and pointers 16-byte aligned");
    if (a.wtype == W_BF16 && (a.K & 7)) throw std::runtime_error("gemm: bf16 weights need K % 8 == 0");
    if (a.rpb <= 0) throw std::runtime_error("gemm: rows per batch must be positive");
    // 128 x 128 tiles only when they still give most CUs a workgroup; a short product (one utterance's chunk in the
    // decoder) runs on 64 x 64 tiles, four times as many workgroups
    const bool big = a.M > 256 && a.N >= 128 && (long long)((a.M + 127) / 128) * ((a.N + 127) / 128) >= 192;
    dim3 block(256);
    if (a.wtype == W_BF16) {
        if (big) {
            dim3 grid((a.M + 127) / 128, (a.N + 127) / 128);
            if (a.split_a) hipLaunchKernelGGL((gemm_kernel<__bf16, 128, 128, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((gemm_kernel<__bf16, 128, 128, false>), grid, block, 0, s, a);
        } else {
            dim3 grid((a.M + 63) / 64, (a.N + 63) / 64);
            if (a.split_a) hipLaunchKernelGGL((gemm_kernel<__bf16, 64, 64, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((gemm_kernel<__bf16, 64, 64, false>), grid, block, 0, s, a);
        }
    } else if (a.wtype == W_F32) {
        if (big) {
            dim3 grid((a.M + 127) / 128, (a.N + 127) / 128);
            hipLaunchKernelGGL((gemm_kernel<float, 128, 128>), grid, block, 0, s, a);
        } else {
            dim3 grid((a.M + 63) / 64, (a.N + 63) / 64);
            hipLaunchKernelGGL((gemm_kernel<float, 64, 64>), grid, block, 0, s, a);
        }
    } else {
        throw std::runtime_error("gemm: unknown weight type");
    }
}

}  // namespace ptts
