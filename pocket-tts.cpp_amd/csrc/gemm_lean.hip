// bf16-weight GEMM for the shapes the Mimi decoder / encoder and the LM prefill are made of (M in the thousands to
// millions, N and K a few hundred): C[m][n] = epilogue(sum_k act_a(A[m][k]) * W[n][k]) with the epilogue subset
//   v = acc + bias[n % bias_mod];  v *= colscale[n] * alpha;  v += R[m][n];  dead rows -> 0;  C = c_act(v);  C2 = c2_act(v)
// At these shapes the K loop is short and the tile's loads and stores are most of the work, so the kernel is built
// around them: 128 x 128 tile, K in steps of 64 (half the barriers of a 32-deep step), 16-byte global loads, and
// the MFMA operands swapped (weights as A, activations as B) so that the accumulator holds C transposed: every
// lane owns four consecutive n of one row m, hence 16-byte bias / residual loads and 16-byte (f32) or 8-byte
// (bf16) stores, and only two row-address computations per lane.  A rows and C rows may be bf16 in memory.
// A is rounded to bf16 once while staged, or (SPLIT) carried as two bf16 terms.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <stdexcept>

#include "kernels.h"

namespace ptts {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        case ACT_SILU: return v / (1.0f + expf(-v));
        case ACT_ELU: return v > 0.f ? v : expm1f(v);
        default: return v;
    }
}

constexpr int LBM = 128, LBN = 128, LBK = 64, LLD = LBK + 8;

// SPLIT (f32 A only): A is staged as two bf16 terms, hi = bf16(a) and lo = bf16(a - hi), and every weight fragment
// meets both, so the product carries the activation to ~16 mantissa bits (the latent path's prefill).
// WSPLIT (with SPLIT): the weights come as two bf16 terms as well (a.W, a.W2); the third MFMA adds A_hi W_lo.
// OCC: workgroups per CU the register allocation is held to (3: 168 registers per lane)
template <bool ABF, bool CBF, bool SPLIT, bool WSPLIT = false, int OCC = 2>
__global__ __launch_bounds__(256, OCC) void gemm_lean_kernel(GemmArgs a, int gw) {
    __shared__ __attribute__((aligned(16))) __bf16 As[LBM * LLD];
    __shared__ __attribute__((aligned(16))) __bf16 Al[SPLIT ? LBM * LLD : 8];
    __shared__ __attribute__((aligned(16))) __bf16 Ws[LBN * LLD];
    __shared__ __attribute__((aligned(16))) __bf16 Wl[WSPLIT ? LBN * LLD : 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int M = a.M, N = a.N, K = a.K;
    // Tile of this workgroup.  gw == 0: blockIdx (x, y) = (m tile, n tile).  gw > 0: a one-dimensional grid laid out for
    // the eight XCDs, each with an L2 of its own: workgroup L is dispatched to XCD L % 8, so the sequence L / 8 is what
    // one XCD sees in order.  That sequence walks groups of gw n tiles (weights of a group stay in the XCD's L2) and,
    // inside a group, every n tile of an m tile before the next m tile: the re-reads of an A tile come from the workgroups
    // that follow each other on the same XCD and hit its L2 instead of going out to the fabric.
    int mt = blockIdx.x, nt = blockIdx.y;
    if (gw > 0) {
        const int MT = (M + LBM - 1) / LBM, NT = (N + LBN - 1) / LBN, MT8 = (MT + 7) >> 3;
        const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3;
        const int g = s / (MT8 * gw), r = s - g * (MT8 * gw);
        const int width = min(gw, NT - g * gw);
        mt = (r / width) * 8 + xcd;
        nt = g * gw + r % width;
        if (mt >= MT) return;
    }
    const int m0 = mt * LBM, n0 = nt * LBN;

    // ---- staging assignment: A as 16 (f32) or 8 (bf16) pieces per row of a K step, W as 8 pieces per row
    constexpr int APR = ABF ? 8 : 16, APT = LBM * APR / 256;     // pieces per row, pieces per thread
    constexpr int AEL = ABF ? 8 : 4;                             // elements per piece
    const char* aptr[APT];
    int ak[APT];
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int id = tid + 256 * i, row = id / APR, p = id % APR;
        const int m = m0 + row;
        ak[i] = p * AEL;
        aptr[i] = nullptr;
        if (m < M) {
            const long long off = (long long)(m / a.rpb) * a.a_bs + (long long)(m % a.rpb) * a.lda + p * AEL;
            aptr[i] = (const char*)a.A + off * (ABF ? 2 : 4);
        }
    }
    constexpr int WPT = LBN * 8 / 256;
    const __bf16* wptr[WPT];
    int wk[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int id = tid + 256 * i, row = id >> 3, p = id & 7;
        wk[i] = p * 8;
        wptr[i] = (n0 + row < N) ? (const __bf16*)a.W + (long long)(n0 + row) * K + p * 8 : nullptr;
    }
    uint4 areg[APT], wreg[WPT], wreg2[WSPLIT ? WPT : 1];
    const long long w2_off = WSPLIT ? (const char*)a.W2 - (const char*)a.W : 0;
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < APT; ++i)
            areg[i] = (aptr[i] && k0 + ak[i] < K) ? *(const uint4*)(aptr[i] + (long long)k0 * (ABF ? 2 : 4)) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            wreg[i] = (wptr[i] && k0 + wk[i] < K) ? *(const uint4*)(wptr[i] + k0) : make_uint4(0u, 0u, 0u, 0u);
        if constexpr (WSPLIT) {
#pragma unroll
            for (int i = 0; i < WPT; ++i)
                wreg2[i] = (wptr[i] && k0 + wk[i] < K) ? *(const uint4*)((const char*)(wptr[i] + k0) + w2_off) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int id = tid + 256 * i, row = id / APR, p = id % APR;
            if constexpr (ABF) {
                if (a.a_act != ACT_NONE) {
                    bf16x8 v = __builtin_bit_cast(bf16x8, areg[i]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (__bf16)act_apply((float)v[e], a.a_act);
                    *(bf16x8*)&As[row * LLD + p * 8] = v;
                } else {
                    *(uint4*)&As[row * LLD + p * 8] = areg[i];
                }
            } else {
                float4 v = __builtin_bit_cast(float4, areg[i]);
                if (a.a_act != ACT_NONE) {
                    v.x = act_apply(v.x, a.a_act); v.y = act_apply(v.y, a.a_act); v.z = act_apply(v.z, a.a_act); v.w = act_apply(v.w, a.a_act);
                }
                bf16x4 q;
                q[0] = (__bf16)v.x; q[1] = (__bf16)v.y; q[2] = (__bf16)v.z; q[3] = (__bf16)v.w;
                *(bf16x4*)&As[row * LLD + p * 4] = q;
                if constexpr (SPLIT) {
                    bf16x4 lo;
                    lo[0] = (__bf16)(v.x - (float)q[0]); lo[1] = (__bf16)(v.y - (float)q[1]);
                    lo[2] = (__bf16)(v.z - (float)q[2]); lo[3] = (__bf16)(v.w - (float)q[3]);
                    *(bf16x4*)&Al[row * LLD + p * 4] = lo;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int id = tid + 256 * i, row = id >> 3, p = id & 7;
            *(uint4*)&Ws[row * LLD + p * 8] = wreg[i];
            if constexpr (WSPLIT) *(uint4*)&Wl[row * LLD + p * 8] = wreg2[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    load_tiles(0);
    for (int k0 = 0; k0 < K; k0 += LBK) {
        store_tiles();
        __syncthreads();
        if (k0 + LBK < K) load_tiles(k0 + LBK);
#pragma unroll
        for (int ks = 0; ks < LBK / 16; ++ks) {
            bf16x8 af[2], al[2], wf[2], wl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *(const bf16x8*)&As[(wr * 64 + i * 32 + lr) * LLD + ks * 16 + lh * 8];
                if constexpr (SPLIT) al[i] = *(const bf16x8*)&Al[(wr * 64 + i * 32 + lr) * LLD + ks * 16 + lh * 8];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                wf[j] = *(const bf16x8*)&Ws[(wc * 64 + j * 32 + lr) * LLD + ks * 16 + lh * 8];
                if constexpr (WSPLIT) wl[j] = *(const bf16x8*)&Wl[(wc * 64 + j * 32 + lr) * LLD + ks * 16 + lh * 8];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (WSPLIT) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[j], af[i], acc[i][j], 0, 0, 0);
                    if constexpr (SPLIT) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], al[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);   // C^T tile
                }
        }
        __syncthreads();
    }

    // ---- epilogue: lane = row m (lr), registers = n: 8 * (r >> 2) + 4 * lh + (r & 3)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wr * 64 + i * 32 + lr;
        if (m >= M) continue;
        const int b = m / a.rpb, t = m % a.rpb;
        const bool dead = a.row_len && t >= a.row_len[b];
        const long long crow = (long long)b * a.c_bs + (long long)t * a.ldc;
        const long long c2row = (long long)b * a.c2_bs + (long long)t * a.ldc2;
        const float* rrow = a.R ? a.R + (long long)b * a.r_bs + (long long)t * a.ldr : nullptr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wc * 64 + j * 32 + 8 * g + 4 * lh;
                if (n >= N) continue;
                float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                if (a.bias) {
                    const float4 bb = *(const float4*)(a.bias + (a.bias_mod ? n % a.bias_mod : n));
                    v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                }
                if (a.colscale) {
                    const float4 cs = *(const float4*)(a.colscale + n);
                    v.x *= cs.x * a.alpha; v.y *= cs.y * a.alpha; v.z *= cs.z * a.alpha; v.w *= cs.w * a.alpha;
                } else if (a.alpha != 1.0f) {
                    v.x *= a.alpha; v.y *= a.alpha; v.z *= a.alpha; v.w *= a.alpha;
                }
                if (rrow) {
                    const float4 rr = *(const float4*)(rrow + n);
                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
                if (dead) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.C) {
                    float4 o = v;
                    if (a.c_act != ACT_NONE) { o.x = act_apply(o.x, a.c_act); o.y = act_apply(o.y, a.c_act); o.z = act_apply(o.z, a.c_act); o.w = act_apply(o.w, a.c_act); }
                    if constexpr (CBF) {
                        bf16x4 q;
                        q[0] = (__bf16)o.x; q[1] = (__bf16)o.y; q[2] = (__bf16)o.z; q[3] = (__bf16)o.w;
                        *(bf16x4*)((__bf16*)a.C + crow + n) = q;
                    } else {
                        *(float4*)(a.C + crow + n) = o;
                    }
                }
                if (a.C2) {
                    float4 o = v;
                    if (a.c2_act != ACT_NONE) { o.x = act_apply(o.x, a.c2_act); o.y = act_apply(o.y, a.c2_act); o.z = act_apply(o.z, a.c2_act); o.w = act_apply(o.w, a.c2_act); }
                    *(float4*)(a.C2 + c2row + n) = o;
                }
            }
        }
    }
}

}  // namespace

// Whether launch_gemm may hand this problem to the lean kernel.
bool gemm_lean_eligible(const GemmArgs& a) {
    if (a.wtype != W_BF16 || a.G || a.act_pre != ACT_NONE) return false;
    if (a.split_a && (a.a_bf16 || a.c_bf16)) return false;
    if (a.W2 && (!a.split_a || ((uintptr_t)a.W2 & 15))) return false;
    // small problems: the 64 x 64 tiles of the general kernel fill the chip better (bf16 rows and two-term weights exist only here)
    if (!(a.a_bf16 || a.c_bf16 || a.W2) && (a.M < 512 || a.N < 32)) return false;
    if ((a.N & 3) || (a.K & 7) || (a.ldc & 3) || (a.c_bs & 3)) return false;
    const int ael = a.a_bf16 ? 8 : 4;
    if ((a.lda % ael) || (a.a_bs % ael) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15)) return false;
    if (a.C && ((uintptr_t)a.C & (a.c_bf16 ? 7 : 15))) return false;
    if (a.C2 && (((uintptr_t)a.C2 & 15) || (a.ldc2 & 3) || (a.c2_bs & 3))) return false;
    if (a.bias && (((uintptr_t)a.bias & 15) || (a.bias_mod & 3))) return false;
    if (a.colscale && ((uintptr_t)a.colscale & 15)) return false;
    if (a.R && (((uintptr_t)a.R & 15) || (a.ldr & 3) || (a.r_bs & 3))) return false;
    return true;
}

void launch_gemm_lean(const GemmArgs& a, hipStream_t s) {
    static const int xcd_map = [] { const char* e = std::getenv("POCKET_TTS_GEMM_XCD_MAP"); return e ? std::atoi(e) : 1; }();
    const int MT = (a.M + LBM - 1) / LBM, NT = (a.N + LBN - 1) / LBN;
    // n tiles per group: about 2 MiB of bf16 weights (half of one XCD's L2), at least 2
    int gw = 0;
    // (an XCD takes whole m tiles: with few m tiles that are not a multiple of 8 some XCDs would get a tile less or none)
    if (xcd_map && NT > 1 && (((MT + 7) / 8) * 8 - MT) * 16 <= MT) gw = std::min(NT, std::max(2, (2 << 20) / (LBN * a.K * 2 * (a.W2 ? 2 : 1))));
    dim3 grid(MT, NT), block(256);
    if (gw) grid = dim3(((MT + 7) / 8) * 8 * NT, 1);
    if (a.split_a && a.W2) {
        hipLaunchKernelGGL((gemm_lean_kernel<false, false, true, true>), grid, block, 0, s, a, gw);
    } else if (a.split_a) {
        hipLaunchKernelGGL((gemm_lean_kernel<false, false, true>), grid, block, 0, s, a, gw);
    } else if (a.a_bf16) {
        if (a.c_bf16) hipLaunchKernelGGL((gemm_lean_kernel<true, true, false>), grid, block, 0, s, a, gw);
        else hipLaunchKernelGGL((gemm_lean_kernel<true, false, false>), grid, block, 0, s, a, gw);
    } else {
        // f32 rows in, K of at most 1024: three workgroups per CU (168 registers, 28 bytes of scratch per lane) hide more of
        // the load latency than two; with longer K loops the spilled words cost more than that buys (measured per shape).
        static const int occ3 = [] { const char* e = std::getenv("POCKET_TTS_GEMM_OCC3"); return e ? std::atoi(e) : 1; }();
        if (occ3 && a.K <= 1024) {
            if (a.c_bf16) hipLaunchKernelGGL((gemm_lean_kernel<false, true, false, false, 3>), grid, block, 0, s, a, gw);
            else hipLaunchKernelGGL((gemm_lean_kernel<false, false, false, false, 3>), grid, block, 0, s, a, gw);
        } else
        if (a.c_bf16) hipLaunchKernelGGL((gemm_lean_kernel<false, true, false>), grid, block, 0, s, a, gw);
        else hipLaunchKernelGGL((gemm_lean_kernel<false, false, false>), grid, block, 0, s, a, gw);
    }
}

}  // namespace ptts
