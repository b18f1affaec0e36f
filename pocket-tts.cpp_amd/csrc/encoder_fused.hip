// First stage of the Mimi voice encoder in one launch: the 1-input-channel causal convolution, and the residual unit
// that follows it at full sample rate (ELU -> conv k3, 64 -> 32 channels -> ELU -> conv k1, 32 -> 64 -> + skip -> ELU),
// /root/reference/src/pocket_tts.cpp:261-291 runs them inside the mimi_encoder graph.
// As separate launches the stage writes and re-reads four [samples][64] float arrays (about 180 MB for a 5 s clip); here
// a workgroup takes 64 consecutive samples of one clip, keeps the 64-channel rows in LDS, and only the unit's output
// goes to memory (31 MB per clip).  The two convolutions of the unit run on bf16 MFMA with both operands as two bf16
// terms (hi hi + lo hi + hi lo, f32 accumulation: f32-grade), like the encoder's other matrices.
#include <hip/hip_runtime.h>

#include <stdexcept>

#include "kernels.h"

namespace ptts {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

constexpr int C0 = 64, C1 = 32, TT = 64, RK = 3, HALO = RK - 1;
constexpr int XP = C0 + 8, HP = C1 + 8;      // LDS pitches of the bf16 rows (16-byte aligned, conflict-poor)
constexpr int kMaxTaps = 16;

__device__ __forceinline__ float elu(float v) { return v > 0.f ? v : expm1f(v); }

__global__ __launch_bounds__(256, 2) void enc_stage0_kernel(EncStage0Args a) {
    __shared__ float aud[TT + HALO + kMaxTaps];
    __shared__ __attribute__((aligned(16))) float xraw[TT + HALO][C0];
    __shared__ __attribute__((aligned(16))) __bf16 xhi[TT + HALO][XP];
    __shared__ __attribute__((aligned(16))) __bf16 xlo[TT + HALO][XP];
    __shared__ __attribute__((aligned(16))) __bf16 hhi[TT][HP];
    __shared__ __attribute__((aligned(16))) __bf16 hlo[TT][HP];
    __shared__ __attribute__((aligned(16))) float outs[TT][C0];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.y, t0 = blockIdx.x * TT;
    const float* au = a.audio + (long long)b * a.a_bs;
    // samples t0 - HALO - (k - 1) .. t0 + TT - 1 of the clip; zero outside it (the causal padding and the clip's end)
    const int na = TT + HALO + a.k - 1, a0 = t0 - HALO - (a.k - 1);
    for (int i = tid; i < na; i += 256) {
        const int src = a0 + i;
        aud[i] = (src >= 0 && src < a.T) ? au[src] : 0.f;
    }
    __syncthreads();
    // rows t0 - HALO .. t0 + TT - 1 of the first convolution: raw (the skip input) and ELU'd as two bf16 terms;
    // rows before the clip's start are the unit's zero padding
    {
        const int co = tid & 63;
        float wt[kMaxTaps];
#pragma unroll
        for (int tap = 0; tap < kMaxTaps; ++tap) wt[tap] = tap < a.k ? a.w0[co * a.k + tap] : 0.f;
        const float bias = a.b0[co];
        for (int rr = tid >> 6; rr < TT + HALO; rr += 4) {
            float v = bias;
#pragma unroll
            for (int tap = 0; tap < kMaxTaps; ++tap)
                if (tap < a.k) v += wt[tap] * aud[rr + tap];
            xraw[rr][co] = v;
            const float e = (t0 - HALO + rr < 0) ? 0.f : elu(v);
            const __bf16 hi = (__bf16)e;
            xhi[rr][co] = hi;
            xlo[rr][co] = (__bf16)(e - (float)hi);
        }
    }
    __syncthreads();
    // From here every wave works on its own 16 rows (local rows 16 w .. 16 w + 15).
    const int r = lane & 15, g = lane >> 4, row0 = 16 * w;
    // ---- conv k3, 64 -> 32: K = 3 taps x 64 channels = 6 k-steps of 32; output row t reads input rows t .. t + 2
    {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < RK * C0 / 32; ++ks) {
            const int tap = ks >> 1, ci0 = (ks & 1) * 32;
            const bf16x8 ahi = *(const bf16x8*)&xhi[row0 + r + tap][ci0 + 8 * g];
            const bf16x8 alo = *(const bf16x8*)&xlo[row0 + r + tap][ci0 + 8 * g];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const long long wo = (long long)(nt * 16 + r) * (RK * C0) + ks * 32 + 8 * g;
                const bf16x8 bhi = *(const bf16x8*)((const __bf16*)a.w1_hi + wo);
                const bf16x8 blo = *(const bf16x8*)((const __bf16*)a.w1_lo + wo);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, bhi, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, blo, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, bhi, acc[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float bias = a.b1[nt * 16 + r];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float e = elu(acc[nt][i] + bias);
                const __bf16 hi = (__bf16)e;
                hhi[row0 + 4 * g + i][nt * 16 + r] = hi;
                hlo[row0 + 4 * g + i][nt * 16 + r] = (__bf16)(e - (float)hi);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // ---- conv k1, 32 -> 64, + the raw row of the same sample, ELU
    {
        const bf16x8 ahi = *(const bf16x8*)&hhi[row0 + r][8 * g];
        const bf16x8 alo = *(const bf16x8*)&hlo[row0 + r][8 * g];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            const long long wo = (long long)(nt * 16 + r) * C1 + 8 * g;
            const bf16x8 bhi = *(const bf16x8*)((const __bf16*)a.w2_hi + wo);
            const bf16x8 blo = *(const bf16x8*)((const __bf16*)a.w2_lo + wo);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, bhi, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, blo, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, bhi, acc, 0, 0, 0);
            const float bias = a.b2[nt * 16 + r];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * g + i;
                outs[row][nt * 16 + r] = elu(acc[i] + bias + xraw[row + HALO][nt * 16 + r]);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // ---- the wave's 16 rows x 64 channels, 16 bytes a lane
    float* out = a.out + (long long)b * a.out_bs;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = it * 64 + lane, row = row0 + (idx >> 4), c4 = idx & 15;
        if (t0 + row < a.T) *(float4*)(out + (long long)(t0 + row) * C0 + 4 * c4) = *(const float4*)&outs[row][4 * c4];
    }
}

}  // namespace

bool enc_stage0_supported(const EncStage0Args& a) {
    return a.c0 == C0 && a.c1 == C1 && a.rk == RK && a.k >= 1 && a.k <= kMaxTaps && a.w1_hi && a.w1_lo && a.w2_hi && a.w2_lo &&
           ((uintptr_t)a.out & 15) == 0 && (a.out_bs & 3) == 0;
}

void launch_enc_stage0(const EncStage0Args& a, hipStream_t s) {
    if (a.T <= 0 || a.B <= 0) return;
    if (!enc_stage0_supported(a)) throw std::runtime_error("encoder stage 0: shape not supported (64 / 32 channels, kernel 3, two-term weights)");
    hipLaunchKernelGGL(enc_stage0_kernel, dim3((a.T + TT - 1) / TT, a.B), dim3(256), 0, s, a);
}

}  // namespace ptts
