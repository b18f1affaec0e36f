// Fused LSD flow-matching head: all lsd_steps Euler iterations of the flow MLP in ONE launch.
//
// The reference runs the flow graph lsd_steps times per frame through ONNX Runtime and applies
// x += v * dt on the host between calls (/root/reference/src/pocket_tts.cpp:414-451, 615-622).  Here a
// workgroup owns 16 utterance rows and carries them through every layer of every Euler step with the
// residual stream, the latent and the MFMA A-operands resident in LDS; nothing but weights is read
// from memory inside the loop and there is no inter-workgroup synchronisation.
//
// Work split: 4 waves; every GEMM [16 x K] x [K x N] is cut into 16-column MFMA tiles dealt round-robin
// to the waves.  Weights stream L2 -> VGPR (each wave reads a different slice, so an LDS round trip
// would add nothing); f32 weights use v_mfma_f32_16x16x4_f32, bf16 weights v_mfma_f32_16x16x32_bf16.
// The adaLN shift/scale/gate vectors depend only on the conditioning and the Euler step index, never
// on x, so they are produced for all steps by one batched GEMM before this kernel (see engine.cpp).
#include <hip/hip_runtime.h>

#include <stdexcept>
#include <type_traits>

#include "kernels.h"

namespace ptts {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <typename WT> struct FlowPad { static constexpr int value = 4; };
template <> struct FlowPad<__bf16> { static constexpr int value = 8; };

// acc (16 rows x 16 cols n0..n0+15) = A[16][K] (LDS, row stride lda) * W[n][K]^T (global)
template <typename WT>
__device__ __forceinline__ f32x4 tile_mm(const WT* __restrict__ A, int lda, const WT* __restrict__ W, int K, int n0, int lane) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, g = lane >> 4;
    if constexpr (std::is_same<WT, float>::value) {
        const float* ap = A + r * lda + 4 * g;
        const float* wp = W + (long long)(n0 + r) * K + 4 * g;
#pragma unroll 4
        for (int k0 = 0; k0 < K; k0 += 16) {
            const float4 a4 = *(const float4*)(ap + k0);
            const float4 b4 = *(const float4*)(wp + k0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc, 0, 0, 0);
        }
    } else {
        const __bf16* ap = A + r * lda + 8 * g;
        const __bf16* wp = W + (long long)(n0 + r) * K + 8 * g;
#pragma unroll 8
        for (int k0 = 0; k0 < K; k0 += 32) {
            const bf16x8 a8 = *(const bf16x8*)(ap + k0);
            const bf16x8 b8 = *(const bf16x8*)(wp + k0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc, 0, 0, 0);
        }
    }
    return acc;
}

__device__ __forceinline__ float silu_dev(float v) { return v / (1.0f + expf(-v)); }

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// LayerNorm (optional affine) + adaLN modulate of 16 rows of h -> A-operand buffer; wave w owns rows 4w..4w+3.
template <typename WT>
__device__ __forceinline__ void ln_modulate(const float* h, int C, WT* dst, int ldd, const float* g, const float* b,
                                            const float* mods, long long mod_row_stride, int row0, int nrows, int shift_off,
                                            int scale_off, float eps, int wave, int lane) {
    for (int rr = 0; rr < 4; ++rr) {
        const int row = wave * 4 + rr;
        const float* hr = h + row * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += hr[c];
        const float mean = wsum(s) / (float)C;
        float q = 0.f;
        for (int c = lane; c < C; c += 64) { float d = hr[c] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wsum(q) / (float)C + eps);
        const int grow = row < nrows ? row0 + row : row0 + nrows - 1;
        const float* mr = mods + (long long)grow * mod_row_stride;
        for (int c = lane; c < C; c += 64) {
            float v = (hr[c] - mean) * rstd;
            if (g) v = v * g[c] + b[c];
            v = v * (1.0f + mr[scale_off + c]) + mr[shift_off + c];
            dst[row * ldd + c] = (WT)v;
        }
    }
}

template <typename WT>
__global__ __launch_bounds__(256) void flow_head_kernel(FlowHeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int PAD = FlowPad<WT>::value;
    const int C = a.C, L = a.L;
    const int lda = C + PAD, ldx = L + PAD;
    float* h = (float*)smem_raw;             // [16][C]   residual stream
    float* xs = h + 16 * C;                  // [16][L]   latent
    WT* abuf0 = (WT*)(xs + 16 * L);          // [16][lda] LN/modulate output
    WT* abuf1 = abuf0 + 16 * lda;            // [16][lda] fc1 output
    WT* abx = abuf1 + 16 * lda;              // [16][ldx] latent as A operand

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * 16;
    const int nrows = (a.B - row0) < 16 ? (a.B - row0) : 16;
    const int crow = (lane >> 4) * 4, ccol = lane & 15;  // accumulator rows crow..crow+3, column ccol of a tile
    const float dt = 1.0f / (float)a.steps;

    for (int i = tid; i < 16 * L; i += 256) {
        int r = i / L;
        xs[i] = r < nrows ? a.x0[(long long)(row0 + r) * a.x0_stride + (i % L)] : 0.f;
    }
    __syncthreads();

    for (int j = 0; j < a.steps; ++j) {
        const float* mods_j = a.mods + (long long)j * a.B * a.mod_width;
        for (int i = tid; i < 16 * L; i += 256) abx[(i / L) * ldx + (i % L)] = (WT)xs[i];
        __syncthreads();
        // input projection: h = x W_in^T + b_in
        for (int nt = wave; nt < C / 16; nt += 4) {
            f32x4 acc = tile_mm<WT>(abx, ldx, (const WT*)a.w_in, L, nt * 16, lane);
            const int col = nt * 16 + ccol;
            const float bias = a.b_in[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) h[(crow + r) * C + col] = acc[r] + bias;
        }
        __syncthreads();
        for (int blk = 0; blk < a.blocks; ++blk) {
            const int moff = blk * 3 * C;
            ln_modulate<WT>(h, C, abuf0, lda, a.ln_g[blk], a.ln_b[blk], mods_j, a.mod_width, row0, nrows, moff, moff + C, a.eps, wave, lane);
            __syncthreads();
            const WT* w1 = (const WT*)a.w_fc1[blk];
            const float* b1 = a.b_fc1[blk];
            for (int nt = wave; nt < C / 16; nt += 4) {
                f32x4 acc = tile_mm<WT>(abuf0, lda, w1, C, nt * 16, lane);
                const int col = nt * 16 + ccol;
                const float bias = b1[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) abuf1[(crow + r) * lda + col] = (WT)silu_dev(acc[r] + bias);
            }
            __syncthreads();
            const WT* w2 = (const WT*)a.w_fc2[blk];
            const float* b2 = a.b_fc2[blk];
            for (int nt = wave; nt < C / 16; nt += 4) {
                f32x4 acc = tile_mm<WT>(abuf1, lda, w2, C, nt * 16, lane);
                const int col = nt * 16 + ccol;
                const float bias = b2[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = crow + r;
                    const int grow = row < nrows ? row0 + row : row0 + nrows - 1;
                    const float gate = mods_j[(long long)grow * a.mod_width + moff + 2 * C + col];
                    h[row * C + col] += gate * (acc[r] + bias);
                }
            }
            __syncthreads();
        }
        const int foff = a.blocks * 3 * C;
        ln_modulate<WT>(h, C, abuf0, lda, nullptr, nullptr, mods_j, a.mod_width, row0, nrows, foff, foff + C, a.eps, wave, lane);
        __syncthreads();
        for (int nt = wave; nt < L / 16; nt += 4) {
            f32x4 acc = tile_mm<WT>(abuf0, lda, (const WT*)a.w_out, C, nt * 16, lane);
            const int col = nt * 16 + ccol;
            const float bias = a.b_out[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) xs[(crow + r) * L + col] += (acc[r] + bias) * dt;
        }
        __syncthreads();
    }
    for (int i = tid; i < 16 * L; i += 256) {
        int r = i / L;
        if (r < nrows) a.x_out[(long long)(row0 + r) * L + (i % L)] = xs[i];
    }
}

void launch_flow_head(const FlowHeadArgs& a, hipStream_t s) {
    if (a.B <= 0) return;
    if (a.C % 64 || a.L % 32 || a.steps <= 0) throw std::runtime_error("flow head: flow_dim must be a multiple of 64 and latent_dim of 32");
    const int wbytes = a.wtype == W_BF16 ? 2 : 4;
    const int pad = a.wtype == W_BF16 ? 8 : 4;
    size_t smem = (size_t)16 * a.C * 4 + (size_t)16 * a.L * 4 + (size_t)2 * 16 * (a.C + pad) * wbytes + (size_t)16 * (a.L + pad) * wbytes;
    dim3 grid((a.B + 15) / 16), block(256);
    if (a.wtype == W_BF16) {
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)flow_head_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(flow_head_kernel<__bf16>, grid, block, smem, s, a);
    } else {
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)flow_head_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(flow_head_kernel<float>, grid, block, smem, s, a);
    }
}

}  // namespace ptts
