// Fused LSD flow-matching head: all lsd_steps Euler iterations of the flow MLP in ONE launch.
//
// The reference runs the flow graph lsd_steps times per frame through ONNX Runtime and applies
// x += v * dt on the host between calls (/root/reference/src/pocket_tts.cpp:414-451, 615-622).  Here a
// workgroup owns 16 utterance rows and carries them through every layer of every Euler step with the
// residual stream, the latent and the MFMA A-operands resident in LDS; nothing but weights and the
// precomputed adaLN vectors is read from memory inside the loop, and there is no inter-workgroup
// synchronisation.
//
// gfx950 mapping: 1024 threads = 16 waves (4 per SIMD).  Every GEMM [16 x K] x [K x N] is cut into
// 16-column MFMA tiles; wave w takes tiles w and w+16, so for N = 512 all waves issue their whole
// weight slice (2 tiles x K/32 fragments of 1 KiB) back to back and ~0.5 MB is in flight per CU.
// Weights are stored fragment-major (pack_flow_weight): one wave-level load = 1 KiB contiguous, each
// lane receiving exactly its MFMA B fragment, straight from L2 into VGPRs (every wave reads a different
// slice, so an LDS round trip would add nothing).  f32 weights use v_mfma_f32_16x16x4_f32, bf16 weights
// v_mfma_f32_16x16x32_bf16.  The adaLN shift/scale/gate vectors depend only on the conditioning and
// the Euler step index, never on x; they come from one batched GEMM ahead of this kernel and are
// requested before the matrix loop of the layer that needs them, so their latency hides behind it.
#include <hip/hip_runtime.h>

#include <cstring>
#include <stdexcept>
#include <type_traits>
#include <vector>

#include "kernels.h"

namespace ptts {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kFlowThreads = 1024;
constexpr int kFlowWaves = kFlowThreads / 64;

template <typename WT> struct FlowTraits { static constexpr int pad = 4, kstep = 16; };
template <> struct FlowTraits<__bf16> { static constexpr int pad = 8, kstep = 32; };

// Two 16x16 output tiles (columns of tiles nt0 and nt1; nt1 < 0: none) of A[16][K] (LDS) x Wp (packed).
template <typename WT>
__device__ __forceinline__ void mm2(const WT* __restrict__ A, int lda, const WT* __restrict__ Wp, int K, int nt0, int nt1, int lane,
                                    f32x4& acc0, f32x4& acc1) {
    constexpr int KSTEP = FlowTraits<WT>::kstep;
    const int KS = K / KSTEP;
    const int r = lane & 15, g = lane >> 4;
    const uint4* w0 = (const uint4*)Wp + (long long)nt0 * KS * 64 + lane;
    const uint4* w1 = (const uint4*)Wp + (long long)(nt1 < 0 ? nt0 : nt1) * KS * 64 + lane;
    acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (std::is_same<WT, float>::value) {
        const float* ap = A + r * lda + 4 * g;
#pragma unroll 8
        for (int ks = 0; ks < KS; ++ks) {
            const float4 a4 = *(const float4*)(ap + ks * 16);
            const uint4 u0 = w0[(long long)ks * 64];
            const uint4 u1 = w1[(long long)ks * 64];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, __uint_as_float(u0.x), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, __uint_as_float(u1.x), acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, __uint_as_float(u0.y), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, __uint_as_float(u1.y), acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, __uint_as_float(u0.z), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, __uint_as_float(u1.z), acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, __uint_as_float(u0.w), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, __uint_as_float(u1.w), acc1, 0, 0, 0);
        }
    } else {
        const __bf16* ap = A + r * lda + 8 * g;
#pragma unroll 16
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 a8 = *(const bf16x8*)(ap + ks * 32);
            const uint4 u0 = w0[(long long)ks * 64];
            const uint4 u1 = w1[(long long)ks * 64];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, __builtin_bit_cast(bf16x8, u0), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, __builtin_bit_cast(bf16x8, u1), acc1, 0, 0, 0);
        }
    }
}

__device__ __forceinline__ float silu_dev(float v) { return v / (1.0f + expf(-v)); }

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// LayerNorm (optional affine) + adaLN modulate of row `wave` of h -> A-operand buffer.
// The modulation vectors are requested first so their memory latency overlaps the two reductions.
template <typename WT>
__device__ __forceinline__ void ln_modulate(const float* h, int C, WT* dst, int ldd, const float* g, const float* b,
                                            const float* mod_row, int shift_off, int scale_off, float eps, int row, int lane) {
    constexpr int MAXI = 16;  // C <= 1024
    const float* hr = h + row * C;
    const int n = C / 64;
    float sc[MAXI], sh[MAXI], v[MAXI];
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
        if (i < n) { sc[i] = mod_row[scale_off + lane + 64 * i]; sh[i] = mod_row[shift_off + lane + 64 * i]; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
        if (i < n) { v[i] = hr[lane + 64 * i]; s += v[i]; }
    const float mean = wsum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
        if (i < n) { float d = v[i] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wsum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
        if (i < n) {
            const int c = lane + 64 * i;
            float y = (v[i] - mean) * rstd;
            if (g) y = y * g[c] + b[c];
            y = y * (1.0f + sc[i]) + sh[i];
            dst[row * ldd + c] = (WT)y;
        }
}

template <typename WT>
__global__ __launch_bounds__(kFlowThreads) void flow_head_kernel(FlowHeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int PAD = FlowTraits<WT>::pad;
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
    const int NT = C / 16;
    // global rows whose modulation vectors this lane needs: its LayerNorm row and its 4 accumulator rows
    const int ln_grow = row0 + (wave < nrows ? wave : nrows - 1);
    int acc_grow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc_grow[r] = row0 + ((crow + r) < nrows ? (crow + r) : nrows - 1);

    for (int i = tid; i < 16 * L; i += kFlowThreads) {
        int r = i / L;
        xs[i] = r < nrows ? a.x0[(long long)(row0 + r) * a.x0_stride + (i % L)] : 0.f;
    }
    __syncthreads();

    for (int j = 0; j < a.steps; ++j) {
        const float* mods_j = a.mods + (long long)j * a.B * a.mod_width;
        const float* ln_mods = mods_j + (long long)ln_grow * a.mod_width;
        for (int i = tid; i < 16 * L; i += kFlowThreads) abx[(i / L) * ldx + (i % L)] = (WT)xs[i];
        __syncthreads();
        // input projection: h = x W_in^T + b_in
        for (int nt = wave; nt < NT; nt += 2 * kFlowWaves) {
            const int nt1 = nt + kFlowWaves < NT ? nt + kFlowWaves : -1;
            f32x4 acc0, acc1;
            mm2<WT>(abx, ldx, (const WT*)a.w_in, L, nt, nt1, lane, acc0, acc1);
            const int c0 = nt * 16 + ccol;
            const float b0 = a.b_in[c0];
#pragma unroll
            for (int r = 0; r < 4; ++r) h[(crow + r) * C + c0] = acc0[r] + b0;
            if (nt1 >= 0) {
                const int c1 = nt1 * 16 + ccol;
                const float b1 = a.b_in[c1];
#pragma unroll
                for (int r = 0; r < 4; ++r) h[(crow + r) * C + c1] = acc1[r] + b1;
            }
        }
        __syncthreads();
        for (int blk = 0; blk < a.blocks; ++blk) {
            const int moff = blk * 3 * C;
            ln_modulate<WT>(h, C, abuf0, lda, a.ln_g[blk], a.ln_b[blk], ln_mods, moff, moff + C, a.eps, wave, lane);
            __syncthreads();
            const WT* w1 = (const WT*)a.w_fc1[blk];
            const float* b1p = a.b_fc1[blk];
            for (int nt = wave; nt < NT; nt += 2 * kFlowWaves) {
                const int nt1 = nt + kFlowWaves < NT ? nt + kFlowWaves : -1;
                f32x4 acc0, acc1;
                mm2<WT>(abuf0, lda, w1, C, nt, nt1, lane, acc0, acc1);
                const int c0 = nt * 16 + ccol;
                const float bb0 = b1p[c0];
#pragma unroll
                for (int r = 0; r < 4; ++r) abuf1[(crow + r) * lda + c0] = (WT)silu_dev(acc0[r] + bb0);
                if (nt1 >= 0) {
                    const int c1 = nt1 * 16 + ccol;
                    const float bb1 = b1p[c1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) abuf1[(crow + r) * lda + c1] = (WT)silu_dev(acc1[r] + bb1);
                }
            }
            __syncthreads();
            const WT* w2 = (const WT*)a.w_fc2[blk];
            const float* b2p = a.b_fc2[blk];
            for (int nt = wave; nt < NT; nt += 2 * kFlowWaves) {
                const int nt1 = nt + kFlowWaves < NT ? nt + kFlowWaves : -1;
                const int c0 = nt * 16 + ccol, c1 = (nt1 < 0 ? nt : nt1) * 16 + ccol;
                // gates requested ahead of the matrix loop
                float g0[4], g1[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* gr = mods_j + (long long)acc_grow[r] * a.mod_width + moff + 2 * C;
                    g0[r] = gr[c0];
                    g1[r] = gr[c1];
                }
                f32x4 acc0, acc1;
                mm2<WT>(abuf1, lda, w2, C, nt, nt1, lane, acc0, acc1);
                const float bb0 = b2p[c0];
#pragma unroll
                for (int r = 0; r < 4; ++r) h[(crow + r) * C + c0] += g0[r] * (acc0[r] + bb0);
                if (nt1 >= 0) {
                    const float bb1 = b2p[c1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[(crow + r) * C + c1] += g1[r] * (acc1[r] + bb1);
                }
            }
            __syncthreads();
        }
        const int foff = a.blocks * 3 * C;
        ln_modulate<WT>(h, C, abuf0, lda, nullptr, nullptr, ln_mods, foff, foff + C, a.eps, wave, lane);
        __syncthreads();
        for (int nt = wave; nt < L / 16; nt += kFlowWaves) {
            f32x4 acc0, acc1;
            mm2<WT>(abuf0, lda, (const WT*)a.w_out, C, nt, -1, lane, acc0, acc1);
            const int col = nt * 16 + ccol;
            const float bias = a.b_out[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) xs[(crow + r) * L + col] += (acc0[r] + bias) * dt;
        }
        __syncthreads();
    }
    for (int i = tid; i < 16 * L; i += kFlowThreads) {
        int r = i / L;
        if (r < nrows) a.x_out[(long long)(row0 + r) * L + (i % L)] = xs[i];
    }
}

// Reorders W[N][K] (row major) into the fragment-major layout mm2 streams: tile (nt, ks) holds, for
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

void launch_flow_head(const FlowHeadArgs& a, hipStream_t s) {
    if (a.B <= 0) return;
    if (a.C % 64 || a.C > 1024 || a.L % 32 || a.steps <= 0)
        throw std::runtime_error("flow head: flow_dim must be a multiple of 64 (<= 1024) and latent_dim a multiple of 32");
    const int wbytes = a.wtype == W_BF16 ? 2 : 4;
    const int pad = a.wtype == W_BF16 ? 8 : 4;
    size_t smem = (size_t)16 * a.C * 4 + (size_t)16 * a.L * 4 + (size_t)2 * 16 * (a.C + pad) * wbytes + (size_t)16 * (a.L + pad) * wbytes;
    dim3 grid((a.B + 15) / 16), block(kFlowThreads);
    if (a.wtype == W_BF16) {
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)flow_head_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(flow_head_kernel<__bf16>, grid, block, smem, s, a);
    } else {
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)flow_head_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(flow_head_kernel<float>, grid, block, smem, s, a);
    }
}

}  // namespace ptts
