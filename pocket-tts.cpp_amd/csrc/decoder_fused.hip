// Fused residual unit of the Mimi decoder's SEANet stages (the reference runs it inside the mimi_decoder graph,
// /root/reference/src/pocket_tts.cpp:454-546):
//     e = ELU(y + W2 . ELU(W1 * ELU(y) + b1) + b2)          W1: k = 3 causal taps, C -> C/2;  W2: k = 1, C/2 -> C
// y is the up-sampling convolution's raw output, channels-last bf16 rows with the two previous rows of the stream
// in front (these stages are HBM traffic and little else, and every consumer rounds to bf16 anyway; e is bf16 too).
// The late stages are long and thin (1.8 M rows x 64 channels per 15-frame chunk of 64 utterances): as
// separate GEMM launches they are pure HBM traffic (y twice, ELU(y), the C/2-wide intermediate written and read),
// so here a workgroup takes 128 consecutive rows of one utterance through both products with the intermediate
// in LDS: y is read once (and once more from L2 for the skip), e is written once.
//
// gfx950 mapping: weights are the MFMA A operand (rows = output channels), the data rows the B operand, so each
// lane ends up with four consecutive channels of one data row: 16-byte loads of the skip and 16-byte stores.
// Wave w owns data rows 32w..32w+31 of the tile for both products, so the intermediate never crosses waves.
// v_mfma_f32_16x16x32_bf16, activations rounded to bf16 once when staged (as the decoder's GEMM path does).
#include <hip/hip_runtime.h>

#include <mutex>
#include <set>
#include <stdexcept>

#include "kernels.h"

namespace ptts {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ELU whose results are stored as bf16: exp(v) - 1 by the hardware exponential is accurate to ~1e-7 absolute, far
// below the bf16 rounding of the value it feeds
__device__ __forceinline__ float elu_dev(float v) { return v > 0.f ? v : __expf(v) - 1.0f; }

template <int CO>
__global__ __launch_bounds__(256) void resblock_kernel(ResblockArgs a) {
    constexpr int CH = CO / 2, TM = 128, TAPS = 3, HIST = TAPS - 1;
    constexpr int LDA = CO + 8, LDH = CH + 8, LDW1 = TAPS * CO + 8, LDW2 = CH + 8;   // bf16 elements; +8 staggers the banks
    constexpr int KS1 = TAPS * CO / 32, KS2 = CH / 32, MT1 = CH / 16, MT2 = CO / 16;
    static_assert(CH % 32 == 0, "resblock: channel count must be a multiple of 64");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* As = (__bf16*)smem_raw;                  // [TM + HIST][LDA]   ELU(y), bf16
    __bf16* Hs = As + (TM + HIST) * LDA;             // [TM][LDH]          ELU(c1), bf16
    __bf16* W1s = Hs + TM * LDH;                     // [CH][LDW1]
    __bf16* W2s = W1s + CH * LDW1;                   // [CO][LDW2]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.y;
    const __bf16* Y = (const __bf16*)a.Y + (long long)b * a.y_bs;    // row j = data row j - HIST
    // ---- the weights, once per workgroup (they are as large as a tile's data)
    {
        constexpr int P1 = TAPS * CO / 8, P2 = CH / 8;
        for (int i = tid; i < CH * P1; i += 256) {
            const int r = i / P1, p = i % P1;
            *(uint4*)(W1s + r * LDW1 + 8 * p) = *(const uint4*)((const __bf16*)a.W1 + (long long)r * TAPS * CO + 8 * p);
        }
        for (int i = tid; i < CO * P2; i += 256) {
            const int r = i / P2, p = i % P2;
            *(uint4*)(W2s + r * LDW2 + 8 * p) = *(const uint4*)((const __bf16*)a.W2 + (long long)r * CH + 8 * p);
        }
    }
    const int li = lane & 15, g4 = lane >> 4;
    const int rw = w * 32;                            // this wave's first row inside the tile
    const int ntiles = (a.T + TM - 1) / TM;
    // A tile's rows (t0 - 2 .. t0 + TM - 1 of the stream, 16 bytes per lane and piece) are requested one tile ahead, so
    // that their way from HBM overlaps the previous tile's products and stores instead of heading a dependent chain.
    constexpr int PPR = CO / 8, NPIECE = ((TM + HIST) * PPR + 255) / 256;
    bf16x8 ynext[NPIECE];
    auto request_tile = [&](int tile) {
        const int t0n = tile * TM;
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) {
            const int i = tid + 256 * j, r = i / PPR, p = i % PPR;
            bf16x8 v = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            if (i < (TM + HIST) * PPR && t0n + r < a.T + HIST) v = *(const bf16x8*)(Y + (long long)(t0n + r) * CO + 8 * p);
            ynext[j] = v;
        }
    };
    if ((int)blockIdx.x < ntiles) request_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int t0 = tile * TM;
    __syncthreads();                                  // the previous tile's readers of As are done
    // ---- stage the tile: ELU, bf16
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
        const int i = tid + 256 * j, r = i / PPR, p = i % PPR;
        if (i >= (TM + HIST) * PPR) continue;
        bf16x8 e;
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = (__bf16)elu_dev((float)ynext[j][k]);
        *(bf16x8*)(As + r * LDA + 8 * p) = e;
    }
    if (tile + (int)gridDim.x < ntiles) request_tile(tile + gridDim.x);
    __syncthreads();
    // ---- c1: H^T[CH][rows] = W1[CH][3 CO] . A^T
    f32x4 acc1[MT1][2];
#pragma unroll
    for (int m = 0; m < MT1; ++m) { acc1[m][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[m][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        const int tap = ks / (CO / 32), c0 = (ks % (CO / 32)) * 32;
        const bf16x8 d0 = *(const bf16x8*)(As + (rw + li + tap) * LDA + c0 + 8 * g4);
        const bf16x8 d1 = *(const bf16x8*)(As + (rw + 16 + li + tap) * LDA + c0 + 8 * g4);
#pragma unroll
        for (int m = 0; m < MT1; ++m) {
            const bf16x8 wf = *(const bf16x8*)(W1s + (m * 16 + li) * LDW1 + ks * 32 + 8 * g4);
            acc1[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, d0, acc1[m][0], 0, 0, 0);
            acc1[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, d1, acc1[m][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int m = 0; m < MT1; ++m) {
        const int ch = m * 16 + 4 * g4;
        const float4 bb = *(const float4*)(a.b1 + ch);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            bf16x4 h;
            h[0] = (__bf16)elu_dev(acc1[m][n][0] + bb.x); h[1] = (__bf16)elu_dev(acc1[m][n][1] + bb.y);
            h[2] = (__bf16)elu_dev(acc1[m][n][2] + bb.z); h[3] = (__bf16)elu_dev(acc1[m][n][3] + bb.w);
            *(bf16x4*)(Hs + (rw + n * 16 + li) * LDH + ch) = h;
        }
    }
    // the wave reads back only rows it wrote itself
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // ---- c2: E^T[CO][rows] = W2[CO][CH] . H^T, + bias + y, ELU
    f32x4 acc2[MT2][2];
#pragma unroll
    for (int m = 0; m < MT2; ++m) { acc2[m][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[m][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
        const bf16x8 d0 = *(const bf16x8*)(Hs + (rw + li) * LDH + ks * 32 + 8 * g4);
        const bf16x8 d1 = *(const bf16x8*)(Hs + (rw + 16 + li) * LDH + ks * 32 + 8 * g4);
#pragma unroll
        for (int m = 0; m < MT2; ++m) {
            const bf16x8 wf = *(const bf16x8*)(W2s + (m * 16 + li) * LDW2 + ks * 32 + 8 * g4);
            acc2[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, d0, acc2[m][0], 0, 0, 0);
            acc2[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, d1, acc2[m][1], 0, 0, 0);
        }
    }
    __bf16* E = (__bf16*)a.E + (long long)b * a.e_bs + (long long)a.e_off * CO;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int t = t0 + rw + n * 16 + li;
        if (t >= a.T) continue;
#pragma unroll
        for (int m = 0; m < MT2; ++m) {
            const int ch = m * 16 + 4 * g4;
            const float4 bb = *(const float4*)(a.b2 + ch);
            const bf16x4 y4 = *(const bf16x4*)(Y + (long long)(t + HIST) * CO + ch);   // the skip, from L2
            bf16x4 o;
            o[0] = (__bf16)elu_dev(acc2[m][n][0] + bb.x + (float)y4[0]); o[1] = (__bf16)elu_dev(acc2[m][n][1] + bb.y + (float)y4[1]);
            o[2] = (__bf16)elu_dev(acc2[m][n][2] + bb.z + (float)y4[2]); o[3] = (__bf16)elu_dev(acc2[m][n][3] + bb.w + (float)y4[3]);
            *(bf16x4*)(E + (long long)t * CO + ch) = o;
        }
    }
    }   // tiles
}

template <int CO> static size_t resblock_smem() {
    constexpr int CH = CO / 2;
    return sizeof(__bf16) * ((size_t)(128 + 2) * (CO + 8) + (size_t)128 * (CH + 8) + (size_t)CH * (3 * CO + 8) + (size_t)CO * (CH + 8));
}

template <int CO> static void launch_one(const ResblockArgs& a, hipStream_t s) {
    const size_t smem = resblock_smem<CO>();
    if (smem > 64 * 1024) {   // per device, once
        static std::mutex mu;
        static std::set<int> done;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> g(mu);
        if (done.insert(dev).second &&
            hipFuncSetAttribute((const void*)resblock_kernel<CO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            throw std::runtime_error("resblock: cannot raise the dynamic LDS limit");
    }
    const int ntiles = (a.T + 127) / 128;
    const int per_wg = ntiles >= 64 ? 8 : 1;          // long streams: a workgroup walks 8 tiles with one copy of the weights
    hipLaunchKernelGGL(resblock_kernel<CO>, dim3((ntiles + per_wg - 1) / per_wg, a.B), dim3(256), smem, s, a);
}

bool resblock_supported(int co, int taps, int wtype) { return wtype == W_BF16 && taps == 3 && (co == 64 || co == 128); }

// Raises the LDS limits outside any stream capture.
void resblock_prepare() {
    ResblockArgs a;
    (void)a;
    const size_t smem = resblock_smem<128>();
    (void)hipFuncSetAttribute((const void*)resblock_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
}

void launch_resblock(const ResblockArgs& a, hipStream_t s) {
    if (a.T <= 0 || a.B <= 0) return;
    if (!resblock_supported(a.co, 3, W_BF16)) throw std::runtime_error("resblock: fused unit supports 64 or 128 channels");
    if (((uintptr_t)a.Y & 15) || ((uintptr_t)a.E & 15) || ((uintptr_t)a.W1 & 15) || ((uintptr_t)a.W2 & 15) || (a.y_bs & 7) || (a.e_bs & 7))
        throw std::runtime_error("resblock: operands must be 16-byte aligned");
    if (a.co == 64) launch_one<64>(a, s);
    else launch_one<128>(a, s);
}

}  // namespace ptts
