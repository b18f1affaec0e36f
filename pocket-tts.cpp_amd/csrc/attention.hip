// Multi-row cached attention (prefill of the LM, Mimi decoder and encoder transformers): many query rows of
// one (utterance, head) against that utterance's K/V cache, causal with an optional sliding window.
//
// gfx950 mapping: a 4-wave workgroup owns 64 consecutive query rows of one (slot, head); each wave owns 16 of
// them.  K/V rows are staged through LDS in blocks of 64 keys shared by the four waves, then consumed 16 keys
// at a time with v_mfma_f32_16x16x4_f32 (plain f32 products and sums, so the fp32 engine stays at f32
// accuracy).  The products are arranged transposed so that no operand ever has to change layout:
//     S^T[key][query] = K . Q^T      A = K rows from LDS, B = Q rows held in registers
//     O^T[dim][query] = V^T . P^T    A = V from LDS,      B = P^T = exp(S^T - max), already in B-operand layout
// Every lane therefore keeps exactly one query (lane % 16) for the whole kernel: the running max, the running
// sum and the rescale factor of the online softmax are per-lane scalars.
#include <hip/hip_runtime.h>

#include <stdexcept>

#include "kernels.h"

namespace ptts {

typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kKB = 64;        // keys per staged block
constexpr int kKStride = 68;   // floats per K row in LDS (16-byte aligned rows, 4-bank skew)
constexpr int kVStride = 66;   // floats per V row in LDS (read as scalars)

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;

template <bool KV16>
__global__ __launch_bounds__(256) void attention_tile_kernel(AttentionArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[kKB * kKStride];
    __shared__ __attribute__((aligned(16))) float Vs[kKB * kVStride];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g4 = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int t_wg = blockIdx.x * 64;                      // first query row (within the slot's rows) of this workgroup
    const int pos0 = a.pos[b];
    const int tq = t_wg + wave * 16 + li;                  // this lane's query row
    const bool q_live = tq < a.rpb;
    const int pq = pos0 + (q_live ? tq : a.rpb - 1);       // its absolute position
    const int lo_q = (a.window > 0 && pq - a.window + 1 > 0) ? pq - a.window + 1 : 0;

    // Q^T operand: dims dg*16 + 4*g4 .. +3 of this lane's query, pre-scaled by 1/sqrt(64)
    float4 qreg[4];
    {
        const long long m = (long long)b * a.rpb + (q_live ? tq : a.rpb - 1);
        const float* qp = a.q + m * a.d + h * 64 + 4 * g4;
#pragma unroll
        for (int dg = 0; dg < 4; ++dg) {
            float4 v = *(const float4*)(qp + dg * 16);
            qreg[dg] = make_float4(v.x * 0.125f, v.y * 0.125f, v.z * 0.125f, v.w * 0.125f);
        }
    }

    // key range needed by any query of the workgroup
    const int t_last = (t_wg + 63 < a.rpb ? t_wg + 63 : a.rpb - 1);
    const int p_hi = pos0 + t_last;
    int k_lo = 0;
    if (a.window > 0 && pos0 + t_wg - a.window + 1 > 0) k_lo = pos0 + t_wg - a.window + 1;
    const long long cbase = (long long)(a.slot0 + b) * a.cap * a.d + h * 64;

    // staging assignment: thread -> key rows (tid / 16) + 16 r, dims 4 * (tid % 16) .. +3
    const int s_key = tid >> 4, s_dim = 4 * (tid & 15);
    float4 kreg[4], vreg[4];
    auto load_block = [&](int kb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int key = kb + s_key + 16 * r;
            if (key > p_hi) key = p_hi;
            const long long off = cbase + (long long)(key % a.cap) * a.d + s_dim;
            if constexpr (KV16) {
                const bf16x4_t k4b = *(const bf16x4_t*)((const __bf16*)a.kcache + off), v4b = *(const bf16x4_t*)((const __bf16*)a.vcache + off);
                kreg[r] = make_float4((float)k4b[0], (float)k4b[1], (float)k4b[2], (float)k4b[3]);
                vreg[r] = make_float4((float)v4b[0], (float)v4b[1], (float)v4b[2], (float)v4b[3]);
            } else {
                kreg[r] = *(const float4*)((const float*)a.kcache + off);
                vreg[r] = *(const float4*)((const float*)a.vcache + off);
            }
        }
    };
    auto store_block = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = s_key + 16 * r;
            *(float4*)(Ks + row * kKStride + s_dim) = kreg[r];
            *(float2*)(Vs + row * kVStride + s_dim) = make_float2(vreg[r].x, vreg[r].y);
            *(float2*)(Vs + row * kVStride + s_dim + 2) = make_float2(vreg[r].z, vreg[r].w);
        }
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) o[it] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_block(k_lo);
    for (int kb = k_lo; kb <= p_hi; kb += kKB) {
        __syncthreads();   // previous block fully consumed
        store_block();
        __syncthreads();
        if (kb + kKB <= p_hi) load_block(kb + kKB);
#pragma unroll
        for (int sub = 0; sub < kKB / 16; ++sub) {
            const int key0 = kb + sub * 16;
            if (key0 > p_hi) break;                                   // uniform over the workgroup
            // S^T[key][query]
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* kp = Ks + (sub * 16 + li) * kKStride + 4 * g4;
#pragma unroll
            for (int dg = 0; dg < 4; ++dg) {
                const float4 k4 = *(const float4*)(kp + dg * 16);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(k4.x, qreg[dg].x, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(k4.y, qreg[dg].y, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(k4.z, qreg[dg].z, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(k4.w, qreg[dg].w, s, 0, 0, 0);
            }
            // this lane: query li, keys key0 + 4*g4 + v
            float bm = -INFINITY;
            bool ok[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int key = key0 + 4 * g4 + v;
                ok[v] = q_live && key <= pq && key >= lo_q;
                if (ok[v]) bm = fmaxf(bm, s[v]);
            }
            bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
            const float m_new = fmaxf(m_run, bm);
            float p[4], ps = 0.f, scale = 1.f;
            if (m_new == -INFINITY) {
#pragma unroll
                for (int v = 0; v < 4; ++v) p[v] = 0.f;
            } else {
                scale = expf(m_run - m_new);                          // m_run = -inf -> 0
#pragma unroll
                for (int v = 0; v < 4; ++v) { p[v] = ok[v] ? expf(s[v] - m_new) : 0.f; ps += p[v]; }
            }
            ps += __shfl_xor(ps, 16, 64);
            ps += __shfl_xor(ps, 32, 64);
            l_run = l_run * scale + ps;
            m_run = m_new;
            // O^T[dim][query] = scale * O^T + V^T . P^T
            const float* vp = Vs + (sub * 16 + 4 * g4) * kVStride + li;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                o[it][0] *= scale; o[it][1] *= scale; o[it][2] *= scale; o[it][3] *= scale;
#pragma unroll
                for (int v = 0; v < 4; ++v) o[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[v * kVStride + it * 16], p[v], o[it], 0, 0, 0);
            }
        }
    }
    if (!q_live) return;
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    float* op = a.out + ((long long)b * a.rpb + tq) * a.d + h * 64 + 4 * g4;
#pragma unroll
    for (int it = 0; it < 4; ++it)
        *(float4*)(op + it * 16) = make_float4(o[it][0] * inv, o[it][1] * inv, o[it][2] * inv, o[it][3] * inv);
}

void launch_attention_tiled(const AttentionArgs& a, hipStream_t s) {
    if (a.M <= 0) return;
    if (a.d != a.heads * 64) throw std::runtime_error("attention: head dimension must be 64");
    if (a.rpb <= 0 || a.M % a.rpb) throw std::runtime_error("attention: rows must be a whole number of slots");
    dim3 grid((a.rpb + 63) / 64, a.heads, a.M / a.rpb), block(256);
    if (a.kv16) hipLaunchKernelGGL(attention_tile_kernel<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(attention_tile_kernel<false>, grid, block, 0, s, a);
}

}  // namespace ptts
