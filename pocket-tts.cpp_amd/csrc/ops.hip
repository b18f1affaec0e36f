// Row-wise and state-handling kernels around the GEMMs: LayerNorm, RoPE + KV-cache append, cached
// attention, embedding gather, latent preparation, the two single-channel convolutions of the codec,
// the depthwise up-sampler and the history carry of the streaming convolutions.
// Wave = 64 lanes throughout; the attention head dimension is fixed at 64 so one lane owns one channel.
#include <hip/hip_runtime.h>

#include <stdexcept>

#include "kernels.h"

namespace ptts {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ------------------------------------------------------------------ LayerNorm: one wave per row
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_ln;
__global__ __launch_bounds__(256) void layernorm_kernel(LayerNormArgs a) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= a.M) return;
    const int b = m / a.rpb, t = m % a.rpb;
    const float* x = a.X + (long long)b * a.x_bs + (long long)t * a.ldx;
    float* y = a.Y + (long long)b * a.y_bs + (long long)t * a.ldy;
    __bf16* y16 = (__bf16*)a.Y + (long long)b * a.y_bs + (long long)t * a.ldy;
    const int D = a.D;
    float s = 0.f;
    for (int i = lane * 4; i < D; i += 256) {
        float4 v = *(const float4*)(x + i);
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int i = lane * 4; i < D; i += 256) {
        float4 v = *(const float4*)(x + i);
        float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + a.eps);
    for (int i = lane * 4; i < D; i += 256) {
        float4 v = *(const float4*)(x + i);
        float4 o;
        o.x = (v.x - mean) * rstd; o.y = (v.y - mean) * rstd; o.z = (v.z - mean) * rstd; o.w = (v.w - mean) * rstd;
        if (a.g) {
            float4 g = *(const float4*)(a.g + i), bb = *(const float4*)(a.b + i);
            o.x = o.x * g.x + bb.x; o.y = o.y * g.y + bb.y; o.z = o.z * g.z + bb.z; o.w = o.w * g.w + bb.w;
        }
        if (a.y_bf16) {
            bf16x4_ln o16;
            o16[0] = (__bf16)o.x; o16[1] = (__bf16)o.y; o16[2] = (__bf16)o.z; o16[3] = (__bf16)o.w;
            *(bf16x4_ln*)(y16 + i) = o16;
        } else {
            *(float4*)(y + i) = o;
        }
    }
}

void launch_layernorm(const LayerNormArgs& a, hipStream_t s) {
    if (a.M <= 0) return;
    if ((a.D & 3) || (a.ldx & 3) || (a.ldy & 3) || (a.x_bs & 3) || (a.y_bs & 3))
        throw std::runtime_error("layernorm: widths and strides must be multiples of 4");
    hipLaunchKernelGGL(layernorm_kernel, dim3((a.M + 3) / 4), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------ RoPE + KV append
// One thread per (row, pair): pair index over d/2; q/k pairs rotate, v copies.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;

// 4 consecutive cache values as float4, from an f32 or a bf16 cache
template <bool KV16>
__device__ __forceinline__ float4 kv_load4(const void* base, long long idx) {
    if constexpr (KV16) {
        const bf16x4_t v = *(const bf16x4_t*)((const __bf16*)base + idx);
        return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    } else {
        return *(const float4*)((const float*)base + idx);
    }
}

template <bool KV16>
__global__ __launch_bounds__(256) void rope_append_kernel(RopeAppendArgs a) {
    const int half = a.d >> 1;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.M * half) return;
    const int m = (int)(idx / half), pi = (int)(idx % half);
    const int dh2 = (a.d / a.heads) >> 1;
    const int i = pi % dh2;  // pair index inside the head
    const int b = m / a.rpb, t = m % a.rpb;
    const int p = a.pos[b] + t;
    const int pc = p < a.tab_len ? p : a.tab_len - 1;
    const float c = a.cos_tab[(long long)pc * dh2 + i], s = a.sin_tab[(long long)pc * dh2 + i];
    const float* row = a.qkv + (long long)m * 3 * a.d;
    const float2 q = *(const float2*)(row + 2 * pi);
    const float2 k = *(const float2*)(row + a.d + 2 * pi);
    const float2 v = *(const float2*)(row + 2 * a.d + 2 * pi);
    float2 qo, ko;
    qo.x = q.x * c - q.y * s; qo.y = q.x * s + q.y * c;
    ko.x = k.x * c - k.y * s; ko.y = k.x * s + k.y * c;
    *(float2*)(a.q_out + (long long)m * a.d + 2 * pi) = qo;
    const long long crow = ((long long)(a.slot0 + b) * a.cap + (p % a.cap)) * a.d + 2 * pi;
    if constexpr (KV16) {
        bf16x2_t kb, vb;
        kb[0] = (__bf16)ko.x; kb[1] = (__bf16)ko.y; vb[0] = (__bf16)v.x; vb[1] = (__bf16)v.y;
        *(bf16x2_t*)((__bf16*)a.kcache + crow) = kb;
        *(bf16x2_t*)((__bf16*)a.vcache + crow) = vb;
    } else {
        *(float2*)((float*)a.kcache + crow) = ko;
        *(float2*)((float*)a.vcache + crow) = v;
    }
}

void launch_rope_append(const RopeAppendArgs& a, hipStream_t s) {
    if (a.M <= 0) return;
    long long n = (long long)a.M * (a.d / 2);
    if (a.kv16) hipLaunchKernelGGL(rope_append_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(rope_append_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------ cached attention, one wave per (row, head)
// Pass 1: lane j scores keys j, j+64, ... against q (held by every lane); scores go to LDS.
// Pass 2: lane e accumulates channel e of the output over all keys (coalesced 256-byte V rows).
template <bool KV16>
__global__ __launch_bounds__(256) void attention_kernel(AttentionArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long item = (long long)blockIdx.x * 4 + w;
    const bool live = item < (long long)a.M * a.heads;
    const int m = live ? (int)(item / a.heads) : 0, h = live ? (int)(item % a.heads) : 0;
    const int b = m / a.rpb, t = m % a.rpb;
    const int p = a.pos[b] + t;
    int j0 = 0;
    if (a.window > 0 && p - a.window + 1 > 0) j0 = p - a.window + 1;
    const int nk = live ? (p - j0 + 1) : 0;
    float* sc = smem + (long long)w * a.cap;
    const long long cbase = (long long)(a.slot0 + b) * a.cap * a.d + h * 64;
    const float* qp = a.q + (long long)m * a.d + h * 64;
    float4 q[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) q[i] = *(const float4*)(qp + 4 * i);
    float mx = -INFINITY;
    for (int j = lane; j < nk; j += 64) {
        const long long kr = cbase + (long long)((j0 + j) % a.cap) * a.d;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float4 kv = kv_load4<KV16>(a.kcache, kr + 4 * i);
            s0 += q[i].x * kv.x; s1 += q[i].y * kv.y; s2 += q[i].z * kv.z; s3 += q[i].w * kv.w;
        }
        float sv = ((s0 + s1) + (s2 + s3)) * 0.125f;
        sc[j] = sv;
        mx = fmaxf(mx, sv);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < nk; j += 64) {
        float e = expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS scores written by other lanes of this wave
    __builtin_amdgcn_wave_barrier();
    // P.V with the wave split into 4 key groups x 16 channel quads (four keys in flight per lane)
    const int kg = lane >> 4, cq = lane & 15;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    int j = kg;
    for (; j + 12 < nk; j += 16) {
        const float4 v0 = kv_load4<KV16>(a.vcache, cbase + (long long)((j0 + j) % a.cap) * a.d + 4 * cq);
        const float4 v1 = kv_load4<KV16>(a.vcache, cbase + (long long)((j0 + j + 4) % a.cap) * a.d + 4 * cq);
        const float4 v2 = kv_load4<KV16>(a.vcache, cbase + (long long)((j0 + j + 8) % a.cap) * a.d + 4 * cq);
        const float4 v3 = kv_load4<KV16>(a.vcache, cbase + (long long)((j0 + j + 12) % a.cap) * a.d + 4 * cq);
        const float p0 = sc[j], p1 = sc[j + 4], p2 = sc[j + 8], p3 = sc[j + 12];
        o.x += (p0 * v0.x + p1 * v1.x) + (p2 * v2.x + p3 * v3.x);
        o.y += (p0 * v0.y + p1 * v1.y) + (p2 * v2.y + p3 * v3.y);
        o.z += (p0 * v0.z + p1 * v1.z) + (p2 * v2.z + p3 * v3.z);
        o.w += (p0 * v0.w + p1 * v1.w) + (p2 * v2.w + p3 * v3.w);
    }
    for (; j < nk; j += 4) {
        const float4 v0 = kv_load4<KV16>(a.vcache, cbase + (long long)((j0 + j) % a.cap) * a.d + 4 * cq);
        const float p0 = sc[j];
        o.x += p0 * v0.x; o.y += p0 * v0.y; o.z += p0 * v0.z; o.w += p0 * v0.w;
    }
    o.x += __shfl_xor(o.x, 16, 64); o.y += __shfl_xor(o.y, 16, 64); o.z += __shfl_xor(o.z, 16, 64); o.w += __shfl_xor(o.w, 16, 64);
    o.x += __shfl_xor(o.x, 32, 64); o.y += __shfl_xor(o.y, 32, 64); o.z += __shfl_xor(o.z, 32, 64); o.w += __shfl_xor(o.w, 32, 64);
    if (live && kg == 0) {
        const float inv = 1.0f / sum;
        *(float4*)(a.out + (long long)m * a.d + h * 64 + 4 * cq) = make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);
    }
}

void launch_attention(const AttentionArgs& a, hipStream_t s) {
    if (a.M <= 0) return;
    if (a.d != a.heads * 64) throw std::runtime_error("attention: head dimension must be 64");
    if (a.rpb >= 16 && a.M % a.rpb == 0) { launch_attention_tiled(a, s); return; }   // many rows per slot: MFMA tiles
    long long items = (long long)a.M * a.heads;
    size_t smem = (size_t)4 * a.cap * sizeof(float);
    if (smem > 160 * 1024) throw std::runtime_error("attention: context capacity too large for LDS score buffer");
    if (a.kv16) hipLaunchKernelGGL(attention_kernel<true>, dim3((unsigned)((items + 3) / 4)), dim3(256), smem, s, a);
    else hipLaunchKernelGGL(attention_kernel<false>, dim3((unsigned)((items + 3) / 4)), dim3(256), smem, s, a);
}

// ------------------------------------------------------------------ small helpers
__global__ void advance_pos_kernel(int* pos, int n, int delta, unsigned long long mask, int slot0) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (mask == ~0ull || ((mask >> (slot0 + i)) & 1ull))) pos[i] += delta;
}
void launch_advance_pos(int* pos, int n, int delta, hipStream_t s, unsigned long long mask, int slot0) {
    if (n <= 0) return;
    hipLaunchKernelGGL(advance_pos_kernel, dim3((n + 63) / 64), dim3(64), 0, s, pos, n, delta, mask, slot0);
}

__global__ void poke_int_kernel(int* p, int v) { *p = v; }
void launch_poke_int(int* p, int v, hipStream_t s) { hipLaunchKernelGGL(poke_int_kernel, dim3(1), dim3(1), 0, s, p, v); }

__global__ void reset_pos_mask_kernel(int* pos, unsigned long long mask) {
    const int i = threadIdx.x;
    if ((mask >> i) & 1ull) pos[i] = 0;
}
void launch_reset_pos_mask(int* pos, unsigned long long mask, hipStream_t s) {
    if (mask) hipLaunchKernelGGL(reset_pos_mask_kernel, dim3(1), dim3(64), 0, s, pos, mask);
}

// grid (8, 2 * layers, k): block (x, y, i) copies an eighth of one layer's K (y even) or V (y odd) rows of newcomer i;
// the blocks with y == 0 also fill the newcomer's start-point ring rows, block (0, 0, i) its position and BOS row.
__global__ __launch_bounds__(256) void admit_kernel(AdmitArgs a) {
    const int i = blockIdx.z, layer = blockIdx.y >> 1, tid = threadIdx.x;
    const int slot = a.slot[i], si = a.src[i];
    const size_t slot_bytes = static_cast<size_t>(a.cap) * a.row_bytes;
    const size_t n16 = static_cast<size_t>(a.len[i]) * a.row_bytes / 16;
    const char* src = ((blockIdx.y & 1) ? a.v_src : a.k_src) + (static_cast<size_t>(layer) * a.src_slots + si) * slot_bytes;
    char* dst = ((blockIdx.y & 1) ? a.v_dst : a.k_dst) + (static_cast<size_t>(layer) * a.dst_slots + slot) * slot_bytes;
    for (size_t j = blockIdx.x * 256 + tid; j < n16; j += gridDim.x * 256)
        reinterpret_cast<float4*>(dst)[j] = reinterpret_cast<const float4*>(src)[j];
    if (blockIdx.y == 0) {
        float* ring = a.noise_ring + static_cast<size_t>(slot) * a.ring * a.L;
        const float* nz = a.noise_stage + static_cast<size_t>(si) * a.frames * a.L;
        const bool has = a.has_noise[i] != 0;
        for (int j = blockIdx.x * 256 + tid; j < a.frames * a.L; j += gridDim.x * 256) {
            const int f = j / a.L, c = j - f * a.L;
            ring[static_cast<size_t>((a.g0 + f) % a.ring) * a.L + c] = has ? nz[j] : 0.f;
        }
        if (blockIdx.x == 0) {
            if (tid == 0) a.lm_pos[slot] = a.len[i];
            for (int c = tid; c < a.L; c += 256) a.xcur[static_cast<size_t>(slot) * a.L + c] = a.bos[c];
        }
    }
}
void launch_admit(const AdmitArgs& a, hipStream_t s) {
    if (a.k <= 0) return;
    hipLaunchKernelGGL(admit_kernel, dim3(8, 2 * a.layers, a.k), dim3(256), 0, s, a);
}

__global__ __launch_bounds__(256) void dec_reset_kernel(DecResetArgs a) {
    const int bi = blockIdx.x, slot = a.slot[blockIdx.y];
    unsigned* p = reinterpret_cast<unsigned*>(a.buf[bi] + static_cast<size_t>(slot) * a.per_slot[bi]);
    for (int j = threadIdx.x; j < a.hist[bi] / 4; j += 256) p[j] = 0u;
    if (bi == 0 && threadIdx.x == 0) a.dec_pos[slot] = 0;
}
void launch_dec_reset(const DecResetArgs& a, hipStream_t s) {
    if (a.k <= 0 || a.nbuf <= 0) return;
    hipLaunchKernelGGL(dec_reset_kernel, dim3(a.nbuf, a.k), dim3(256), 0, s, a);
}

__global__ void gather_latents_kernel(const float* lat, int ring, int slot0, int nb, int start, int n, int L, const float* stdv, const float* mean,
                                      float* z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb * n * L) return;
    const int c = i % L, f = (i / L) % n, b = i / (L * n);
    const float v = lat[((long long)(slot0 + b) * ring + (start + f) % ring) * L + c];
    z[i] = v * stdv[c] + mean[c];
}
void launch_gather_latents(const float* lat, int ring, int slot0, int nb, int start, int n, int L, const float* stdv, const float* mean, float* z,
                           hipStream_t s) {
    const int total = nb * n * L;
    if (total <= 0) return;
    hipLaunchKernelGGL(gather_latents_kernel, dim3((total + 255) / 256), dim3(256), 0, s, lat, ring, slot0, nb, start, n, L, stdv, mean, z);
}

__global__ void advance_pos_vec_kernel(int* pos, const int* delta, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pos[i] += delta[i];
}
void launch_advance_pos_vec(int* pos, const int* delta, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(advance_pos_vec_kernel, dim3((n + 63) / 64), dim3(64), 0, s, pos, delta, n);
}

__global__ void assemble_prefill_kernel(const float* voice, long long voice_bs, const long long* ids, int ids_stride, const int* nv,
                                        const int* nt, const float* table, int vocab_rows, int D, int rpb, float* out) {
    const int t = blockIdx.x, b = blockIdx.y;
    const int v = nv[b], n = nt[b];
    float4* dst = (float4*)(out + ((long long)b * rpb + t) * D);
    const float4* src = nullptr;
    if (t < v) {
        src = (const float4*)(voice + (long long)b * voice_bs + (long long)t * D);
    } else if (t < v + n) {
        long long id = ids[(long long)b * ids_stride + (t - v)];
        if (id < 0 || id >= vocab_rows) id = vocab_rows - 1;
        src = (const float4*)(table + id * D);
    }
    for (int i = threadIdx.x; i < D / 4; i += blockDim.x) dst[i] = src ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
}
void launch_assemble_prefill(const float* voice, long long voice_bs, const long long* ids, int ids_stride, const int* nv, const int* nt,
                             const float* table, int vocab_rows, int D, int rpb, int B, float* out, hipStream_t s) {
    if (rpb <= 0 || B <= 0) return;
    hipLaunchKernelGGL(assemble_prefill_kernel, dim3(rpb, B), dim3(256), 0, s, voice, voice_bs, ids, ids_stride, nv, nt, table, vocab_rows,
                       D, rpb, out);
}

__global__ void embed_rows_kernel(const long long* ids, int T, const float* table, int vocab_rows, int D, float* out) {
    const int t = blockIdx.x;
    long long id = ids[t];
    if (id < 0 || id >= vocab_rows) id = vocab_rows - 1;
    const float4* src = (const float4*)(table + id * D);
    float4* dst = (float4*)(out + (long long)t * D);
    for (int i = threadIdx.x; i < D / 4; i += blockDim.x) dst[i] = src[i];
}
void launch_embed_rows(const long long* ids, int T, const float* table, int vocab_rows, int D, float* out, hipStream_t s) {
    if (T <= 0) return;
    hipLaunchKernelGGL(embed_rows_kernel, dim3(T), dim3(256), 0, s, ids, T, table, vocab_rows, D, out);
}

__global__ void prep_latent_kernel(const float* x, float* y, int n, int L, int mode, const float* p0, const float* p1) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = i % L;
    float v = x[i];
    y[i] = mode == 0 ? (isnan(v) ? p0[c] : v) : (v * p0[c] + p1[c]);
}
void launch_prep_latent(const float* x, float* y, int rows, int L, int mode, const float* p0, const float* p1, hipStream_t s) {
    int n = rows * L;
    if (n <= 0) return;
    hipLaunchKernelGGL(prep_latent_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, y, n, L, mode, p0, p1);
}

__global__ __launch_bounds__(256) void rowdot_kernel(const float* X, int ldx, int M, int D, const float* w, const float* b, float* out) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* x = X + (long long)m * ldx;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += x[i] * w[i];
    s = wave_sum(s);
    if (lane == 0) out[m] = s + b[0];
}
void launch_rowdot(const float* X, int ldx, int M, int D, const float* w, const float* b, float* out, hipStream_t s) {
    if (M <= 0) return;
    hipLaunchKernelGGL(rowdot_kernel, dim3((M + 3) / 4), dim3(256), 0, s, X, ldx, M, D, w, b, out);
}

// y[(b*n + p)*U + r][c] = q[b][1+p][c] * w[c][r] + q[b][p][c] * w[c][r+U]
__global__ void upsample_dw_kernel(const float* q, long long q_bs, int n, int M, int U, const float* w, float* y, long long total) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int c = (int)(i % M);
    long long row = i / M;
    int r = (int)(row % U);
    long long bp = row / U;
    int p = (int)(bp % n), b = (int)(bp / n);
    const float* qb = q + (long long)b * q_bs;
    y[i] = qb[(long long)(1 + p) * M + c] * w[(long long)c * 2 * U + r] + qb[(long long)p * M + c] * w[(long long)c * 2 * U + r + U];
}
void launch_upsample_dw(const float* q, long long q_bs, int n, int M, int U, const float* w, float* y, int B, hipStream_t s) {
    long long total = (long long)B * n * U * M;
    if (total <= 0) return;
    hipLaunchKernelGGL(upsample_dw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, q, q_bs, n, M, U, w, y, total);
}

// One wave per 64 consecutive outputs; lane t' computes a dot over its contiguous KC-float window.
__global__ __launch_bounds__(256) void conv_to_mono_kernel(const float* X, long long x_bs, int T, int KC, int C, const float* w,
                                                           const float* bias, float* out, long long out_bs) {
    extern __shared__ float wsh[];
    for (int i = threadIdx.x; i < KC; i += blockDim.x) wsh[i] = w[i];
    __syncthreads();
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const float* x = X + (long long)b * x_bs + (long long)t * C;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int i = 0; i < KC; i += 4) {
        float4 v = *(const float4*)(x + i);
        s0 += v.x * wsh[i]; s1 += v.y * wsh[i + 1]; s2 += v.z * wsh[i + 2]; s3 += v.w * wsh[i + 3];
    }
    out[(long long)b * out_bs + t] = ((s0 + s1) + (s2 + s3)) + bias[0];
}

// bf16 rows: a workgroup stages the 256 + k - 1 rows its outputs need into LDS with 16-byte loads (each row is read
// from memory once instead of k times by k different lanes), then every thread walks its own window.  Row stride
// C + 8 elements keeps 16 consecutive lanes on distinct banks.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_m;
__global__ __launch_bounds__(256) void conv_to_mono_bf16_kernel(const __bf16* X, long long x_bs, int T, int KC, int C, const float* w,
                                                                const float* bias, float* out, long long out_bs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mono_smem[];
    float* wsh = (float*)mono_smem;                       // [KC]
    __bf16* xs = (__bf16*)(wsh + KC);                     // [256 + k - 1][C + 8]
    const int k = KC / C, ld = C + 8, rows = 256 + k - 1, ppr = C / 8;
    const int b = blockIdx.y, t0 = blockIdx.x * 256;
    for (int i = threadIdx.x; i < KC; i += 256) wsh[i] = w[i];
    const __bf16* x = X + (long long)b * x_bs + (long long)t0 * C;
    for (int i = threadIdx.x; i < rows * ppr; i += 256) {
        const int r = i / ppr, p = i % ppr;
        bf16x8_m v = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        if (t0 + r < T + k - 1) v = *(const bf16x8_m*)(x + (long long)r * C + 8 * p);
        *(bf16x8_m*)(xs + r * ld + 8 * p) = v;
    }
    __syncthreads();
    const int t = t0 + threadIdx.x;
    if (t >= T) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int tap = 0; tap < k; ++tap) {
        const __bf16* xr = xs + (threadIdx.x + tap) * ld;
        const float* wr = wsh + tap * C;
        for (int c = 0; c < C; c += 8) {
            const bf16x8_m v = *(const bf16x8_m*)(xr + c);
            const float4 wa = *(const float4*)(wr + c), wb = *(const float4*)(wr + c + 4);
            s0 += (float)v[0] * wa.x + (float)v[4] * wb.x; s1 += (float)v[1] * wa.y + (float)v[5] * wb.y;
            s2 += (float)v[2] * wa.z + (float)v[6] * wb.z; s3 += (float)v[3] * wa.w + (float)v[7] * wb.w;
        }
    }
    out[(long long)b * out_bs + t] = ((s0 + s1) + (s2 + s3)) + bias[0];
}

void launch_conv_to_mono(const void* X, int x_bf16, long long x_bs, int T, int KC, int C, const float* w, const float* bias,
                         float* out, long long out_bs, int B, hipStream_t s) {
    if (T <= 0 || B <= 0) return;
    if ((KC & 3) || (C & 3) || (x_bs & 3)) throw std::runtime_error("conv_to_mono: channel counts must be multiples of 4");
    if (x_bf16) {
        if ((C & 7) || (x_bs & 7) || KC % C) throw std::runtime_error("conv_to_mono: bf16 rows need channel counts that are multiples of 8");
        const size_t smem = (size_t)KC * 4 + (size_t)(256 + KC / C - 1) * (C + 8) * 2;
        if (smem > 64 * 1024) throw std::runtime_error("conv_to_mono: window too large for LDS");
        hipLaunchKernelGGL(conv_to_mono_bf16_kernel, dim3((T + 255) / 256, B), dim3(256), smem, s, (const __bf16*)X, x_bs, T, KC, C, w, bias, out,
                           out_bs);
    } else {
        hipLaunchKernelGGL(conv_to_mono_kernel, dim3((T + 255) / 256, B), dim3(256), KC * sizeof(float), s, (const float*)X, x_bs, T, KC, C, w,
                           bias, out, out_bs);
    }
}

__global__ __launch_bounds__(256) void conv_from_mono_kernel(const float* audio, long long a_bs, const int* len, int T, int k, int Co,
                                                             const float* w, const float* bias, float* raw, long long raw_bs,
                                                             float* elu, long long elu_bs) {
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)T * Co) return;
    const int t = (int)(i / Co), co = (int)(i % Co);
    const float* au = audio + (long long)b * a_bs;
    float v = bias[co];
    for (int tap = 0; tap < k; ++tap) {
        int src = t + tap - (k - 1);
        if (src >= 0) v += w[co * k + tap] * au[src];
    }
    if (len && t >= len[b]) v = 0.f;
    if (raw) raw[(long long)b * raw_bs + i] = v;
    if (elu) elu[(long long)b * elu_bs + i] = v > 0.f ? v : expm1f(v);
}
void launch_conv_from_mono(const float* audio, long long a_bs, const int* len, int T, int k, int Co, const float* w,
                           const float* bias, float* raw, long long raw_bs, float* elu, long long elu_bs, int B, hipStream_t s) {
    long long n = (long long)T * Co;
    if (n <= 0 || B <= 0) return;
    hipLaunchKernelGGL(conv_from_mono_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, s, audio, a_bs, len, T, k, Co, w,
                       bias, raw, raw_bs, elu, elu_bs);
}

// One thread per (batch element, channel) walks the H history rows in ascending order, so the copy is
// a correct forward memmove even when the chunk is shorter than the history (source rows overlap
// destination rows); threads never share a column.
__global__ void shift_history_kernel(float* buf, long long bs, int T, int H, int C, unsigned long long mask, int slot0) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (!((mask >> (slot0 + blockIdx.y)) & 1ull)) return;   // this slot's stream does not advance
    float* base = buf + (long long)blockIdx.y * bs;
    for (int h = 0; h < H; ++h) base[(long long)h * C + c] = base[(long long)(T + h) * C + c];
}
void launch_shift_history(float* buf, long long bs, int T, int H, int C, int B, hipStream_t s, unsigned long long mask, int slot0) {
    if (H <= 0 || C <= 0 || B <= 0 || T <= 0) return;
    if (slot0 + B > 64 && mask != ~0ull) throw std::runtime_error("shift_history: a slot mask covers 64 slots");
    hipLaunchKernelGGL(shift_history_kernel, dim3((C + 63) / 64, B), dim3(64), 0, s, buf, bs, T, H, C, slot0 + B > 64 ? ~0ull : mask, slot0 + B > 64 ? 0 : slot0);
}

// cemb is given as `splits` K-slice slabs [splits][B][C] of the conditioning product plus its bias; the slabs are
// added in slice order.
// A frame of the autoregressive loop is replayed as one hipGraph, so nothing in it may depend on a host-side frame
// index: the index lives in device memory.  Block 0 of this launch also hands the EOS logits of the current frame to
// the host through mapped pinned memory, raises the host-visible flag and advances the index.
// FRAG: the value leaves as its two bf16 terms, hi = bf16(v) and lo = bf16(v - hi), in MFMA fragment order (see
// launch_flow_mod_input's declaration): the product that consumes the rows loads each fragment with one 16-byte load
// and neither converts nor reorders.  Eight consecutive threads fill one 16-byte unit of each term.
template <bool FRAG>
__global__ void flow_mod_input_kernel(const float* te, const float* cemb, int splits, const float* bias, float* Y, int steps, int B, int C,
                                      FramePublish pub, unsigned* zero, int zero_words) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i * 4 < zero_words) *(uint4*)(zero + i * 4) = make_uint4(0u, 0u, 0u, 0u);   // 16 bytes a lane; zero_words is a multiple of 4
    if (i < (long long)steps * B * C) {
        int c = (int)(i % C);
        long long jb = i / C;
        int b = (int)(jb % B), j = (int)(jb / B);
        float e = cemb[(long long)b * C + c];
        for (int s = 1; s < splits; ++s) e += cemb[((long long)s * B + b) * C + c];
        if (bias) e += bias[c];
        float v = te[(long long)j * C + c] + e;
        v = v / (1.0f + expf(-v));
        if constexpr (FRAG) {
            const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
            const long long unit = ((jb / 16) * (C / 32) + c / 32) * 2;
            const long long at = (unit * 64 + (c % 32 / 8) * 16 + (int)(jb % 16)) * 8 + c % 8;
            ((__bf16*)Y)[at] = hi;
            ((__bf16*)Y)[at + 64 * 8] = lo;
        } else {
            Y[i] = v;
        }
    }
    if (pub.step && blockIdx.x == 0) {   // frame bookkeeping: see FramePublish
        const int st = *pub.step;
        const long long row = (long long)(pub.ring > 0 ? st % pub.ring : st) * (pub.stride > 0 ? pub.stride : B);
        for (int b = threadIdx.x; b < B; b += blockDim.x) pub.h_eos_all[row + b] = pub.eos[b];
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            if (pub.h_ring_err) *pub.h_ring_err = *pub.ring_err;
            __hip_atomic_store(pub.h_flag, st + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            *pub.step = st + 1;
        }
    }
}
void launch_flow_mod_input(const float* te, const float* cemb, int splits, const float* bias, float* Y, int steps, int B, int C, hipStream_t s,
                           const FramePublish* pub, unsigned* zero, int zero_words, int frag) {
    long long n = (long long)steps * B * C;
    if (n <= 0) return;
    if (zero_words % 4 || zero_words / 4 > n) throw std::runtime_error("flow_mod_input: cannot zero that many control words");
    if (frag && (C % 32 || ((long long)steps * B) % 16)) throw std::runtime_error("flow_mod_input: fragment order needs C % 32 == 0 and rows % 16 == 0");
    if (frag)
        hipLaunchKernelGGL(flow_mod_input_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, te, cemb, splits, bias, Y, steps, B, C,
                           pub ? *pub : FramePublish(), zero, zero ? zero_words : 0);
    else
    hipLaunchKernelGGL(flow_mod_input_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, te, cemb, splits, bias, Y, steps, B, C,
                       pub ? *pub : FramePublish(), zero, zero ? zero_words : 0);
}

}  // namespace ptts
