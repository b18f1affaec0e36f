// Kernel micro-benchmarks: times single launches of the hot kernels at the shapes the B=64 frame loop and the
// decoder use, with HIP events over many back-to-back launches.  Build: make -C pocket-tts.cpp_amd kbench
// Usage: kbench [case ...]     cases: skinny_qkv skinny_ff1 skinny_out skinny_ff2 gemm_ada gemm_dec_qkv gemm_dec_up2
//                                     gemm_dec_c1_2 gemm_conv0 flow attn_decode
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <algorithm>
#include <string>
#include <vector>

#include "../csrc/kernels.h"

using namespace ptts;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

static void* dalloc(size_t bytes, int fill = 1) {
    void* p = nullptr;
    CK(hipMalloc(&p, bytes));
    if (fill) {
        std::vector<float> h(bytes / 4);
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
        CK(hipMemcpy(p, h.data(), bytes / 4 * 4, hipMemcpyHostToDevice));
    } else {
        CK(hipMemset(p, 0, bytes));
    }
    return p;
}

static double time_it(const char* name, int iters, const std::function<void(hipStream_t)>& fn, double bytes, double flops) {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) fn(st);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < iters; ++i) fn(st);
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / iters;
    std::printf("%-16s %9.2f us/launch  %8.1f GB/s  %8.2f TFLOP/s\n", name, us, bytes / us * 1e-3, flops / us * 1e-6);
    CK(hipStreamDestroy(st));
    return us;
}

static std::vector<float> rnd(size_t n) {
    std::vector<float> h(n);
    unsigned s = 777u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.1f; }
    return h;
}

static void* packed_weight(int N, int K, int wtype) {
    auto w = rnd((size_t)N * K);
    auto bytes = pack_flow_weight(w.data(), N, K, wtype);
    void* p = nullptr;
    CK(hipMalloc(&p, bytes.size()));
    CK(hipMemcpy(p, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
    return p;
}

static void* bf16_weight(int N, int K) {
    auto w = rnd((size_t)N * K);
    std::vector<unsigned short> h(w.size());
    for (size_t i = 0; i < w.size(); ++i) { unsigned u; std::memcpy(&u, &w[i], 4); h[i] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
    void* p = nullptr;
    CK(hipMalloc(&p, h.size() * 2));
    CK(hipMemcpy(p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    return p;
}

static void bench_skinny(const char* name, int M, int N, int K, int splits, int mt = 0, int kch = 0) {
    SkinnyArgs a;
    a.mtiles = mt; a.kchunk = kch;
    a.A = (const float*)dalloc((size_t)M * K * 4); a.lda = K;
    a.Wp = packed_weight(N, K, W_BF16); a.wtype = W_BF16;
    a.M = M; a.N = N; a.K = K; a.splits = splits;
    if (splits > 1) a.P = (float*)dalloc((size_t)splits * M * N * 4, 0);
    else { a.C = (float*)dalloc((size_t)M * N * 4, 0); a.ldc = N; }
    time_it(name, 300, [&](hipStream_t s) { launch_skinny_gemm(a, s); }, (double)N * K * 2 + (double)M * K * 4 + (double)M * N * 4 * splits, 2.0 * M * N * K);
}

static void bench_skinny_q8(const char* name, int M, int N, int K, int splits, int mt = 0) {
    auto w = rnd((size_t)N * K);
    std::vector<float> sc;
    auto bytes = pack_q8_weight(w.data(), N, K, sc);
    void* wp = nullptr;
    CK(hipMalloc(&wp, bytes.size()));
    CK(hipMemcpy(wp, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
    float* sp = (float*)dalloc(sc.size() * 4);
    SkinnyArgs a;
    a.A = (const float*)dalloc((size_t)M * K * 4); a.lda = K;
    a.Wp = wp; a.wtype = W_I8; a.wscale = sp; a.mtiles = mt;
    a.M = M; a.N = N; a.K = K; a.splits = splits;
    if (splits > 1) a.P = (float*)dalloc((size_t)splits * M * N * 4, 0);
    else { a.C = (float*)dalloc((size_t)M * N * 4, 0); a.ldc = N; }
    time_it(name, 300, [&](hipStream_t s) { launch_skinny_gemm(a, s); }, (double)N * K + (double)M * K * 4 + (double)M * N * 4 * splits, 2.0 * M * N * K);
}

// int8 product on activations quantised by their producer; cq: the result leaves quantised too (the FFN-1 form)
static void bench_skinny_q8p(const char* name, int M, int N, int K, int splits, bool cq) {
    auto w = rnd((size_t)N * K);
    std::vector<float> sc;
    auto bytes = pack_q8_weight(w.data(), N, K, sc);
    void* wp = nullptr;
    CK(hipMalloc(&wp, bytes.size()));
    CK(hipMemcpy(wp, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
    float* sp = (float*)dalloc(sc.size() * 4);
    const int Mq = (M + 15) & ~15;
    SkinnyArgs a;
    a.Aq = (const signed char*)dalloc((size_t)M * K); a.ldaq = K; a.Ad = (const float*)dalloc((size_t)(K / 64) * Mq * 4); a.Mq = Mq;
    a.Wp = wp; a.wtype = W_I8; a.wscale = sp;
    a.M = M; a.N = N; a.K = K; a.splits = splits;
    if (cq) { a.Cq = (signed char*)dalloc((size_t)M * N, 0); a.Cd = (float*)dalloc((size_t)(N / 64) * Mq * 4, 0); a.act = ACT_GELU; }
    else if (splits > 1) a.P = (float*)dalloc((size_t)splits * M * N * 4, 0);
    else { a.C = (float*)dalloc((size_t)M * N * 4, 0); a.ldc = N; }
    time_it(name, 300, [&](hipStream_t s) { launch_skinny_gemm(a, s); }, (double)N * K + (double)M * K + (double)M * N * 4 * splits, 2.0 * M * N * K);
}

static void bench_gemm(const char* name, int M, int N, int K, int split_a, int lda = 0) {
    GemmArgs g;
    if (!lda) lda = K;
    g.A = (const float*)dalloc(((size_t)M * lda + K) * 4); g.lda = lda;
    g.W = bf16_weight(N, K); g.wtype = W_BF16;
    g.C = (float*)dalloc((size_t)M * N * 4, 0); g.ldc = N;
    g.M = M; g.N = N; g.K = K; g.split_a = split_a;
    // KB_GEMM_ABF / KB_GEMM_CBF: rows in / out as bf16, as the decoder's late stages and transformer hold them
    const int abf = !split_a && std::getenv("KB_GEMM_ABF") ? 1 : 0, cbf = !split_a && std::getenv("KB_GEMM_CBF") ? 1 : 0;
    g.a_bf16 = abf; g.c_bf16 = cbf;
    time_it(name, 50, [&](hipStream_t s) { launch_gemm(g, s); }, (double)N * K * 2 + (double)M * lda * (abf ? 2 : 4) + (double)M * N * (cbf ? 2 : 4), 2.0 * M * N * K);
}


// ---- MFMA issue rate with the operand pattern of wide_rows_kernel: 64 row fragments of 4 registers each held per wave
// (256 registers, so about half of them live in AccVGPRs), 16 column fragments per "tile", 64 MFMAs per tile on two
// accumulators.  LDSB = 1: the 16 column fragments are re-read from LDS for every tile, as in the kernel.
typedef __attribute__((ext_vector_type(8))) __bf16 kb_bf16x8;
typedef __attribute__((ext_vector_type(4))) float kb_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned kb_u32x4;
template <int LDSB>
__global__ __launch_bounds__(256, 1) void mfma_rate_kernel(const kb_u32x4* src, float* out, int tiles) {
    __shared__ __attribute__((aligned(16))) kb_u32x4 Wl[16 * 64];
    const int lane = threadIdx.x & 63;
    kb_bf16x8 ah[2][16], al[2][16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            ah[mt][ks] = __builtin_bit_cast(kb_bf16x8, src[((mt * 16 + ks) * 2 + 0) * 64 + lane]);
            al[mt][ks] = __builtin_bit_cast(kb_bf16x8, src[((mt * 16 + ks) * 2 + 1) * 64 + lane]);
        }
    kb_u32x4 bw[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) { bw[ks] = src[(64 + ks) * 64 + lane]; Wl[ks * 64 + lane] = bw[ks]; }
    __syncthreads();
    kb_f32x4 acc[2] = {kb_f32x4{0.f, 0.f, 0.f, 0.f}, kb_f32x4{0.f, 0.f, 0.f, 0.f}};
    for (int t = 0; t < tiles; ++t) {
        if (LDSB) {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) bw[ks] = Wl[ks * 64 + ((lane + t) & 63)];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const kb_bf16x8 b8 = __builtin_bit_cast(kb_bf16x8, bw[ks]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mt][ks], b8, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt][ks], b8, acc[mt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[0][1] + acc[0][2] + acc[0][3] + acc[1][0] + acc[1][1] + acc[1][2] + acc[1][3];
}

static void bench_mfma_rate() {
    const int tiles = 2000, wgs = 256;
    void* src = dalloc((size_t)(64 + 16) * 64 * 16 + 4096);
    float* out = (float*)dalloc((size_t)wgs * 256 * 4, 0);
    for (int ldsb = 0; ldsb < 2; ++ldsb)
        for (int n : {wgs, 64}) {
            hipEvent_t a, b;
            CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            auto run = [&] {
                if (ldsb) hipLaunchKernelGGL(mfma_rate_kernel<1>, dim3(n), dim3(256), 0, nullptr, (const kb_u32x4*)src, out, tiles);
                else hipLaunchKernelGGL(mfma_rate_kernel<0>, dim3(n), dim3(256), 0, nullptr, (const kb_u32x4*)src, out, tiles);
            };
            run();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a, nullptr));
            run();
            CK(hipEventRecord(b, nullptr));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            const double ns_per_mfma = ms * 1e6 / ((double)tiles * 64);
            std::printf("mfma_rate lds=%d workgroups=%3d: %.2f ns per MFMA and wave (%.1f cycles at 2.4 GHz), %.0f TFLOP/s\n", ldsb, n, ns_per_mfma,
                        ns_per_mfma * 2.4, (double)n * 4 * tiles * 64 * 16384.0 / (ms * 1e-3) * 1e-12);
        }
}

static void bench_flow(int B) {
    const char* wenv = std::getenv("KB_FLOW_WTYPE");
    const int W_T = wenv && !std::strcmp(wenv, "f32") ? W_F32 : W_BF16;
    const bool q8 = wenv && !std::strcmp(wenv, "i8");   // int8 residual blocks, bf16 in / out projections
    const int C = 512, L = 32, nb = std::getenv("KB_FLOW_BLOCKS") ? std::atoi(std::getenv("KB_FLOW_BLOCKS")) : 6, steps = 10, mw = nb * 3 * C + 2 * C;
    FlowHostTabs tabs;
    for (int i = 0; i < nb; ++i) {
        if (q8) {
            for (int k = 0; k < 2; ++k) {
                auto w = rnd((size_t)C * C);
                std::vector<float> sc;
                auto bytes = pack_q8_weight(w.data(), C, C, sc);
                void* wp = nullptr;
                CK(hipMalloc(&wp, bytes.size()));
                CK(hipMemcpy(wp, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
                float* sp = nullptr;
                CK(hipMalloc(&sp, sc.size() * 4));
                CK(hipMemcpy(sp, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
                (k ? tabs.w_fc2 : tabs.w_fc1).push_back(wp);
                (k ? tabs.s_fc2 : tabs.s_fc1).push_back(sp);
            }
        } else {
            tabs.w_fc1.push_back(packed_weight(C, C, W_T));
            tabs.w_fc2.push_back(packed_weight(C, C, W_T));
            tabs.s_fc1.push_back(nullptr); tabs.s_fc2.push_back(nullptr);
        }
        tabs.b_fc1.push_back((const float*)dalloc(C * 4)); tabs.b_fc2.push_back((const float*)dalloc(C * 4));
        tabs.ln_g.push_back((const float*)dalloc(C * 4)); tabs.ln_b.push_back((const float*)dalloc(C * 4));
    }
    FlowHeadArgs a;
    a.x0 = (const float*)dalloc((size_t)B * L * 4); a.x0_stride = L;
    a.x_out = (float*)dalloc((size_t)B * L * 4, 0);
    a.mods = (const float*)dalloc((size_t)steps * B * mw * 4);
    a.B = B; a.L = L; a.C = C; a.blocks = nb; a.steps = steps; a.mod_width = mw;
    a.w_in = packed_weight(C, L, W_T); a.b_in = (const float*)dalloc(C * 4);
    a.w_out = packed_weight(L, C, W_T); a.b_out = (const float*)dalloc(L * 4);
    if (W_T == W_BF16) { tabs.r_fc1 = tabs.w_fc1; tabs.r_fc2 = tabs.w_fc2; tabs.r_in = a.w_in; tabs.r_out = a.w_out; }   // the persistent form's planes
    if (W_T == W_BF16 && !q8 && std::getenv("KB_FLOW_IO2")) { tabs.r_in_lo = a.w_in; tabs.r_out_lo = a.w_out; }   // timing only: second planes = copies of the first
    a.host_tabs = &tabs; a.workspace = (float*)dalloc(flow_head_workspace_bytes(B, C, W_T, nb), 0); a.wtype = W_T;
    if (q8) a.block_wtype = W_I8;
    flow_head_prepare(C, W_T);
    if (std::getenv("KB_FLOW_RING")) {
        // persistent form against the launch chain: same inputs, outputs compared, then timed (its control words are cleared
        // by a memset ahead of every launch here; the engine clears them in an earlier kernel of the frame)
        const size_t ctl_bytes = flow_ring_ctl_bytes(B, nb);
        unsigned* ctl = (unsigned*)dalloc(ctl_bytes, 0);
        std::vector<float> ref((size_t)B * L), got((size_t)B * L);
        if (!q8) setenv("FLOW_HEAD_SPLIT", "8", 1);   // the chain variant whose sums are formed in the same order
        launch_flow_head(a, 0);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ref.data(), a.x_out, ref.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(a.x_out, 0, ref.size() * 4));
        a.ring_ctl = ctl;
        launch_flow_head(a, 0);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), a.x_out, got.size() * 4, hipMemcpyDeviceToHost));
        std::vector<unsigned> hc(ctl_bytes / 4);
        CK(hipMemcpy(hc.data(), ctl, ctl_bytes, hipMemcpyDeviceToHost));
        double worst = 0, mag = 0;
        size_t differ = 0;
        for (size_t i = 0; i < ref.size(); ++i) { worst = std::max(worst, (double)std::fabs(ref[i] - got[i])); mag = std::max(mag, (double)std::fabs(ref[i])); differ += ref[i] != got[i]; }
        std::printf("flow_ring vs chain (8 slices): max-abs diff %.3e (largest value %.3f), %zu of %zu elements differ, error word %u\n", worst, mag, differ,
                    ref.size(), hc[0]);
        if (hc[0] == 0) {
            const double usr = time_it("flow_ring", 30, [&](hipStream_t s) { CK(hipMemsetAsync(ctl, 0, ctl_bytes, s)); launch_flow_head(a, s); }, 0,
                                       2.0 * B * steps * (2.0 * L * C + nb * 2.0 * C * C));
            std::printf("   = %.2f us per hand-off over %d (memset included)\n", usr / (steps * nb), steps * nb);
        }
        {
            long long* st = (long long*)dalloc(16 * 8, 0);
            a.cycle_probe = st;
            CK(hipMemset(ctl, 0, ctl_bytes));
            launch_flow_head(a, 0);
            CK(hipDeviceSynchronize());
            long long h[16];
            CK(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
            std::printf("   visit cycles: poll %lld, barrier %lld, loads+residual %lld, norm+barrier %lld, fc1+silu+barrier %lld, fc2+stores %lld, drain %lld, barrier+signal %lld, total %lld\n",
                        h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[8] - h[7], h[8] - h[0]);
            a.cycle_probe = nullptr;
        }
        a.ring_ctl = nullptr;
        unsetenv("FLOW_HEAD_SPLIT");
    }
    const double us = time_it("flow_chain", 30, [&](hipStream_t s) { launch_flow_head(a, s); }, 0, 2.0 * B * steps * (2.0 * L * C + nb * 2.0 * C * C));
    std::printf("   = %.2f us per launch over %d launches\n", us / (steps * (nb + 1)), steps * (nb + 1));
    long long* cyc = (long long*)dalloc(64, 0);
    a.cycle_probe = cyc;
    launch_flow_head(a, 0);
    CK(hipDeviceSynchronize());
    long long h[8];
    CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
    std::printf("   cycles: loads-issued->residual %lld, norm %lld, sync %lld, fc1 %lld, sync %lld, fc2+store %lld, total %lld\n", h[1] - h[0], h[2] - h[1],
                h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[6] - h[0]);
}

static void bench_attn_decode(int ctx, int kv16, int cap = 256) {
    const int M = 64, d = 1024, heads = 16, splits = 4;
    AttnDecodeArgs a;
    a.qkv = (const float*)dalloc((size_t)splits * M * 3 * d * 4); a.qkv_splits = splits;
    a.kcache = dalloc((size_t)M * cap * d * 4); a.vcache = dalloc((size_t)M * cap * d * 4); a.kv16 = kv16;
    std::vector<int> pos(M, ctx);
    int* dp = nullptr; CK(hipMalloc(&dp, M * 4)); CK(hipMemcpy(dp, pos.data(), M * 4, hipMemcpyHostToDevice));
    a.pos = dp; a.M = M; a.d = d; a.heads = heads; a.cap = cap;
    a.cos_tab = (const float*)dalloc(1024 * 32 * 4); a.sin_tab = (const float*)dalloc(1024 * 32 * 4); a.tab_len = 1024;
    a.out = (float*)dalloc((size_t)M * d * 4, 0);
    char nm[64];
    std::snprintf(nm, sizeof nm, "attn_%s_ctx%d", kv16 ? "bf16" : "f32", ctx);
    time_it(nm, 300, [&](hipStream_t s) { launch_attn_decode(a, s); },
            2.0 * M * ctx * d * (kv16 ? 2 : 4) + (double)splits * M * 3 * d * 4, 4.0 * M * ctx * d);
}

int main(int argc, char** argv) {
    std::vector<std::string> cases;
    for (int i = 1; i < argc; ++i) cases.push_back(argv[i]);
    auto want = [&](const char* c) {
        if (cases.empty()) return true;
        for (auto& s : cases) if (s == c) return true;
        return false;
    };
    CK(hipSetDevice(0));
    if (want("skinny_qkv")) bench_skinny("skinny_qkv", 64, 3072, 1024, 1);
    if (want("skinny_ff1")) bench_skinny("skinny_ff1", 64, 4096, 1024, 1);
    if (want("skinny_out")) bench_skinny("skinny_out", 64, 1024, 1024, 4);
    if (want("skinny_ff2")) bench_skinny("skinny_ff2", 64, 1024, 4096, 16);
    if (want("skinny_q8_qkv")) bench_skinny_q8("skinny_q8_qkv", 64, 3072, 1024, 1);
    if (want("skinny_q8_ff2")) bench_skinny_q8("skinny_q8_ff2", 64, 1024, 4096, 16);
    if (want("gemm_ada")) bench_gemm("gemm_ada", 640, 10240, 512, 1);
    if (want("gemm_pre_qkv")) bench_gemm("gemm_pre_qkv", 6016, 3072, 1024, 1);
    if (want("gemm_pre_ff2")) bench_gemm("gemm_pre_ff2", 6016, 1024, 4096, 1);
    if (want("gemm_dec_qkv")) bench_gemm("gemm_dec_qkv", 15360, 1536, 512, 0);
    if (want("gemm_dec_ff2")) bench_gemm("gemm_dec_ff2", 15360, 512, 2048, 0);
    if (want("gemm_conv0")) bench_gemm("gemm_conv0", 15360, 512, 3584, 0, 512);
    if (want("gemm_dec_up1")) bench_gemm("gemm_dec_up1", 92160, 640, 512, 0, 256);
    if (want("gemm_dec_up2")) bench_gemm("gemm_dec_up2", 460800, 256, 256, 0, 128);
    if (want("gemm_dec_c1_2")) bench_gemm("gemm_dec_c1_2", 1843200, 32, 192, 0, 64);
    if (want("gemm_dec_c2_2")) bench_gemm("gemm_dec_c2_2", 1843200, 64, 32, 0, 32);
    if (want("skinny_sweep")) {
        for (int mt : {4, 2, 1})
            for (int sp : {1, 2, 4}) {
                char nm[64];
                std::snprintf(nm, sizeof nm, "qkv mt%d sp%d", mt, sp); bench_skinny(nm, 64, 3072, 1024, sp, mt);
                std::snprintf(nm, sizeof nm, "ff1 mt%d sp%d", mt, sp); bench_skinny(nm, 64, 4096, 1024, sp, mt);
            }
        for (int mt : {4, 2, 1})
            for (int sp : {2, 4, 8, 16}) {
                char nm[64];
                std::snprintf(nm, sizeof nm, "out mt%d sp%d", mt, sp); bench_skinny(nm, 64, 1024, 1024, sp, mt);
                std::snprintf(nm, sizeof nm, "ff2 mt%d sp%d", mt, sp); bench_skinny(nm, 64, 1024, 4096, sp, mt);
            }
    }
    if (want("skinny_q8_sweep")) {
        for (int mt : {4, 2, 1}) {
            char nm[64];
            std::snprintf(nm, sizeof nm, "q8 qkv mt%d sp1", mt); bench_skinny_q8(nm, 64, 3072, 1024, 1, mt);
            std::snprintf(nm, sizeof nm, "q8 qkv mt%d sp4", mt); bench_skinny_q8(nm, 64, 3072, 1024, 4, mt);
            std::snprintf(nm, sizeof nm, "q8 ff1 mt%d sp1", mt); bench_skinny_q8(nm, 64, 4096, 1024, 1, mt);
            std::snprintf(nm, sizeof nm, "q8 out mt%d sp4", mt); bench_skinny_q8(nm, 64, 1024, 1024, 4, mt);
            std::snprintf(nm, sizeof nm, "q8 ff2 mt%d sp8", mt); bench_skinny_q8(nm, 64, 1024, 4096, 8, mt);
        }
    }
    if (want("skinny_q8p")) {
        bench_skinny_q8p("q8p qkv sp4", 64, 3072, 1024, 4, false);
        bench_skinny_q8p("q8p qkv sp2", 64, 3072, 1024, 2, false);
        bench_skinny_q8p("q8p qkv sp1", 64, 3072, 1024, 1, false);
        bench_skinny_q8p("q8p out sp4", 64, 1024, 1024, 4, false);
        bench_skinny_q8p("q8p out sp2", 64, 1024, 1024, 2, false);
        bench_skinny_q8p("q8p ff1 cq", 64, 4096, 1024, 1, true);
        bench_skinny_q8p("q8p ff1 f32", 64, 4096, 1024, 1, false);
        bench_skinny_q8p("q8p ff2 sp8", 64, 1024, 4096, 8, false);
        bench_skinny_q8p("q8p ff2 sp4", 64, 1024, 4096, 4, false);
        bench_skinny_q8p("q8p ff2 sp16", 64, 1024, 4096, 16, false);
    }
    if (want("mfma_rate")) bench_mfma_rate();
    if (want("ada_dump")) {   // the adaLN product's result bytes into $KB_DUMP (compare POCKET_TTS_WIDE_ROWS=1 / 0 with cmp); KB_DUMP_M rows
        const int M = std::getenv("KB_DUMP_M") ? std::atoi(std::getenv("KB_DUMP_M")) : 640, N = 10240, K = 512;
        SkinnyArgs a;
        a.A = (const float*)dalloc((size_t)M * K * 4); a.lda = K;
        a.Wp = packed_weight(N, K, W_BF16); a.wtype = W_BF16;
        a.M = M; a.N = N; a.K = K; a.splits = 1; a.mtiles = 2; a.kchunk = 128;
        a.bias = (const float*)dalloc((size_t)N * 4);
        a.C = (float*)dalloc((size_t)M * N * 4, 0); a.ldc = N;
        launch_skinny_gemm(a, nullptr);
        CK(hipDeviceSynchronize());
        std::vector<float> h((size_t)M * N);
        CK(hipMemcpy(h.data(), a.C, h.size() * 4, hipMemcpyDeviceToHost));
        double sum = 0;
        for (float v : h) sum += std::fabs(v);
        std::printf("ada_dump M=%d: sum|C| = %.6f\n", M, sum);
        if (const char* path = std::getenv("KB_DUMP")) {
            FILE* f = std::fopen(path, "wb");
            if (f) { std::fwrite(h.data(), 4, h.size(), f); std::fclose(f); }
        }
    }
    if (want("ada")) {   // the adaLN product as the flow phase launches it (with its bias)
        const int M = std::getenv("KB_ADA_M") ? std::atoi(std::getenv("KB_ADA_M")) : 640, N = std::getenv("KB_ADA_N") ? std::atoi(std::getenv("KB_ADA_N")) : 10240, K = 512;
        SkinnyArgs a;
        a.A = (const float*)dalloc((size_t)M * K * 4); a.lda = K;
        a.Wp = packed_weight(N, K, W_BF16); a.wtype = W_BF16;
        a.M = M; a.N = N; a.K = K; a.splits = 1; a.mtiles = 2; a.kchunk = 128;
        a.bias = (const float*)dalloc((size_t)N * 4);
        a.C = (float*)dalloc((size_t)M * N * 4, 0); a.ldc = N;
        time_it("ada product", 300, [&](hipStream_t s) { launch_skinny_gemm(a, s); }, (double)N * K * 2 + (double)M * K * 4 + (double)M * N * 4, 2.0 * M * N * K);
    }
    if (want("skinny_ada")) {
        for (int mt : {4, 2, 1})
            for (int kch : {128, 256}) {
                char nm[64];
                std::snprintf(nm, sizeof nm, "ada mt%d k%d", mt, kch); bench_skinny(nm, 640, 10240, 512, 1, mt, kch);
            }
        bench_skinny("cond sp4 auto", 64, 512, 1024, 4);
        bench_skinny("cond sp8 auto", 64, 512, 1024, 8);
        bench_skinny("cond sp1 mt1", 64, 512, 1024, 1, 1, 512);
        bench_skinny("cond sp2 mt1", 64, 512, 1024, 2, 1, 512);
    }
    if (want("skinny_kch")) {
        for (int mt : {2, 1})
            for (int kch : {128, 256, 512}) {
                if (mt == 2 && kch == 512) continue;
                char nm[64];
                std::snprintf(nm, sizeof nm, "qkv mt%d k%d sp1", mt, kch); bench_skinny(nm, 64, 3072, 1024, 1, mt, kch);
                std::snprintf(nm, sizeof nm, "qkv mt%d k%d sp2", mt, kch); bench_skinny(nm, 64, 3072, 1024, 2, mt, kch);
                std::snprintf(nm, sizeof nm, "ff1 mt%d k%d sp1", mt, kch); bench_skinny(nm, 64, 4096, 1024, 1, mt, kch);
                std::snprintf(nm, sizeof nm, "out mt%d k%d sp4", mt, kch); bench_skinny(nm, 64, 1024, 1024, 4, mt, kch);
                std::snprintf(nm, sizeof nm, "out mt%d k%d sp8", mt, kch); bench_skinny(nm, 64, 1024, 1024, 8, mt, kch);
                std::snprintf(nm, sizeof nm, "ff2 mt%d k%d sp8", mt, kch); bench_skinny(nm, 64, 1024, 4096, 8, mt, kch);
                std::snprintf(nm, sizeof nm, "ff2 mt%d k%d sp16", mt, kch); bench_skinny(nm, 64, 1024, 4096, 16, mt, kch);
            }
    }
    if (want("attn_decode")) { bench_attn_decode(156, 1); bench_attn_decode(156, 0); bench_attn_decode(250, 1, 640); bench_attn_decode(300, 1, 640); bench_attn_decode(500, 1, 640); }
    if (want("flow")) bench_flow(std::getenv("KB_FLOW_B") ? std::atoi(std::getenv("KB_FLOW_B")) : 64);
    return 0;
}
