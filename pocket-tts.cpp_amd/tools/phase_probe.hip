// What a phase boundary costs INSIDE one persistent launch, measured on the shapes an LM frame step as one launch would
// have: a grid of one workgroup per CU walks `phases` phases; between phases every workgroup must see what every other one
// wrote (the step's products are all-to-all: each output tile needs every row of the previous phase's result).
//
//   mode 0  one counter per phase: every storing wave drains its stores, workgroup barrier, one lane adds (agent scope),
//           one lane polls with sc1 loads + s_sleep, workgroup barrier            (the flow ring's hand-off, grid-wide)
//   mode 1  XCD-hierarchical: workgroup L adds to the counter of XCD L % 8; the last arriver of an XCD adds to the top
//           counter; everybody polls the top counter for 8
//   payload each workgroup stores `pay` bytes (sc1, whole lines) before the barrier and loads `gather` bytes of other
//           workgroups' payload after it (sc1), as a consumer of the previous phase's rows would
//   work    each phase also streams `wbytes` bytes of "weights" per workgroup from a large buffer (plain loads), requested
//           BEFORE the barrier when prefetch = 1 (what a launch chain cannot do), after it when prefetch = 0
//
// Prints microseconds per phase.  Compare with a dependent kernel boundary (kbench `boundary`: the same grid as `phases`
// empty launches) to see what a persistent step would have to win back through prefetch.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

namespace {

typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
constexpr int kLine = 32;              // words between counters
constexpr unsigned kSpin = 4000000u;   // bounded: a probe that cannot finish reports it instead of hanging

struct ProbeArgs {
    unsigned* ctl;          // [phases][9] counters, one 128-byte line each; word 0 of the block after them = error
    float* pay;             // [grid][pay bytes]
    const float* weights;   // large buffer
    float* sink;
    int phases, mode, pay_bytes, gather_bytes, wbytes, prefetch;
    long long wtotal;       // floats in weights
};

__device__ __forceinline__ bool wait_for(unsigned* cnt, unsigned target, unsigned* err) {
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        if (spins >= kSpin) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        if ((spins & 63u) == 63u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

__global__ __launch_bounds__(256) void phase_probe_kernel(ProbeArgs a) {
    __shared__ int ok;
    const int tid = threadIdx.x, wg = blockIdx.x, G = gridDim.x;
    unsigned* err = a.ctl + (size_t)a.phases * 9 * kLine;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.pay, 0, (int)((size_t)G * a.pay_bytes), 0x00020000);
    float acc = 0.f;
    const int wvec = a.wbytes / 16 / 256;                  // 16-byte loads per thread and phase
    float4 wreg[8];
    for (int p = 0; p < a.phases; ++p) {
        const float4* wsrc = (const float4*)a.weights + (((long long)p * G + wg) * (a.wbytes / 16)) % (a.wtotal / 4 - a.wbytes / 16);
        if (a.prefetch)
            for (int i = 0; i < wvec && i < 8; ++i) wreg[i] = wsrc[i * 256 + tid];
        // ---- publish this phase's rows
        for (int off = tid * 16; off < a.pay_bytes; off += 256 * 16) {
            const u32x4 v = {(unsigned)p, (unsigned)wg, (unsigned)off, 1u};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, (unsigned)((size_t)wg * a.pay_bytes + off), 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            unsigned* base = a.ctl + (size_t)p * 9 * kLine;
            bool good;
            if (a.mode == 0) {
                __hip_atomic_fetch_add(base, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = wait_for(base, (unsigned)G, err);
            } else {
                const int x = wg & 7;
                const unsigned members = (unsigned)((G - x + 7) / 8);
                const unsigned old = __hip_atomic_fetch_add(base + (1 + x) * kLine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == members) __hip_atomic_fetch_add(base, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = wait_for(base, (unsigned)(G < 8 ? G : 8), err);
            }
            ok = good ? 1 : 0;
        }
        __syncthreads();
        if (!ok) return;
        // ---- consume: rows of other workgroups, then this phase's weights
        for (int off = tid * 16; off < a.gather_bytes; off += 256 * 16) {
            const int src = (wg + 1 + off / a.pay_bytes) % G;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((size_t)src * a.pay_bytes + off % a.pay_bytes), 0, 16);
            acc += (float)(v[0] + v[3]);
        }
        if (!a.prefetch)
            for (int i = 0; i < wvec && i < 8; ++i) wreg[i] = wsrc[i * 256 + tid];
        for (int i = 0; i < wvec && i < 8; ++i) acc += wreg[i].x + wreg[i].w;
    }
    if (acc == 12345.678f) a.sink[wg] = acc;
}

__global__ void empty_kernel(float* sink) { if (sink && threadIdx.x == 9999) sink[0] = 1.f; }

}  // namespace

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    const int phases = 43;
    const long long wtotal = 96ll << 20;   // 384 MB of floats: more than the Infinity Cache
    float *weights, *pay, *sink;
    unsigned* ctl;
    CK(hipMalloc(&weights, wtotal * 4));
    CK(hipMemset(weights, 0, wtotal * 4));
    CK(hipMalloc(&pay, 256 * 65536));
    CK(hipMalloc(&sink, 4096));
    CK(hipMalloc(&ctl, ((size_t)phases * 9 + 1) * kLine * 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int grid, int mode, int pay_b, int gather_b, int wbytes, int prefetch) {
        ProbeArgs a{ctl, pay, weights, sink, phases, mode, pay_b, gather_b, wbytes, prefetch, wtotal};
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemsetAsync(ctl, 0, ((size_t)phases * 9 + 1) * kLine * 4, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(phase_probe_kernel, dim3(grid), dim3(256), 0, st, a);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        unsigned err = 0;
        CK(hipMemcpy(&err, ctl + (size_t)phases * 9 * kLine, 4, hipMemcpyDeviceToHost));
        std::printf("%-58s grid %3d  %7.2f us per phase  (%7.1f us for %d phases)%s\n", name, grid, best * 1e3 / phases, best * 1e3, phases, err ? "  GAVE UP" : "");
    };
    for (int grid : {256, 128}) {
        run("barrier only, one counter", grid, 0, 0, 0, 0, 0);
        run("barrier only, per-XCD counters", grid, 1, 0, 0, 0, 0);
        run("+ 1 KB published, 16 KB gathered per workgroup", grid, 1, 1024, 16384, 0, 0);
        run("+ 32 KB of weights per workgroup, requested after the barrier", grid, 1, 1024, 16384, 32768, 0);
        run("+ 32 KB of weights per workgroup, requested before the barrier", grid, 1, 1024, 16384, 32768, 1);
    }
    // the alternative: the same number of dependent (empty) launches on one stream
    for (int grid : {256, 512}) {
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, st, sink);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        std::printf("%-58s grid %3d  %7.2f us per launch (%7.1f us for %d launches)\n", "dependent empty launches on one stream (eager)", grid, best * 1e3 / phases, best * 1e3, phases);
    }
    return 0;
}
