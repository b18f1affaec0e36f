// Tuning aid: where the workgroups of a one-per-CU launch land (XCD of each workgroup index), and what one hand-off
// between two resident workgroups costs by scope of the memory operations:
//   mode 0: store and polling load with sc1 (visible across XCDs: the persistent flow head's hand-off)
//   mode 1: plain store and polling load with sc0 (the XCD's own L2 is the meeting point: same-XCD pairs only)
//   mode 2: agent-scope atomic add and atomic load
// Every spin is bounded; a pair that never sees its partner reports "no hand-off".
//   hipcc --offload-arch=gfx950 -O2 tools/xcd_probe.hip -o xcd_probe && ./xcd_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF; }   // HW_REG_XCC_ID[3:0]

template <int AUX> __device__ __forceinline__ unsigned ld(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, AUX);
}
template <int AUX> __device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned v) {
    __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, AUX);
}

__global__ __launch_bounds__(256) void where_kernel(unsigned* xcc) {
    extern __shared__ unsigned char lds[];
    if (threadIdx.x == 0) { lds[0] = 1; xcc[blockIdx.x] = xcc_id(); }
}

// workgroups a and b bounce a counter `iters` times; a reports wall-clock ticks (100 MHz) for the whole exchange
__global__ __launch_bounds__(256) void pingpong_kernel(unsigned* flags, int a, int b, int mode, int iters, long long* ticks, unsigned* ok) {
    extern __shared__ unsigned char lds[];
    const int wg = blockIdx.x;
    if (threadIdx.x != 0 || (wg != a && wg != b)) return;
    lds[0] = 1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(flags, 0, 4096, 0x00020000);
    const unsigned mine = wg == a ? 0u : 256u, theirs = wg == a ? 256u : 0u;   // byte offsets, separate lines
    const long long t0 = wall_clock64();
    bool good = true;
    for (int it = 1; it <= iters && good; ++it) {
        if (wg == a) {
            if (mode == 0) st<16>(rs, mine, it); else if (mode == 1) st<0>(rs, mine, it);
            else __hip_atomic_fetch_add(flags + mine / 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned spins = 0, v = 0;
        do {
            if (mode == 0) v = ld<16>(rs, theirs); else if (mode == 1) v = ld<1>(rs, theirs);
            else v = __hip_atomic_load(flags + theirs / 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while (v < (unsigned)it && ++spins < 200000u);
        if (v < (unsigned)it) good = false;
        if (wg == b && good) {
            if (mode == 0) st<16>(rs, mine, it); else if (mode == 1) st<0>(rs, mine, it);
            else __hip_atomic_fetch_add(flags + mine / 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (wg == a) { *ticks = wall_clock64() - t0; *ok = good ? 1u : 0u; }
}

int main() {
    const int grid = 192, lds = 96 * 1024;   // the flow head's footprint: one workgroup per CU
    CHECK(hipFuncSetAttribute((const void*)where_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void*)pingpong_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    unsigned *d_xcc, *d_flags, *d_ok;
    long long* d_ticks;
    CHECK(hipMalloc(&d_xcc, grid * 4));
    CHECK(hipMalloc(&d_flags, 4096));
    CHECK(hipMalloc(&d_ok, 4));
    CHECK(hipMalloc(&d_ticks, 8));
    std::vector<unsigned> xcc(grid);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(where_kernel, dim3(grid), dim3(256), lds, 0, d_xcc);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(xcc.data(), d_xcc, grid * 4, hipMemcpyDeviceToHost));
        int rr = 0;
        for (int i = 0; i < grid; ++i) rr += xcc[i] == xcc[i % 8] ? 1 : 0;
        std::printf("launch %d: XCD of workgroups 0..15:", rep);
        for (int i = 0; i < 16; ++i) std::printf(" %u", xcc[i]);
        std::printf("   | workgroups whose XCD equals that of (index %% 8): %d of %d\n", rr, grid);
    }
    const int iters = 2000;
    const int pairs[3][2] = {{0, 8}, {1, 2}, {3, 100}};
    for (int mode = 0; mode < 3; ++mode)
        for (auto& p : pairs) {
            CHECK(hipMemset(d_flags, 0, 4096));
            hipLaunchKernelGGL(pingpong_kernel, dim3(grid), dim3(256), lds, 0, d_flags, p[0], p[1], mode, iters, d_ticks, d_ok);
            CHECK(hipDeviceSynchronize());
            long long t = 0;
            unsigned ok = 0;
            CHECK(hipMemcpy(&t, d_ticks, 8, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&ok, d_ok, 4, hipMemcpyDeviceToHost));
            std::printf("mode %d (%s) workgroups %3d <-> %3d (XCD %u, %u): %s, %.3f us per hop\n", mode,
                        mode == 0 ? "sc1 store / sc1 load" : mode == 1 ? "plain store / sc0 load" : "agent atomics", p[0], p[1], xcc[p[0]], xcc[p[1]],
                        ok ? "ok" : "no hand-off", t * 0.01 / (2.0 * iters));
        }
    return 0;
}
