// Flow phase for callers that cannot allocate device memory (code under HIP graph capture, serving loops that
// forbid hipMalloc): the scratch that Engine::op_flow_euler takes from a pooled Batch (conditioning slabs, adaLN
// input rows, adaLN vectors, the head's residual / partial-sum workspace) lives in one static device array here.
//
// The array's name is a host-side stand-in, not a device address.  Each GPU holds its own copy at its own address;
// Engine::init_static_scratch asks for that address once per engine, with the engine's device current, and every
// later use (kernel arguments) goes through the pointer it got.
//
// Every call on one GPU shares that GPU's array and finds the previous call's contents in it.  The only words that
// must start a call at zero are the persistent head's counters, which the call's second launch clears before the
// head starts, and its error word, whose place moves with the row count and which the call clears first.  Every other area is written in full by the launch that produces it before any launch reads it.  Calls that
// overlap on one GPU would race on the array; a process-wide mutex per device serialises the calls made through
// this library.
#include <mutex>

#include "engine.h"

namespace ptts {

namespace {

constexpr size_t kStaticScratchBytes = static_cast<size_t>(32) << 20;   // 32 MiB per GPU, held until the process ends

static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

std::mutex& device_mutex(int device) {
    static std::mutex mus[64];
    return mus[device & 63];
}

}  // namespace

size_t Engine::static_scratch_capacity() { return kStaticScratchBytes; }

// Called from the Engine constructor with the engine's device current.
void Engine::init_static_scratch() {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p)
        throw std::runtime_error(std::string("static scratch array has no address on this device: ") + hipGetErrorString(e));
    static_scratch_ = static_cast<unsigned char*>(p);
}

// cond_dev[B][d_model], x0_dev[B][latent] -> x_out_dev[B][latent], all device memory of this engine's GPU.
// Returns when the result is complete.  No hipMalloc / hipFree / allocating routine is called.
void Engine::op_flow_euler_static(const float* cond_dev, const float* x0_dev, int B, float* x_out_dev) {
    if (B <= 0) throw std::runtime_error("flow_euler_static: rows must be positive");
    if (!cond_dev || !x0_dev || !x_out_dev || x_out_dev == x0_dev) throw std::runtime_error("flow_euler_static: three distinct device buffers are required");
    if (!static_scratch_) throw std::runtime_error("flow_euler_static: the static scratch array was not set up");
    size_t off[6];
    flow_scratch_layout(B, off);
    if (off[5] > kStaticScratchBytes)
        throw std::runtime_error("flow_euler_static: " + std::to_string(B) + " rows need " + std::to_string(off[5]) +
                                 " bytes of scratch, the static array holds " + std::to_string(kStaticScratchBytes));
    std::lock_guard<std::mutex> dev_guard(device_mutex(opt_.device));
    std::lock_guard<std::mutex> guard(mu_);
    PTTS_HIP(hipSetDevice(opt_.device));
    unsigned char* base = static_scratch_;
    FlowScratch fs{reinterpret_cast<float*>(base + off[0]), reinterpret_cast<float*>(base + off[1]),
                   reinterpret_cast<float*>(base + off[2]), reinterpret_cast<float*>(base + off[3]),
                   reinterpret_cast<unsigned*>(base + off[4])};
    PTTS_HIP(hipMemsetAsync(fs.ctl, 0, 4, stream_));   // the error word: this spot may hold another row count's workspace
    flow_core(fs, stream_, cond_dev, x0_dev, spec().latent, x_out_dev, B, nullptr, 0, nullptr, nullptr, nullptr);
    PTTS_HIP(hipStreamSynchronize(stream_));
    check_flow_ring(fs.ctl, B);
}

}  // namespace ptts
