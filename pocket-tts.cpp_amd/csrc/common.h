// Shared declarations for the MI355X Pocket TTS engine: error handling, the model spec, the weight
// pack reader and small device-memory helpers.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace ptts {

#define PTTS_HIP(expr)                                                                              \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " + \
                                     __FILE__ + ":" + std::to_string(__LINE__) + " (" #expr ")");   \
    } while (0)

// Set POCKET_TTS_TRACE=1 to follow engine set-up, tear-down and scheduler activations on stderr.
void trace_line(const char* what);
bool trace_on();
void trace_fmt(const char* fmt, ...) __attribute__((format(printf, 1, 2)));   // with a microsecond time stamp
#define PTTS_TRACE(what) ::ptts::trace_line(what)
#define PTTS_TRACEF(...) do { if (::ptts::trace_on()) ::ptts::trace_fmt(__VA_ARGS__); } while (0)

// Every dimension of the five networks; read from the pack header, never compiled in.
struct Spec {
    int latent = 0, d_model = 0, lm_layers = 0, lm_heads = 0, lm_ffn = 0, lm_max_period = 0, vocab = 0;
    int flow_dim = 0, flow_blocks = 0, flow_freq = 0;
    int mimi_dim = 0, mimi_layers = 0, mimi_heads = 0, mimi_ffn = 0, mimi_context = 0, mimi_max_period = 0;
    int n_filters = 0, ratios[3] = {0, 0, 0}, kernel = 0, last_kernel = 0, res_kernel = 0, upsample = 0;
    int sample_rate = 0;
    int q8_act_scheme = 0;   // int8 engine's activation scheme: 0 per row and 64-block symmetric int8, 1 ONNX dynamic uint8 with zero point
    float lm_eps = 0, flow_eps = 0, mimi_eps = 0, flow_max_period = 0;
    int frame_size() const { return upsample * ratios[0] * ratios[1] * ratios[2]; }
};

struct HostTensor {
    const float* data = nullptr;
    std::vector<int64_t> shape;
    size_t numel() const {
        size_t n = 1;
        for (auto d : shape) n *= static_cast<size_t>(d);
        return n;
    }
};

// int8 form of a matrix carried by the pack beside its f32 tensor (`<name>.q8`, `<name>.q8s`): values [N][K] and one
// scale per output row, W ~ q * scale.
struct Q8Tensor {
    const int8_t* q = nullptr;
    const float* scale = nullptr;
    int N = 0, K = 0;
};

// Read-only view of a .ptts weight pack held in host memory.
class Pack {
public:
    explicit Pack(const std::string& path);
    const Spec& spec() const { return spec_; }
    const HostTensor& get(const std::string& name) const;
    bool has(const std::string& name) const { return tensors_.count(name) != 0; }
    const Q8Tensor* q8(const std::string& name) const {
        auto it = q8_.find(name);
        return it == q8_.end() ? nullptr : &it->second;
    }

private:
    std::vector<unsigned char> blob_;
    Spec spec_;
    std::map<std::string, HostTensor> tensors_;
    std::map<std::string, Q8Tensor> q8_;
};

// Owning device buffer.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    explicit DevBuf(size_t n) { alloc(n); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t n) {
        release();
        if (n == 0) n = 256;
        PTTS_HIP(hipMalloc(&p, n));
        bytes = n;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    float* f() const { return static_cast<float*>(p); }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

// Blocking copies and fills on a stream of their own.  The plain forms (hipMemcpy, hipMemset) run on the legacy stream,
// and the runtime refuses any operation there while another thread of the process is capturing a graph ("would make the
// legacy stream depend on a capturing blocking stream"): engines on several GPUs, or several on one, capture their frame
// graphs while others are being created or are growing their buffers.
void sync_copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
// The stream those run on (one per device for the whole process), for asynchronous side work that must not sit in front of
// an engine's frame loop: the slot table's prefill passes.  Never destroyed.
hipStream_t shared_side_stream();
void sync_fill(void* dst, int value, size_t bytes);

}  // namespace ptts
