// Weight-pack (.ptts) reader.  The pack replaces the five ONNX files the reference opens in
// loadModels() (/root/reference/src/pocket_tts.cpp:89-113); layout is documented in pyhost/ptts_pack.py.
#include "common.h"

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace ptts {

namespace {
struct RawEntry {
    char name[64];
    uint32_t dtype, ndim;
    uint64_t shape[4], offset, nbytes;
};
static_assert(sizeof(RawEntry) == 120, "pack entry layout");
constexpr size_t kHeaderSize = 16 + 64 * 4 + 16 * 4;
}  // namespace

bool trace_on() {
    static const bool on = std::getenv("POCKET_TTS_TRACE") != nullptr;
    return on;
}

void trace_line(const char* what) {
    if (trace_on()) std::fprintf(stderr, "[pocket-tts] %s\n", what);
}

void trace_fmt(const char* fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    const auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    std::fprintf(stderr, "[pocket-tts %lld.%03lld ms] %s\n", static_cast<long long>(us / 1000 % 100000), static_cast<long long>(us % 1000), buf);
}

namespace {
// One stream per device, made on first use and kept for the life of the process; the mutex also orders the callers.  One
// for the whole process on purpose: a process gets few hardware queues per GPU (GPU_MAX_HW_QUEUES, 4 by default), streams
// beyond that share queues, and two streams on one queue run one after the other (measured: two engines with a side stream
// each fell from 7550 to 4180 RTFx).  The engines' frame-loop streams should have queues to themselves.
std::mutex g_util_mu;
hipStream_t g_util_stream[64];

hipStream_t util_stream_locked() {
    int dev = 0;
    PTTS_HIP(hipGetDevice(&dev));
    hipStream_t& s = g_util_stream[dev & 63];
    if (!s) PTTS_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
}
}  // namespace

hipStream_t shared_side_stream() {
    std::lock_guard<std::mutex> g(g_util_mu);
    return util_stream_locked();
}

void sync_copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (bytes == 0) return;
    std::lock_guard<std::mutex> g(g_util_mu);
    hipStream_t s = util_stream_locked();
    PTTS_HIP(hipMemcpyAsync(dst, src, bytes, kind, s));
    PTTS_HIP(hipStreamSynchronize(s));
}

void sync_fill(void* dst, int value, size_t bytes) {
    if (bytes == 0) return;
    std::lock_guard<std::mutex> g(g_util_mu);
    hipStream_t s = util_stream_locked();
    PTTS_HIP(hipMemsetAsync(dst, value, bytes, s));
    PTTS_HIP(hipStreamSynchronize(s));
}

Pack::Pack(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open weight pack: " + path);
    std::fseek(f, 0, SEEK_END);
    long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (sz < static_cast<long>(kHeaderSize)) {
        std::fclose(f);
        throw std::runtime_error("weight pack too small: " + path);
    }
    blob_.resize(static_cast<size_t>(sz));
    size_t got = std::fread(blob_.data(), 1, blob_.size(), f);
    std::fclose(f);
    if (got != blob_.size()) throw std::runtime_error("short read on weight pack: " + path);
    if (std::memcmp(blob_.data(), "PTTSPK01", 8) != 0) throw std::runtime_error("bad magic in weight pack: " + path);
    uint32_t ver = 0, n = 0;
    std::memcpy(&ver, blob_.data() + 8, 4);
    std::memcpy(&n, blob_.data() + 12, 4);
    if (ver != 1) throw std::runtime_error("unsupported weight pack version");
    const int32_t* iv = reinterpret_cast<const int32_t*>(blob_.data() + 16);
    const float* fv = reinterpret_cast<const float*>(blob_.data() + 16 + 256);
    Spec& s = spec_;
    s.latent = iv[0]; s.d_model = iv[1]; s.lm_layers = iv[2]; s.lm_heads = iv[3]; s.lm_ffn = iv[4];
    s.lm_max_period = iv[5]; s.vocab = iv[6]; s.flow_dim = iv[7]; s.flow_blocks = iv[8]; s.flow_freq = iv[9];
    s.mimi_dim = iv[10]; s.mimi_layers = iv[11]; s.mimi_heads = iv[12]; s.mimi_ffn = iv[13];
    s.mimi_context = iv[14]; s.mimi_max_period = iv[15]; s.n_filters = iv[16];
    s.ratios[0] = iv[17]; s.ratios[1] = iv[18]; s.ratios[2] = iv[19];
    s.kernel = iv[20]; s.last_kernel = iv[21]; s.res_kernel = iv[22]; s.upsample = iv[23]; s.sample_rate = iv[24]; s.q8_act_scheme = iv[25];
    s.lm_eps = fv[0]; s.flow_eps = fv[1]; s.mimi_eps = fv[2]; s.flow_max_period = fv[3];
    if (kHeaderSize + static_cast<size_t>(n) * sizeof(RawEntry) > blob_.size())
        throw std::runtime_error("weight pack tensor table truncated");
    for (uint32_t i = 0; i < n; ++i) {
        RawEntry e;
        std::memcpy(&e, blob_.data() + kHeaderSize + i * sizeof(RawEntry), sizeof(RawEntry));
        if (e.dtype > 1) throw std::runtime_error("weight pack: unknown element type");
        if (e.offset + e.nbytes > blob_.size() || e.ndim > 4) throw std::runtime_error("weight pack: bad tensor entry");
        if (e.dtype == 1) {   // int8 matrix "<name>.q8"
            char qn[65];
            std::memcpy(qn, e.name, 64);
            qn[64] = 0;
            std::string base(qn);
            if (e.ndim != 2 || base.size() < 4 || base.compare(base.size() - 3, 3, ".q8") != 0 || e.nbytes != e.shape[0] * e.shape[1])
                throw std::runtime_error("weight pack: bad int8 entry " + base);
            Q8Tensor& q = q8_[base.substr(0, base.size() - 3)];
            q.q = reinterpret_cast<const int8_t*>(blob_.data() + e.offset);
            q.N = static_cast<int>(e.shape[0]);
            q.K = static_cast<int>(e.shape[1]);
            continue;
        }
        HostTensor t;
        t.data = reinterpret_cast<const float*>(blob_.data() + e.offset);
        for (uint32_t d = 0; d < e.ndim; ++d) t.shape.push_back(static_cast<int64_t>(e.shape[d]));
        char nm[65];
        std::memcpy(nm, e.name, 64);
        nm[64] = 0;
        tensors_[nm] = t;
    }
    // scales of the int8 forms; a form without its scales (or with the wrong count) is dropped with an error
    for (auto it = q8_.begin(); it != q8_.end(); ++it) {
        auto sc = tensors_.find(it->first + ".q8s");
        if (sc == tensors_.end() || sc->second.numel() != static_cast<size_t>(it->second.N))
            throw std::runtime_error("weight pack: int8 form of " + it->first + " has no matching scales");
        it->second.scale = sc->second.data;
        auto f = tensors_.find(it->first);
        if (f == tensors_.end() || f->second.shape.size() != 2 || f->second.shape[0] != it->second.N || f->second.shape[1] != it->second.K)
            throw std::runtime_error("weight pack: int8 form of " + it->first + " does not match its f32 tensor");
    }
}

const HostTensor& Pack::get(const std::string& name) const {
    auto it = tensors_.find(name);
    if (it == tensors_.end()) throw std::runtime_error("weight pack: missing tensor " + name);
    return it->second;
}

}  // namespace ptts
