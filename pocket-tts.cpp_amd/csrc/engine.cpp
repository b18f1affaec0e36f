// Model upload, per-batch state and the forward passes of the five networks on one MI355X.
#include "engine.h"

#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstring>

namespace ptts {

// A kernel launch reports a bad configuration (grid, LDS size, missing code object) only through the thread's last
// error; the passes below ask for it once at their end so that such a launch cannot go by unnoticed.
void check_launches(const char* where) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw std::runtime_error(std::string("kernel launch failed in ") + where + ": " + hipGetErrorString(e));
}

// ------------------------------------------------------------------------------------------ Model

namespace {

bool ends_with(const std::string& s, const std::string& suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

// GEMM operands get a low-precision copy; tensors consumed by the scalar kernels stay f32 only.
bool is_gemm_weight(const std::string& name) {
    if (!ends_with(name, ".w")) return false;
    // the reference keeps the voice encoder in fp32 at every precision setting (src/pocket_tts.cpp:92)
    if (name.rfind("enc.", 0) == 0) return false;
    return name != "dec.up.w" && name != "dec.final.w" && name != "lm.eos.w";
}

size_t align256(size_t n) { return (n + 255) & ~static_cast<size_t>(255); }

// v -> bf16(v) as a float, round to nearest even
float bf16_part(float v) {
    uint32_t u;
    std::memcpy(&u, &v, 4);
    if ((u & 0x7f800000u) != 0x7f800000u) u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    std::memcpy(&v, &u, 4);
    return v;
}

}  // namespace

// Matrix families the reduced-precision engines keep as TWO bf16 terms, W = bf16(W) + bf16(W - bf16(W)), i.e. to ~16
// mantissa bits (a third MFMA per fragment), instead of one.  Stored once in bf16, a matrix carries a relative error of
// 2^-9 per element.  Measured on the float32 model with one family rounded at a time (tools/bf16_attribution.py) no family
// owns the bf16 engine's distance from float32: eleven families contribute 3e-3 .. 1.4e-2 each to the per-frame latent
// and add up in quadrature.  The default set is the one that buys the most per microsecond of frame time
// (tools/bf16_attribution_gpu.py, bench A/B): the small projections at both ends of the LM step and of every Euler
// step, and the conditioning and adaLN products of the flow phase.  The LM's FFN-2 is the next candidate (its skinny
// product nearly doubles in time) and, like the LM's QKV and output projections, can be switched on:
// POCKET_TTS_BF16_TWO_TERM = "none", "default" or a comma list of
// lm.in_proj, lm.qkv, lm.out, lm.ff2, flow.cond, flow.ada, flow.in_proj, flow.final.lin.
std::set<std::string> two_term_families(Precision prec, bool int8_flow_blocks) {
    static const char* known[] = {"lm.in_proj", "lm.qkv", "lm.out", "lm.ff2", "flow.cond", "flow.ada", "flow.in_proj", "flow.final.lin"};
    std::set<std::string> out;
    if (prec == Precision::FP32) return out;
    const char* e = std::getenv("POCKET_TTS_BF16_TWO_TERM");
    std::string v = e ? e : "default";
    if (v == "none" || v == "0") return out;
    if (v == "default" || v == "1") v = "lm.in_proj,flow.cond,flow.ada,flow.in_proj,flow.final.lin";
    size_t at = 0;
    while (at <= v.size()) {
        const size_t c = v.find(',', at);
        const std::string name = v.substr(at, c == std::string::npos ? std::string::npos : c - at);
        at = c == std::string::npos ? v.size() + 1 : c + 1;
        if (name.empty()) continue;
        bool ok = false;
        for (const char* k : known) ok = ok || name == k;
        if (!ok) throw std::runtime_error("POCKET_TTS_BF16_TWO_TERM: unknown matrix family " + name);
        out.insert(name);
    }
    if (prec == Precision::INT8) {   // these are int8 matrices there
        out.erase("lm.qkv"); out.erase("lm.out"); out.erase("lm.ff2");
        // the persistent head with int8 blocks has no LDS left for second planes of its projections
        if (int8_flow_blocks) { out.erase("flow.in_proj"); out.erase("flow.final.lin"); }
    }
    return out;
}

Model::Model(const Pack& pack, Precision prec) : spec_(pack.spec()), prec_(prec) {
    const Spec& s = spec_;
    if (s.d_model != s.lm_heads * 64 || s.mimi_dim != s.mimi_heads * 64)
        throw std::runtime_error("model: attention head dimension must be 64");
    if (s.flow_dim % 64 || s.latent % 32) throw std::runtime_error("model: flow_dim must be a multiple of 64 and latent_dim of 32");
    // names in a deterministic order
    std::vector<std::string> names;
    {
        // Pack exposes lookup only; rebuild the list from the known naming scheme by probing.
        auto add = [&](const std::string& n) { if (pack.has(n)) names.push_back(n); else throw std::runtime_error("model: pack lacks " + n); };
        add("text.embed"); add("lm.bos_emb"); add("lm.in_proj.w");
        for (int i = 0; i < s.lm_layers; ++i) {
            std::string p = "lm.l" + std::to_string(i) + ".";
            for (const char* t : {"ln1.g", "ln1.b", "qkv.w", "out.w", "ln2.g", "ln2.b", "ff1.w", "ff2.w"}) add(p + t);
        }
        for (const char* t : {"lm.out_norm.g", "lm.out_norm.b", "lm.eos.w", "lm.eos.b", "flow.in_proj.w", "flow.in_proj.b"}) add(t);
        for (int j = 0; j < 2; ++j) {
            std::string p = "flow.time" + std::to_string(j) + ".";
            for (const char* t : {"l1.w", "l1.b", "l2.w", "l2.b"}) add(p + t);
        }
        add("flow.cond.w"); add("flow.cond.b");
        for (int i = 0; i < s.flow_blocks; ++i) {
            std::string p = "flow.b" + std::to_string(i) + ".";
            for (const char* t : {"ada.w", "ada.b", "ln.g", "ln.b", "fc1.w", "fc1.b", "fc2.w", "fc2.b"}) add(p + t);
        }
        for (const char* t : {"flow.final.ada.w", "flow.final.ada.b", "flow.final.lin.w", "flow.final.lin.b", "mimi.emb_mean", "mimi.emb_std",
                              "dec.in_proj.w", "dec.up.w"}) add(t);
        auto tf = [&](const std::string& pre) {
            for (int i = 0; i < s.mimi_layers; ++i) {
                std::string p = pre + ".t" + std::to_string(i) + ".";
                for (const char* t : {"ln1.g", "ln1.b", "qkv.w", "out.w", "ls1", "ln2.g", "ln2.b", "ff1.w", "ff2.w", "ls2"}) add(p + t);
            }
        };
        tf("dec");
        add("dec.conv0.w"); add("dec.conv0.b");
        for (int i = 0; i < 3; ++i) {
            std::string p = "dec.s" + std::to_string(i) + ".";
            for (const char* t : {"up.w", "up.b", "res.c1.w", "res.c1.b", "res.c2.w", "res.c2.b"}) add(p + t);
        }
        add("dec.final.w"); add("dec.final.b"); add("enc.conv0.w"); add("enc.conv0.b");
        for (int i = 0; i < 3; ++i) {
            std::string p = "enc.s" + std::to_string(i) + ".";
            for (const char* t : {"res.c1.w", "res.c1.b", "res.c2.w", "res.c2.b", "down.w", "down.b"}) add(p + t);
        }
        add("enc.final.w"); add("enc.final.b");
        tf("enc");
        add("enc.down.w"); add("enc.spk_proj.w");
    }
    // layout: f32 blob holds everything; low-precision blob holds GEMM weights as bf16
    size_t f32_bytes = 0, lp_bytes = 0;
    for (auto& n : names) {
        const HostTensor& t = pack.get(n);
        Entry e;
        e.shape = t.shape;
        e.off_f32 = f32_bytes;
        f32_bytes += align256(t.numel() * 4);
        if (prec_ != Precision::FP32 && is_gemm_weight(n)) {
            e.has_lp = true;
            e.off_lp = lp_bytes;
            lp_bytes += align256(t.numel() * 2);
        }
        entries_[n] = e;
    }
    // derived: stacked adaLN weights and bias
    const int C = s.flow_dim;
    mod_width = s.flow_blocks * 3 * C + 2 * C;
    const size_t ada_w_elems = static_cast<size_t>(mod_width) * C;
    size_t ada_off_f32 = f32_bytes;
    f32_bytes += align256(ada_w_elems * 4);
    size_t ada_b_off = f32_bytes;
    f32_bytes += align256(static_cast<size_t>(mod_width) * 4);
    size_t ada_off_lp = lp_bytes;
    if (prec_ != Precision::FP32) lp_bytes += align256(ada_w_elems * 2);

    std::vector<unsigned char> host(f32_bytes, 0);
    for (auto& n : names) {
        const HostTensor& t = pack.get(n);
        std::memcpy(host.data() + entries_[n].off_f32, t.data, t.numel() * 4);
    }
    {
        float* aw = reinterpret_cast<float*>(host.data() + ada_off_f32);
        float* ab = reinterpret_cast<float*>(host.data() + ada_b_off);
        size_t row = 0;
        for (int i = 0; i <= s.flow_blocks; ++i) {
            std::string p = i < s.flow_blocks ? "flow.b" + std::to_string(i) + ".ada" : "flow.final.ada";
            const HostTensor& w = pack.get(p + ".w");
            const HostTensor& b = pack.get(p + ".b");
            std::memcpy(aw + row * C, w.data, w.numel() * 4);
            std::memcpy(ab + row, b.data, b.numel() * 4);
            row += static_cast<size_t>(w.shape[0]);
        }
    }
    blob_f32_.alloc(f32_bytes);
    sync_copy(blob_f32_.p, host.data(), f32_bytes, hipMemcpyHostToDevice);
    if (prec_ != Precision::FP32) {
        std::vector<uint16_t> lp(lp_bytes / 2, 0);
        auto conv = [&](size_t off_lp, const float* src, size_t n) {
            uint16_t* d = lp.data() + off_lp / 2;
            for (size_t i = 0; i < n; ++i) d[i] = f32_to_bf16(src[i]);
        };
        for (auto& n : names)
            if (entries_[n].has_lp) conv(entries_[n].off_lp, pack.get(n).data, pack.get(n).numel());
        conv(ada_off_lp, reinterpret_cast<const float*>(host.data() + ada_off_f32), ada_w_elems);
        blob_lp_.alloc(lp_bytes);
        sync_copy(blob_lp_.p, lp.data(), lp_bytes, hipMemcpyHostToDevice);
        ada_all.p = static_cast<unsigned char*>(blob_lp_.p) + ada_off_lp;
        ada_all.wtype = W_BF16;
    } else {
        ada_all.p = static_cast<unsigned char*>(blob_f32_.p) + ada_off_f32;
        ada_all.wtype = W_F32;
    }
    ada_all.N = mod_width;
    ada_all.K = C;
    ada_all_b = reinterpret_cast<const float*>(static_cast<unsigned char*>(blob_f32_.p) + ada_b_off);

    // fused flow head: fragment-major copies of its sequential weights + per-block pointer tables
    const int nb = s.flow_blocks;
    flow_wtype = prec_ != Precision::FP32 ? W_BF16 : W_F32;
    const bool q8 = prec_ == Precision::INT8;
    lm_wtype = q8 ? W_I8 : flow_wtype;
    q8_act = s.q8_act_scheme == 1 ? 1 : 0;
    if (const char* e = std::getenv("POCKET_TTS_Q8_ACT")) q8_act = std::string(e) == "dyn" ? 1 : 0;
    if (!q8) q8_act = 0;
    // The int8 engine's flow blocks are bf16 unless POCKET_TTS_Q8_FLOW=1 asks for int8 blocks too (per-block scheme, flow_dim a
    // multiple of 256): measured on the 64 x 125 workload the int8 head is slower than the bf16 one (0.369 against 0.336 ms per
    // launch, 6156 against 6416 RTFx) and farther from the float model, so int8 blocks are an option, not the default.
    const bool q8_flow = [] { const char* e = std::getenv("POCKET_TTS_Q8_FLOW"); return e && std::atoi(e) != 0; }();
    flow_block_wtype = (q8 && q8_flow && q8_act == 0 && s.flow_dim % 256 == 0) ? W_I8 : flow_wtype;
    two_term = two_term_families(prec_, flow_block_wtype == W_I8);
    // second plane of a two-term matrix in the fragment-major layout of the skinny / flow kernels
    auto packed_lo = [&](const float* W, int N, int K) -> const void* {
        std::vector<float> lo(static_cast<size_t>(N) * K);
        for (size_t i = 0; i < lo.size(); ++i) lo[i] = W[i] - bf16_part(W[i]);
        auto bytes = pack_flow_weight(lo.data(), N, K, W_BF16);
        extra_.emplace_back(bytes.size());
        sync_copy(extra_.back().p, bytes.data(), bytes.size(), hipMemcpyHostToDevice);
        return extra_.back().p;
    };
    auto packed_lo_named = [&](const std::string& name) -> const void* {
        const HostTensor& t = pack.get(name);
        return packed_lo(t.data, static_cast<int>(t.shape[0]), static_cast<int>(t.shape[1]));
    };
    // ... and row-major, beside the bf16 blob's first plane, for the tiled GEMM of long prefill passes
    auto rowmajor_lo = [&](const std::string& name) {
        const HostTensor& t = pack.get(name);
        std::vector<uint16_t> lo(t.numel());
        for (size_t i = 0; i < lo.size(); ++i) lo[i] = f32_to_bf16(t.data[i] - bf16_part(t.data[i]));
        extra_.emplace_back(lo.size() * 2);
        sync_copy(extra_.back().p, lo.data(), lo.size() * 2, hipMemcpyHostToDevice);
        split_[name] = {static_cast<const unsigned char*>(blob_lp_.p) + entries_[name].off_lp, extra_.back().p};
    };
    // int8 form of a matrix: fragment-major int8 values + per-output-row scales, both uploaded
    auto packed_q8 = [&](const std::string& name, const float*& scale_dev, const int** wsums = nullptr) -> const void* {
        const HostTensor& t = pack.get(name);
        std::vector<float> sc;
        std::vector<unsigned char> bytes;
        std::vector<signed char> requant;
        const signed char* qv = nullptr;
        const int N = static_cast<int>(t.shape[0]), K = static_cast<int>(t.shape[1]);
        // a matrix that is int8 at the source is used as it is (values and per-row scales); POCKET_TTS_REQUANTIZE=1
        // quantises the f32 tensor again instead
        const Q8Tensor* carried = std::getenv("POCKET_TTS_REQUANTIZE") ? nullptr : pack.q8(name);
        if (carried) {
            qv = reinterpret_cast<const signed char*>(carried->q);
            sc.assign(carried->scale, carried->scale + carried->N);
        } else {
            requant = quantize_q8_rows(t.data, N, K, sc);
            qv = requant.data();
        }
        bytes = pack_q8_layout(qv, N, K);
        if (wsums && q8_act == 1) {   // weight row sums per K-slice, for every slice count the passes use
            for (int sp = 1; sp <= 16; ++sp) {
                if ((sp > 8 && sp != 16) || (K / 64) % sp) continue;
                const std::vector<int> sums = q8_row_sums(qv, N, K, sp);
                extra_.emplace_back(sums.size() * 4);
                sync_copy(extra_.back().p, sums.data(), sums.size() * 4, hipMemcpyHostToDevice);
                wsums[sp] = static_cast<const int*>(extra_.back().p);
            }
        }
        extra_.emplace_back(bytes.size());
        sync_copy(extra_.back().p, bytes.data(), bytes.size(), hipMemcpyHostToDevice);
        const void* wp = extra_.back().p;
        extra_.emplace_back(sc.size() * 4);
        sync_copy(extra_.back().p, sc.data(), sc.size() * 4, hipMemcpyHostToDevice);
        scale_dev = static_cast<const float*>(extra_.back().p);
        return wp;
    };
    auto packed = [&](const std::string& name) -> const void* {
        const HostTensor& t = pack.get(name);
        auto bytes = pack_flow_weight(t.data, static_cast<int>(t.shape[0]), static_cast<int>(t.shape[1]), flow_wtype);
        extra_.emplace_back(bytes.size());
        sync_copy(extra_.back().p, bytes.data(), bytes.size(), hipMemcpyHostToDevice);
        return extra_.back().p;
    };
    flow_in_packed = packed("flow.in_proj.w");
    flow_out_packed = packed("flow.final.lin.w");
    flow_cond_packed = packed("flow.cond.w");
    if (two_term.count("flow.cond")) flow_cond_packed_lo = packed_lo_named("flow.cond.w");
    {
        const HostTensor& t = pack.get("lm.in_proj.w");
        if (t.shape[0] % 16 == 0 && t.shape[1] % (flow_wtype == W_BF16 ? 32 : 16) == 0) lm_in_packed = packed("lm.in_proj.w");
    }
    {   // stacked adaLN matrix in the same fragment-major form (the skinny product is 2x faster than the tiled GEMM here)
        const int kstep = flow_wtype == W_BF16 ? 32 : 16;
        if (mod_width % 16 == 0 && C % kstep == 0) {
            auto bytes = pack_flow_weight(reinterpret_cast<const float*>(host.data() + ada_off_f32), mod_width, C, flow_wtype);
            extra_.emplace_back(bytes.size());
            sync_copy(extra_.back().p, bytes.data(), bytes.size(), hipMemcpyHostToDevice);
            ada_packed = extra_.back().p;
            if (two_term.count("flow.ada")) ada_packed_lo = packed_lo(reinterpret_cast<const float*>(host.data() + ada_off_f32), mod_width, C);
        }
    }
    // decoder matrices whose products are short and deep when one utterance streams (5 frames = 80 rows): the Mimi
    // transformer's and the first convolution's; fragment-major copies let those go to the skinny kernels
    {
        std::vector<std::string> small = {"dec.conv0.w"};
        for (int i = 0; i < s.mimi_layers; ++i)
            for (const char* t : {"qkv.w", "out.w", "ff1.w", "ff2.w"}) small.push_back("dec.t" + std::to_string(i) + "." + t);
        for (auto& n : small) {
            const HostTensor& t = pack.get(n);
            const int wt = entries_[n].has_lp ? W_BF16 : W_F32;
            if (t.shape.size() != 2 || t.shape[0] % 16 || t.shape[1] % (wt == W_BF16 ? 32 : 16)) continue;
            auto bytes = pack_flow_weight(t.data, static_cast<int>(t.shape[0]), static_cast<int>(t.shape[1]), wt);
            extra_.emplace_back(bytes.size());
            sync_copy(extra_.back().p, bytes.data(), bytes.size(), hipMemcpyHostToDevice);
            frag_[n] = extra_.back().p;
        }
    }
    // The voice encoder keeps f32 accuracy at every precision setting (src/pocket_tts.cpp:92).  Its matrices are held as two
    // bf16 terms each, W = hi + lo to ~16 mantissa bits, and multiplied by activations split the same way: three bf16 MFMAs
    // per tile (hi hi, lo hi, hi lo) in place of one f32 MFMA at a sixteenth of the bf16 rate.  POCKET_TTS_ENCODER_F32=1
    // keeps the f32 matrices and the f32 MFMA.
    if (!std::getenv("POCKET_TTS_ENCODER_F32") && s.n_filters % 8 == 0 && s.mimi_dim % 8 == 0 && s.d_model % 4 == 0) {
        std::vector<std::string> mats = {"enc.final.w", "enc.spk_proj.w"};
        for (int i = 0; i < 3; ++i)
            for (const char* t : {"res.c1.w", "res.c2.w", "down.w"}) mats.push_back("enc.s" + std::to_string(i) + "." + t);
        for (int i = 0; i < s.mimi_layers; ++i)
            for (const char* t : {"qkv.w", "out.w", "ff1.w", "ff2.w"}) mats.push_back("enc.t" + std::to_string(i) + "." + t);
        for (auto& n : mats) {
            const HostTensor& t = pack.get(n);
            if (t.shape.size() != 2 || t.shape[1] % 8) continue;
            const size_t el = t.numel();
            std::vector<uint16_t> hi(el), lo(el);
            auto to_bf16 = [](float v) {   // round to nearest even, as the device conversion does
                uint32_t u; std::memcpy(&u, &v, 4);
                if ((u & 0x7f800000u) == 0x7f800000u) return static_cast<uint16_t>(u >> 16);
                u += 0x7fffu + ((u >> 16) & 1u);
                return static_cast<uint16_t>(u >> 16);
            };
            for (size_t i = 0; i < el; ++i) {
                hi[i] = to_bf16(t.data[i]);
                const uint32_t hu = static_cast<uint32_t>(hi[i]) << 16;
                float hf; std::memcpy(&hf, &hu, 4);
                lo[i] = to_bf16(t.data[i] - hf);
            }
            extra_.emplace_back(el * 2);
            sync_copy(extra_.back().p, hi.data(), el * 2, hipMemcpyHostToDevice);
            const void* ph = extra_.back().p;
            extra_.emplace_back(el * 2);
            sync_copy(extra_.back().p, lo.data(), el * 2, hipMemcpyHostToDevice);
            split_[n] = {ph, extra_.back().p};
        }
    }
    {   // the encoder's down-sampling product stays on f32 weights (skinny kernel, fragment-major)
        const HostTensor& t = pack.get("enc.down.w");
        if (t.shape.size() == 2 && t.shape[0] % 16 == 0 && t.shape[1] % 16 == 0) {
            auto bytes = pack_flow_weight(t.data, static_cast<int>(t.shape[0]), static_cast<int>(t.shape[1]), W_F32);
            extra_.emplace_back(bytes.size());
            sync_copy(extra_.back().p, bytes.data(), bytes.size(), hipMemcpyHostToDevice);
            enc_ds_packed = extra_.back().p;
        }
    }
    for (int i = 0; i < s.lm_layers; ++i) {
        std::string p = "lm.l" + std::to_string(i) + ".";
        PackedLayer pl;
        if (q8) {
            pl.qkv = packed_q8(p + "qkv.w", pl.s_qkv, pl.ws_qkv); pl.out = packed_q8(p + "out.w", pl.s_out, pl.ws_out);
            pl.ff1 = packed_q8(p + "ff1.w", pl.s_ff1, pl.ws_ff1); pl.ff2 = packed_q8(p + "ff2.w", pl.s_ff2, pl.ws_ff2);
        } else {
            pl.qkv = packed(p + "qkv.w"); pl.out = packed(p + "out.w"); pl.ff1 = packed(p + "ff1.w"); pl.ff2 = packed(p + "ff2.w");
            if (two_term.count("lm.qkv")) { pl.qkv_lo = packed_lo_named(p + "qkv.w"); rowmajor_lo(p + "qkv.w"); }
            if (two_term.count("lm.out")) { pl.out_lo = packed_lo_named(p + "out.w"); rowmajor_lo(p + "out.w"); }
            if (two_term.count("lm.ff2")) { pl.ff2_lo = packed_lo_named(p + "ff2.w"); rowmajor_lo(p + "ff2.w"); }
        }
        lm_packed.push_back(pl);
    }
    for (int i = 0; i < nb; ++i) {
        std::string p = "flow.b" + std::to_string(i) + ".";
        const float *s1 = nullptr, *s2 = nullptr;
        const bool fq8 = flow_block_wtype == W_I8;
        flow_tabs.w_fc1.push_back(fq8 ? packed_q8(p + "fc1.w", s1) : packed(p + "fc1.w"));
        flow_tabs.w_fc2.push_back(fq8 ? packed_q8(p + "fc2.w", s2) : packed(p + "fc2.w"));
        flow_tabs.s_fc1.push_back(s1);
        flow_tabs.s_fc2.push_back(s2);
        flow_tabs.b_fc1.push_back(f(p + "fc1.b"));
        flow_tabs.b_fc2.push_back(f(p + "fc2.b"));
        flow_tabs.ln_g.push_back(f(p + "ln.g"));
        flow_tabs.ln_b.push_back(f(p + "ln.b"));
    }
    // Matrix planes of the persistent flow head.  bf16 engine: the fragment-major matrices as they are.  float engine: every
    // matrix as two bf16 planes, hi = bf16(W) and lo = bf16(W - hi), multiplied with two-term activations on three bf16
    // MFMAs per tile (about 16 mantissa bits; POCKET_TTS_FLOW_F32_CHAIN=1 keeps the float launch chain instead).
    if (flow_wtype == W_BF16) {   // bf16 or int8 blocks (the int8 form keeps its scales in s_fc1 / s_fc2)
        flow_tabs.r_in = flow_in_packed; flow_tabs.r_out = flow_out_packed;
        flow_tabs.r_fc1 = flow_tabs.w_fc1; flow_tabs.r_fc2 = flow_tabs.w_fc2;
        // second planes of the in / out projections: both or neither (a family that was not asked for gets a plane of zeros)
        if (two_term.count("flow.in_proj") || two_term.count("flow.final.lin")) {
            auto lo_or_zero = [&](const std::string& name, bool on) -> const void* {
                if (on) return packed_lo_named(name);
                const HostTensor& t = pack.get(name);
                extra_.emplace_back(t.numel() * 2);
                sync_fill(extra_.back().p, 0, t.numel() * 2);
                return extra_.back().p;
            };
            flow_tabs.r_in_lo = lo_or_zero("flow.in_proj.w", two_term.count("flow.in_proj") != 0);
            flow_tabs.r_out_lo = lo_or_zero("flow.final.lin.w", two_term.count("flow.final.lin") != 0);
        }
    } else if (flow_wtype == W_F32 && C == 512 && s.latent == 32 && !std::getenv("POCKET_TTS_FLOW_F32_CHAIN")) {
        auto planes = [&](const std::string& name, const void*& hi_dev, const void*& lo_dev) {
            const HostTensor& t = pack.get(name);
            const size_t el = t.numel();
            std::vector<float> hi(el), lo(el);
            for (size_t i = 0; i < el; ++i) {
                uint32_t u; std::memcpy(&u, &t.data[i], 4);
                if ((u & 0x7f800000u) != 0x7f800000u) u += 0x7fffu + ((u >> 16) & 1u);
                u &= 0xffff0000u;
                std::memcpy(&hi[i], &u, 4);
                lo[i] = t.data[i] - hi[i];
            }
            for (int k = 0; k < 2; ++k) {
                auto bytes = pack_flow_weight(k ? lo.data() : hi.data(), static_cast<int>(t.shape[0]), static_cast<int>(t.shape[1]), W_BF16);
                extra_.emplace_back(bytes.size());
                sync_copy(extra_.back().p, bytes.data(), bytes.size(), hipMemcpyHostToDevice);
                (k ? lo_dev : hi_dev) = extra_.back().p;
            }
        };
        planes("flow.in_proj.w", flow_tabs.r_in, flow_tabs.r_in_lo);
        planes("flow.final.lin.w", flow_tabs.r_out, flow_tabs.r_out_lo);
        for (int i = 0; i < nb; ++i) {
            const std::string p = "flow.b" + std::to_string(i) + ".";
            const void *h1, *l1, *h2, *l2;
            planes(p + "fc1.w", h1, l1);
            planes(p + "fc2.w", h2, l2);
            flow_tabs.r_fc1.push_back(h1); flow_tabs.r_fc1_lo.push_back(l1);
            flow_tabs.r_fc2.push_back(h2); flow_tabs.r_fc2_lo.push_back(l2);
        }
        flow_ring_split = true;
    }
}

void DevWeight::bind(GemmArgs& g, int M) const {
    g.W = p; g.wtype = wtype; g.W2 = nullptr;
    if (!lo) return;
    const long long tiles = static_cast<long long>((M + 127) / 128) * ((N + 127) / 128);
    if ((tiles >= 128 && N >= 128) || !f32) { g.W2 = lo; g.split_a = 1; }   // below 128 columns half of every tile would be idle
    else { g.W = f32; g.wtype = W_F32; g.split_a = 0; }
}

const float* Model::f(const std::string& name) const {
    auto it = entries_.find(name);
    if (it == entries_.end()) throw std::runtime_error("model: unknown tensor " + name);
    return reinterpret_cast<const float*>(static_cast<const unsigned char*>(blob_f32_.p) + it->second.off_f32);
}

DevWeight Model::w(const std::string& name) const {
    auto it = entries_.find(name);
    if (it == entries_.end()) throw std::runtime_error("model: unknown tensor " + name);
    const Entry& e = it->second;
    if (e.shape.size() != 2) throw std::runtime_error("model: " + name + " is not a matrix");
    DevWeight d;
    d.N = static_cast<int>(e.shape[0]);
    d.K = static_cast<int>(e.shape[1]);
    if (e.has_lp && !(name == "lm.in_proj.w" && two_term.count("lm.in_proj"))) {   // that small matrix is simply used in f32
        d.p = static_cast<const unsigned char*>(blob_lp_.p) + e.off_lp;
        d.wtype = W_BF16;
    } else {
        d.p = static_cast<const unsigned char*>(blob_f32_.p) + e.off_f32;
        d.wtype = W_F32;
    }
    auto fr = frag_.find(name);
    if (fr != frag_.end()) d.frag = fr->second;
    auto sp = split_.find(name);
    if (sp != split_.end()) {
        d.f32 = static_cast<const unsigned char*>(blob_f32_.p) + e.off_f32;
        d.p = sp->second.first; d.lo = sp->second.second; d.wtype = W_BF16;
    }
    return d;
}

size_t Model::gemm_weight_bytes(const std::string& prefix) const {
    size_t n = 0;
    for (auto& kv : entries_) {
        if (kv.first.compare(0, prefix.size(), prefix) != 0 || !is_gemm_weight(kv.first)) continue;
        size_t el = 1;
        for (auto d : kv.second.shape) el *= static_cast<size_t>(d);
        n += el * (kv.second.has_lp ? 2 : 4);
    }
    return n;
}

std::vector<TfLayer> Model::tf_layers(const std::string& prefix, int layers, bool has_ls) const {
    std::vector<TfLayer> out;
    for (int i = 0; i < layers; ++i) {
        std::string p = prefix + std::to_string(i) + ".";
        TfLayer l;
        l.ln1g = f(p + "ln1.g"); l.ln1b = f(p + "ln1.b"); l.ln2g = f(p + "ln2.g"); l.ln2b = f(p + "ln2.b");
        l.ls1 = has_ls ? f(p + "ls1") : nullptr;
        l.ls2 = has_ls ? f(p + "ls2") : nullptr;
        l.qkv = w(p + "qkv.w"); l.out = w(p + "out.w"); l.ff1 = w(p + "ff1.w"); l.ff2 = w(p + "ff2.w");
        out.push_back(l);
    }
    return out;
}

// cos/sin of pos * max_period^(-2i/64) in double, rounded once to f32: identical inputs to the
// rotation on every implementation that builds the table the same way.
void Model::build_rope_tables(int lm_len, int mimi_len) {
    const int half = 32;
    auto fill = [&](std::vector<float>& c, std::vector<float>& s, int len, int max_period) {
        c.resize(static_cast<size_t>(len) * half);
        s.resize(static_cast<size_t>(len) * half);
        for (int i = 0; i < half; ++i) {
            double freq = std::exp(-std::log(static_cast<double>(max_period)) * 2.0 * i / 64.0);
            for (int p = 0; p < len; ++p) {
                double a = static_cast<double>(p) * freq;
                c[static_cast<size_t>(p) * half + i] = static_cast<float>(std::cos(a));
                s[static_cast<size_t>(p) * half + i] = static_cast<float>(std::sin(a));
            }
        }
    };
    std::vector<float> lc, ls, mc, ms;
    fill(lc, ls, lm_len, spec_.lm_max_period);
    fill(mc, ms, mimi_len, spec_.mimi_max_period);
    size_t lb = align256(lc.size() * 4), mb = align256(mc.size() * 4);
    rope_.alloc(2 * lb + 2 * mb);
    auto* base = static_cast<unsigned char*>(rope_.p);
    sync_copy(base, lc.data(), lc.size() * 4, hipMemcpyHostToDevice);
    sync_copy(base + lb, ls.data(), ls.size() * 4, hipMemcpyHostToDevice);
    sync_copy(base + 2 * lb, mc.data(), mc.size() * 4, hipMemcpyHostToDevice);
    sync_copy(base + 2 * lb + mb, ms.data(), ms.size() * 4, hipMemcpyHostToDevice);
    lm_cos = reinterpret_cast<const float*>(base);
    lm_sin = reinterpret_cast<const float*>(base + lb);
    mimi_cos = reinterpret_cast<const float*>(base + 2 * lb);
    mimi_sin = reinterpret_cast<const float*>(base + 2 * lb + mb);
    lm_tab_len = lm_len;
    mimi_tab_len = mimi_len;
}

// ------------------------------------------------------------------------------------------ Batch

static constexpr int kDecChunk = 15;  // frames per decoder call, /root/reference/src/pocket_tts.cpp:461

Batch::Batch(Engine& e, int B_, bool lm_only_) : B(B_), Bw(B_), lm_only(lm_only_) {
    st = e.stream_; dec_st = e.dec_stream_; ready = e.latents_ready_;
    const Spec& s = e.spec();
    const EngineOptions& o = e.options();
    const int D = s.d_model, M = s.mimi_dim, L = s.latent, U = s.upsample;
    lm_cap = 64 + o.max_text_tokens + o.max_frames + 8;
    lm_cap = (lm_cap + 63) / 64 * 64;
    dec_cap = s.mimi_context + kDecChunk * U + 8;
    const int T0 = kDecChunk * U;
    max_rows = std::max(std::max(B, 256), lm_only ? 0 : B * T0);
    // bf16 and int8 engines keep K and V in bf16 (half the cache traffic of the decode step); fp32 keeps f32
    kv16 = e.model().flow_wtype == W_BF16 && std::getenv("POCKET_TTS_KV_F32") == nullptr ? 1 : 0;
    const size_t kvb = kv16 ? 2 : 4;
    lm_k.alloc(static_cast<size_t>(s.lm_layers) * B * lm_cap * D * kvb);
    lm_v.alloc(static_cast<size_t>(s.lm_layers) * B * lm_cap * D * kvb);
    lm_pos.alloc(static_cast<size_t>(B) * 4);
    lm_pos_host.assign(B, 0);
    if (!lm_only) {
        dec_k.alloc(static_cast<size_t>(s.mimi_layers) * B * dec_cap * M * kvb);
        dec_v.alloc(static_cast<size_t>(s.mimi_layers) * B * dec_cap * M * kvb);
    }
    dec_pos.alloc(static_cast<size_t>(B) * 4);
    dec_pos_host.assign(B, 0);
    const int dmax = std::max(D, M), fmax = std::max(s.lm_ffn, s.mimi_ffn);
    // transformer scratch: LM passes use at most max(B, 256) rows of width D, decoder passes B*T0 rows of width M
    size_t rows_lm = std::max(B, 256), rows_dec = lm_only ? 0 : static_cast<size_t>(B) * T0;
    size_t xe = std::max(rows_lm * D, rows_dec * M);
    x.alloc(xe * 4); h.alloc(xe * 4); q.alloc(xe * 4); att.alloc(xe * 4);
    qkv.alloc(xe * 3 * 4);
    scratch_rows_d = xe;
    scratch_rows_f = std::max(rows_lm * s.lm_ffn, rows_dec * s.mimi_ffn);
    ff.alloc(scratch_rows_f * 4);
    if (!lm_only) {
        dx.alloc(rows_dec * M * 4); dh.alloc(rows_dec * M * 4); dq.alloc(rows_dec * M * 4); datt.alloc(rows_dec * M * 4);
        dqkv.alloc(rows_dec * M * 3 * 4); dff.alloc(rows_dec * s.mimi_ffn * 4);
    }
    st_nv.alloc(static_cast<size_t>(B) * 4); st_nt.alloc(static_cast<size_t>(B) * 4); st_len.alloc(static_cast<size_t>(B) * 4);
    (void)dmax; (void)fmax;
    seq_in.alloc(static_cast<size_t>(B) * L * 4);
    cond.alloc(static_cast<size_t>(B) * D * 4);
    eos.alloc(static_cast<size_t>(B) * 4);
    x0.alloc(static_cast<size_t>(B) * L * 4);
    xcur.alloc(static_cast<size_t>(B) * L * 4);
    lat_frames = (o.max_frames + 2 * kDecChunk + kDecChunk - 1) / kDecChunk * kDecChunk;
    latents.alloc(static_cast<size_t>(B) * lat_frames * L * 4);
    ids.alloc(static_cast<size_t>(std::max(o.max_text_tokens, 1)) * 8);
    const int C = s.flow_dim;
    cemb.alloc(static_cast<size_t>(16) * B * C * 4);   // up to 16 K-slice slabs
    ymod.alloc(static_cast<size_t>(o.lsd_steps) * B * C * 4);
    mods.alloc(static_cast<size_t>(o.lsd_steps) * B * e.model().mod_width * 4);
    flow_ws.alloc(flow_head_workspace_bytes(B, C, e.model().flow_wtype, s.flow_blocks));
    flow_ctl.alloc(flow_ring_ctl_bytes(B, s.flow_blocks));
    sync_fill(flow_ctl.p, 0, flow_ctl.bytes);   // the error word is read after every run, whichever form of the head ran
    // decoder activations
    if (!lm_only) {
    zbuf.alloc(static_cast<size_t>(B) * kDecChunk * L * 4);
    qbuf.alloc(static_cast<size_t>(B) * (1 + kDecChunk) * M * 4);
    ybuf.alloc(static_cast<size_t>(B) * (s.kernel - 1 + T0) * M * 4);
    int c = 8 * s.n_filters;
    e0buf.alloc(static_cast<size_t>(B) * (1 + T0) * c * 4);
    int T = T0;
    for (int i = 0; i < 3; ++i) {
        int co = c / 2;
        T *= s.ratios[i];
        int hn = i < 2 ? 1 : s.last_kernel - 1;
        ubuf[i].alloc(static_cast<size_t>(B) * T * co * 4);
        uebuf[i].alloc(static_cast<size_t>(B) * (s.res_kernel - 1 + T) * co * 4);
        h1buf[i].alloc(static_cast<size_t>(B) * T * (co / 2) * 4);
        ebuf[i].alloc(static_cast<size_t>(B) * (hn + T) * co * 4);
        c = co;
    }
    audio.alloc(static_cast<size_t>(B) * kDecChunk * s.frame_size() * 4);
    }
    PTTS_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_eos), static_cast<size_t>(B) * 4));
    PTTS_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_x0), static_cast<size_t>(3) * B * L * 4));   // BOS rows + two sets of start points
    PTTS_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_eos_all), static_cast<size_t>(std::max(o.max_frames, 16)) * B * 4, hipHostMallocMapped));
    PTTS_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_flag), 64, hipHostMallocMapped));
    h_flag[0] = 0; h_flag[1] = 0;   // [0]: frames published; [1]: mirror of the persistent head's error word
    step_dev.alloc(64);
    partial.alloc(static_cast<size_t>(16) * B * D * 4);   // up to 16 K-slice slabs, sized once: the frame graph holds the pointer
    if (e.model().lm_wtype == W_I8 && D % 64 == 0 && s.lm_ffn % 64 == 0 && D == s.lm_heads * 64) {
        q_rows = (B + 15) & ~15;
        auto zeroed = [&](DevBuf& d, size_t bytes) { d.alloc(bytes); sync_fill(d.p, 0, bytes); };
        zeroed(hq, static_cast<size_t>(B) * D); zeroed(hd, static_cast<size_t>(D / 64) * q_rows * 4);
        zeroed(attq, static_cast<size_t>(B) * D); zeroed(attd, static_cast<size_t>(s.lm_heads) * q_rows * 4);
        zeroed(ffq, static_cast<size_t>(B) * s.lm_ffn); zeroed(ffd, static_cast<size_t>(s.lm_ffn / 64) * q_rows * 4);
        if (e.model().q8_act == 1) {
            zeroed(mm_h, static_cast<size_t>(B) * 8); zeroed(mm_att, static_cast<size_t>(s.lm_heads) * B * 8);
            zeroed(mm_ff, static_cast<size_t>(s.lm_ffn / 64) * B * 8);
        }
    }
    for (auto& pair : tev)
        for (auto& ev : pair) PTTS_HIP(hipEventCreate(&ev));
    reset(e.stream());
}

// Grow the transformer scratch so that a pass over `rows` rows of width d (FFN width ffn) fits.
// The captured frame graphs hold the addresses of x / h / qkv / att / ff (the decode step runs in them), so a
// reallocation drops the graphs; the next run captures them again on the new buffers.
void Batch::ensure_rows(size_t rows, int d, int ffn) {
    const size_t need_d = rows * d, need_f = rows * ffn;
    // (the batch's own streams are drained, not the device: a device-wide wait is refused while another engine's thread
    // is capturing its frame graphs)
    auto drain = [&] {
        if (st) PTTS_HIP(hipStreamSynchronize(st));
        if (dec_st && dec_st != st) PTTS_HIP(hipStreamSynchronize(dec_st));
    };
    if (need_d > scratch_rows_d) {
        drain();
        drop_graphs();
        x.alloc(need_d * 4); h.alloc(need_d * 4); q.alloc(need_d * 4); att.alloc(need_d * 4); qkv.alloc(need_d * 3 * 4);
        scratch_rows_d = need_d;
    }
    if (need_f > scratch_rows_f) {
        drain();
        drop_graphs();
        ff.alloc(need_f * 4);
        scratch_rows_f = need_f;
    }
}

void Batch::drop_graphs() {
    for (auto& kv : frame_graphs)
        for (auto& ex : kv.second.exec)
            if (ex) (void)hipGraphExecDestroy(ex);
    frame_graphs.clear();
}

Batch::~Batch() {
    drop_graphs();
    for (auto& pair : tev)
        for (auto& ev : pair)
            if (ev) (void)hipEventDestroy(ev);
    if (h_eos) (void)hipHostFree(h_eos);
    if (h_x0) (void)hipHostFree(h_x0);
    if (h_eos_all) (void)hipHostFree(h_eos_all);
    if (h_flag) (void)hipHostFree(h_flag);
}

// Zero every recurrent state: the reference zero-initialises all state_* inputs (src/pocket_tts.cpp:133-186).
void Batch::reset(hipStream_t s) {
    PTTS_HIP(hipMemsetAsync(lm_pos.p, 0, lm_pos.bytes, s));
    PTTS_HIP(hipMemsetAsync(dec_pos.p, 0, dec_pos.bytes, s));
    std::fill(lm_pos_host.begin(), lm_pos_host.end(), 0);
    std::fill(dec_pos_host.begin(), dec_pos_host.end(), 0);
    if (!lm_only)
        for (DevBuf* b : {&qbuf, &ybuf, &e0buf, &uebuf[0], &uebuf[1], &uebuf[2], &ebuf[0], &ebuf[1], &ebuf[2]})
            PTTS_HIP(hipMemsetAsync(b->p, 0, b->bytes, s));
    PTTS_HIP(hipMemsetAsync(x0.p, 0, x0.bytes, s));
}

// ------------------------------------------------------------------------------------------ Engine

namespace {
float silu_h(float v) { return v / (1.0f + std::exp(-v)); }
// One persistent flow head can be resident per GPU (192 of its 256 CUs at one workgroup each).  Of the engines a process
// creates on a device the first holds that slot until it is destroyed; the others run the head as a launch chain.
std::atomic<int> g_ring_owner[64];
}  // namespace

Engine::Engine(const EngineOptions& opt) : opt_(opt) {
    if (opt_.lsd_steps <= 0 || opt_.max_frames <= 0) throw std::runtime_error("engine: lsd_steps and max_frames must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        throw std::runtime_error("no HIP device available: the MI355X engine has no CPU fallback");
    if (opt_.device < 0 || opt_.device >= ndev) throw std::runtime_error("engine: device ordinal out of range");
    PTTS_HIP(hipSetDevice(opt_.device));
    PTTS_TRACE("ctor: device set");
    {
        int expect = 0;
        owns_ring_ = g_ring_owner[opt_.device & 63].compare_exchange_strong(expect, 1);
    }
    // Spatial split of the compute units: the decoder stream is confined to k of every 8 blocks of 8 CUs
    // (POCKET_TTS_DECODER_CU_EIGHTHS, default 4, 0 = no mask), so its wide kernels always leave CUs free for the
    // frame loop's small dependent launches.  Measured at B = 64 on MI355X: no mask 3350 RTFx, k = 4 3620, k = 3 3540,
    // k = 2 3350, k >= 5 no gain.  POCKET_TTS_FRAME_CU_EIGHTHS = k additionally confines the frame loop to the
    // other blocks instead of letting it use the whole chip (slower: the decode attention wants every CU).
    int dec_eighths = 4, frame_eighths = 0;
    if (const char* e = std::getenv("POCKET_TTS_DECODER_CU_EIGHTHS")) dec_eighths = std::atoi(e);
    if (const char* e = std::getenv("POCKET_TTS_FRAME_CU_EIGHTHS")) { frame_eighths = std::atoi(e); dec_eighths = 8 - frame_eighths; }
    int lo = 0, hi = 0;
    PTTS_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (dec_eighths >= 1 && dec_eighths <= 7) {
        hipDeviceProp_t prop;
        PTTS_HIP(hipGetDeviceProperties(&prop, opt_.device));
        const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
        std::vector<uint32_t> main_mask(words, 0u), dec_mask(words, 0u);
        for (int i = 0; i < ncu; ++i) {
            const bool to_dec = (i / 8) % 8 >= 8 - dec_eighths;
            (to_dec ? dec_mask : main_mask)[i / 32] |= 1u << (i % 32);
        }
        // The frame loop's stream has the default priority.  Measured with two engines on one GPU: two high-priority
        // streams ran their frames one after the other (4600 RTFx for both together against 6060 for one alone), two
        // default-priority streams side by side (7600); one engine alone runs the same either way.
        if (frame_eighths >= 1) PTTS_HIP(hipExtStreamCreateWithCUMask(&stream_, words, main_mask.data()));
        else PTTS_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        if (hipExtStreamCreateWithCUMask(&dec_stream_, words, dec_mask.data()) != hipSuccess) {
            (void)hipGetLastError();   // masks unsupported here: an ordinary low-priority stream does the job, only slower
            PTTS_HIP(hipStreamCreateWithPriority(&dec_stream_, hipStreamNonBlocking, lo));
        }
    } else {
        // the frame loop's small dependent launches go ahead of the decoder's wide ones whenever both are ready
        PTTS_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        PTTS_HIP(hipStreamCreateWithPriority(&dec_stream_, hipStreamNonBlocking, lo));
    }
    prefill_stream_ = shared_side_stream();   // one per device for the whole process: see pack.cpp
    PTTS_HIP(hipEventCreateWithFlags(&latents_ready_, hipEventDisableTiming));
    PTTS_HIP(hipEventCreateWithFlags(&dec_done_, hipEventDisableTiming));
    lanes_ = opt_.lanes;
    if (lanes_ <= 0) {
        const char* e = std::getenv("POCKET_TTS_LANES");
        lanes_ = e ? std::atoi(e) : 1;   // measured on MI355X: kernels of separate streams barely overlap at this launch size
    }
    lanes_ = std::max(1, std::min(lanes_, 8));
    use_graph_ = std::getenv("POCKET_TTS_NO_GRAPH") == nullptr;
    if (const char* e = std::getenv("POCKET_TTS_LANE_DECODER")) lane_shared_decoder_ = std::string(e) == "shared";
    lane_streams_.push_back(stream_);
    lane_events_.push_back(latents_ready_);
    for (int i = 1; i < lanes_; ++i) {
        hipStream_t ls;
        hipEvent_t le;
        PTTS_HIP(hipStreamCreateWithFlags(&ls, hipStreamNonBlocking));
        PTTS_HIP(hipEventCreateWithFlags(&le, hipEventDisableTiming));
        lane_streams_.push_back(ls);
        lane_events_.push_back(le);
    }
    PTTS_TRACE("ctor: streams made");
    Pack pack(opt_.pack_path);
    model_ = std::make_unique<Model>(pack, opt_.precision);
    PTTS_TRACE("ctor: model uploaded");
    const Spec& s = model_->spec();
    const int lm_len = 64 + opt_.max_text_tokens + opt_.max_frames + 8 + 64;
    const int mimi_len = std::max((opt_.max_frames + kDecChunk) * s.upsample + 64, 2048);
    model_->build_rope_tables(lm_len, mimi_len);
    if (opt_.has_seed) rng_.seed(static_cast<uint32_t>(opt_.seed));
    else rng_.seed(std::random_device{}());

    // averaged time embeddings of the fixed Euler grid s_j = j/N, t_j = s_j + 1/N (src/pocket_tts.cpp:122-130)
    const int C = s.flow_dim, Q = s.flow_freq, half = Q / 2, N = opt_.lsd_steps;
    std::vector<float> te(static_cast<size_t>(N) * C, 0.f), fe(Q), hid(C), out(C);
    const float dt = 1.0f / static_cast<float>(N);
    for (int j = 0; j < N; ++j) {
        const float sv = static_cast<float>(j) / static_cast<float>(N);
        const float tv = sv + dt;
        for (int which = 0; which < 2; ++which) {
            const float val = which == 0 ? sv : tv;
            std::string p = "flow.time" + std::to_string(which) + ".";
            const HostTensor &w1 = pack.get(p + "l1.w"), &b1 = pack.get(p + "l1.b"), &w2 = pack.get(p + "l2.w"), &b2 = pack.get(p + "l2.b");
            for (int i = 0; i < half; ++i) {
                double freq = std::exp(-std::log(static_cast<double>(s.flow_max_period)) * i / half);
                double a = static_cast<double>(val) * freq;
                fe[i] = static_cast<float>(std::cos(a));
                fe[half + i] = static_cast<float>(std::sin(a));
            }
            for (int n = 0; n < C; ++n) {
                float acc = 0.f;
                for (int k = 0; k < Q; ++k) acc += fe[k] * w1.data[static_cast<size_t>(n) * Q + k];
                hid[n] = silu_h(acc + b1.data[n]);
            }
            for (int n = 0; n < C; ++n) {
                float acc = 0.f;
                for (int k = 0; k < C; ++k) acc += hid[k] * w2.data[static_cast<size_t>(n) * C + k];
                out[n] = acc + b2.data[n];
            }
            for (int n = 0; n < C; ++n) te[static_cast<size_t>(j) * C + n] += out[n];
        }
        for (int n = 0; n < C; ++n) te[static_cast<size_t>(j) * C + n] *= 0.5f;
    }
    // resolve every per-frame device pointer once
    {
        const Model& m = *model_;
        hd_.lm = m.tf_layers("lm.l", s.lm_layers, false);
        hd_.dec = m.tf_layers("dec.t", s.mimi_layers, true);
        hd_.enc = m.tf_layers("enc.t", s.mimi_layers, true);
        hd_.lm_in = m.w("lm.in_proj.w"); hd_.flow_cond = m.w("flow.cond.w"); hd_.flow_in = m.w("flow.in_proj.w");
        hd_.flow_out = m.w("flow.final.lin.w"); hd_.dec_in = m.w("dec.in_proj.w"); hd_.dec_conv0 = m.w("dec.conv0.w");
        hd_.bos = m.f("lm.bos_emb"); hd_.out_g = m.f("lm.out_norm.g"); hd_.out_b = m.f("lm.out_norm.b");
        hd_.eos_w = m.f("lm.eos.w"); hd_.eos_b = m.f("lm.eos.b"); hd_.flow_cond_b = m.f("flow.cond.b");
        hd_.flow_in_b = m.f("flow.in_proj.b"); hd_.flow_out_b = m.f("flow.final.lin.b");
        hd_.emb_std = m.f("mimi.emb_std"); hd_.emb_mean = m.f("mimi.emb_mean"); hd_.dec_up_w = m.f("dec.up.w");
        hd_.dec_conv0_b = m.f("dec.conv0.b"); hd_.dec_final_w = m.f("dec.final.w"); hd_.dec_final_b = m.f("dec.final.b");
        hd_.enc_conv0_w = m.f("enc.conv0.w"); hd_.enc_conv0_b = m.f("enc.conv0.b");
        hd_.enc_final = m.w("enc.final.w"); hd_.enc_final_b = m.f("enc.final.b");
        hd_.enc_ds = m.w("enc.down.w"); hd_.enc_spk = m.w("enc.spk_proj.w"); hd_.text_embed = m.f("text.embed");
        for (int i = 0; i < 3; ++i) {
            std::string p = "dec.s" + std::to_string(i) + ".", q = "enc.s" + std::to_string(i) + ".";
            hd_.dec_up[i] = m.w(p + "up.w"); hd_.dec_c1[i] = m.w(p + "res.c1.w"); hd_.dec_c2[i] = m.w(p + "res.c2.w");
            hd_.dec_up_b[i] = m.f(p + "up.b"); hd_.dec_c1_b[i] = m.f(p + "res.c1.b"); hd_.dec_c2_b[i] = m.f(p + "res.c2.b");
            hd_.enc_c1[i] = m.w(q + "res.c1.w"); hd_.enc_c2[i] = m.w(q + "res.c2.w"); hd_.enc_down[i] = m.w(q + "down.w");
            hd_.enc_c1_b[i] = m.f(q + "res.c1.b"); hd_.enc_c2_b[i] = m.f(q + "res.c2.b"); hd_.enc_down_b[i] = m.f(q + "down.b");
        }
    }
    {
        int c = 8 * s.n_filters;
        for (int i = 0; i < 3; ++i) {
            c /= 2;
            dec_fused_[i] = !std::getenv("POCKET_TTS_NO_FUSED_RESBLOCK") && hd_.dec_c1[i].wtype == hd_.dec_c2[i].wtype &&
                            resblock_supported(c, s.res_kernel, hd_.dec_c1[i].wtype);
        }
    }
    PTTS_TRACE("ctor: before flow prepare");
    flow_head_prepare(s.flow_dim, model_->flow_wtype);
    init_static_scratch();
    PTTS_TRACE("ctor: static scratch resolved");
    flow_te_.alloc(te.size() * 4);
    sync_copy(flow_te_.p, te.data(), te.size() * 4, hipMemcpyHostToDevice);
    PTTS_HIP(hipEventCreate(&ev_[0]));
    PTTS_HIP(hipEventCreate(&ev_[1]));
    PTTS_TRACE("ctor: done");
}

Engine::~Engine() {
    PTTS_TRACE("dtor: begin");
    struct Release { bool own; int dev; ~Release() { if (own) g_ring_owner[dev & 63].store(0); } } release{owns_ring_ || ring_gave_up_, opt_.device};
    stop_worker();
    PTTS_TRACE("dtor: worker joined");
    (void)hipSetDevice(opt_.device);
    (void)hipStreamSynchronize(stream_);
    batch_pool_.clear();
    drop_table();
    PTTS_TRACE("dtor: table dropped");
    for (auto& E : enc_slot_) {
        if (E.pin_in) (void)hipHostFree(E.pin_in);
        if (E.pin_out) (void)hipHostFree(E.pin_out);
        if (E.done) (void)hipEventDestroy(E.done);
    }
    for (auto& pe : pending_events_) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
    if (ev_[0]) (void)hipEventDestroy(ev_[0]);
    if (ev_[1]) (void)hipEventDestroy(ev_[1]);
    for (size_t i = 1; i < lane_streams_.size(); ++i) {
        (void)hipStreamSynchronize(lane_streams_[i]);
        (void)hipStreamDestroy(lane_streams_[i]);
        (void)hipEventDestroy(lane_events_[i]);
    }
    if (latents_ready_) (void)hipEventDestroy(latents_ready_);
    if (dec_done_) (void)hipEventDestroy(dec_done_);
    if (dec_stream_) { (void)hipStreamSynchronize(dec_stream_); (void)hipStreamDestroy(dec_stream_); }
    if (prefill_stream_) (void)hipStreamSynchronize(prefill_stream_);   // shared, not this engine's to destroy
    if (stream_) (void)hipStreamDestroy(stream_);
    PTTS_TRACE("dtor: streams destroyed");
}

void Engine::enable_phase_timing(bool on) { timing_ = on; times_ = PhaseTimes(); }

void Engine::phase_begin(int which, hipStream_t st) {
    if (!timing_) return;
    if (!st) st = stream_;
    PhaseEvent pe;
    pe.which = which; pe.st = st;
    PTTS_HIP(hipEventCreate(&pe.a));
    PTTS_HIP(hipEventCreate(&pe.b));
    PTTS_HIP(hipEventRecord(pe.a, st));
    std::lock_guard<std::mutex> g(ev_mu_);
    pending_events_.push_back(pe);
}

void Engine::phase_end(int which, hipStream_t st) {
    if (!timing_) return;
    if (!st) st = stream_;
    std::lock_guard<std::mutex> g(ev_mu_);
    for (auto it = pending_events_.rbegin(); it != pending_events_.rend(); ++it)
        if (it->which == which && it->st == st && !it->closed) {
            PTTS_HIP(hipEventRecord(it->b, st));
            it->closed = true;
            return;
        }
}

Engine::PhaseTimes Engine::phase_times() {
    for (hipStream_t ls : lane_streams_) PTTS_HIP(hipStreamSynchronize(ls));
    PTTS_HIP(hipStreamSynchronize(dec_stream_));
    std::lock_guard<std::mutex> g(ev_mu_);
    for (auto& pe : pending_events_) {
        float ms = 0.f;
        if (pe.closed && hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
            if (pe.which == 0) { times_.lm_ms += ms; times_.lm_calls++; }
            else if (pe.which == 1) { times_.flow_ms += ms; times_.flow_calls++; }
            else if (pe.which == 2) { times_.dec_ms += ms; times_.dec_calls++; }
            else { times_.head_ms += ms; times_.head_calls++; }
        }
        (void)hipEventDestroy(pe.a);
        (void)hipEventDestroy(pe.b);
    }
    pending_events_.clear();
    return times_;
}

Batch* Engine::batch_new(int B) {
    PTTS_HIP(hipSetDevice(opt_.device));
    return new Batch(*this, B);
}
void Engine::batch_free(Batch* b) { delete b; }
void Engine::batch_reset(Batch* b) { b->reset(stream_); }

Batch* Engine::acquire_batch(int B, int lane) {
    auto key = std::make_pair(B, lane);
    auto it = batch_pool_.find(key);
    const bool fresh = it == batch_pool_.end();
    if (fresh) it = batch_pool_.emplace(key, std::make_unique<Batch>(*this, B)).first;
    Batch* b = it->second.get();
    // most recently used first; beyond kMaxPooled sizes the least recently used one that the staged job does not hold is freed
    batch_lru_.remove(key);
    batch_lru_.push_front(key);
    for (auto old = batch_lru_.end(); batch_pool_.size() > kMaxPooled && old != batch_lru_.begin();) {
        --old;
        auto victim = batch_pool_.find(*old);
        const bool held = victim == it || std::find(job_.parts.begin(), job_.parts.end(), victim->second.get()) != job_.parts.end();
        if (held) continue;
        (void)hipStreamSynchronize(victim->second->st ? victim->second->st : stream_);
        batch_pool_.erase(victim);
        old = batch_lru_.erase(old);
    }
    b->st = lane_streams_[lane];
    b->ready = lane_events_[lane];
    b->dec_st = dec_stream_;
    if (!fresh) b->reset(b->st);
    return b;
}

// One pre-norm transformer stack over M = nb * rpb new rows X[M][d] (in place), appending to the
// per-slot KV cache.  If last_out is given, the final residual sum is written there (batched-row
// mapping) instead of X, which is how the decoder hands its result to the first SEANet convolution's
// history buffer without a copy.
void Engine::transformer(const std::vector<TfLayer>& layers, void* kc, void* vc, int kv16, long long layer_stride, int cap,
                         const int* pos_dev, int slot0, int nb, int rpb, float* X, int d, int heads, int ffn, float eps,
                         int window, const float* cos_tab, const float* sin_tab, int tab_len, const TfScratch& b, float* last_out,
                         long long last_bs, int last_ld, hipStream_t st, bool split_a, bool bf16_rows) {
    const int M = nb * rpb;
    // One utterance's short pass (a streaming chunk of the decoder): the tiled GEMM would run 16 to 64 workgroups
    // through K loops of up to 2048; the skinny kernels cut the same products into 16-row x 64-column workgroups.
    bool thin = nb == 1 && M <= 256 && !std::getenv("POCKET_TTS_NO_THIN_PASS");
    for (auto& w : layers) thin = thin && w.qkv.frag && w.out.frag && w.ff1.frag && w.ff2.frag;
    auto thin_linear = [&](const DevWeight& w, const float* A, int lda, float* C, int ldc, int N, int K, int act, const float* colscale,
                           const float* R, int ldr) {
        SkinnyArgs k;
        k.A = A; k.lda = lda; k.Wp = w.frag; k.wtype = w.wtype; k.M = M; k.N = N; k.K = K; k.act = act;
        k.C = C; k.ldc = ldc; k.colscale = colscale; k.R = R; k.ldr = ldr;
        launch_skinny_gemm(k, st);
    };
    // bf16 rows between the kernels where the consumer rounds to bf16 once anyway (the normalised rows and the FFN's
    // hidden rows): the same values, half the bytes
    bool bfr = bf16_rows && !thin && !split_a && M >= 512 && (d & 7) == 0 && (ffn & 7) == 0;
    for (auto& w : layers) bfr = bfr && w.qkv.wtype == W_BF16 && w.ff1.wtype == W_BF16 && w.ff2.wtype == W_BF16;
    for (size_t l = 0; l < layers.size(); ++l) {
        const TfLayer& w = layers[l];
        LayerNormArgs ln;
        ln.X = X; ln.ldx = d; ln.Y = b.h; ln.ldy = d; ln.M = M; ln.D = d; ln.g = w.ln1g; ln.b = w.ln1b; ln.eps = eps;
        ln.y_bf16 = bfr;
        launch_layernorm(ln, st);
        GemmArgs g;
        g.A = b.h; g.lda = d; g.W = w.qkv.p; g.wtype = w.qkv.wtype; g.C = b.qkv; g.ldc = 3 * d; g.M = M; g.N = 3 * d; g.K = d;
        g.split_a = split_a; g.a_bf16 = bfr;
        if (w.qkv.lo) w.qkv.bind(g, M);
        if (thin) thin_linear(w.qkv, b.h, d, b.qkv, 3 * d, 3 * d, d, ACT_NONE, nullptr, nullptr, 0);
        else launch_gemm(g, st);
        RopeAppendArgs r;
        const long long kv_off = static_cast<long long>(l) * layer_stride * (kv16 ? 2 : 4);
        r.qkv = b.qkv; r.q_out = b.q; r.kcache = static_cast<char*>(kc) + kv_off; r.vcache = static_cast<char*>(vc) + kv_off; r.kv16 = kv16;
        r.pos = pos_dev; r.slot0 = slot0; r.rpb = rpb; r.M = M; r.d = d; r.heads = heads; r.cap = cap;
        r.cos_tab = cos_tab; r.sin_tab = sin_tab; r.tab_len = tab_len;
        launch_rope_append(r, st);
        AttentionArgs at;
        at.q = b.q; at.out = b.att; at.kcache = r.kcache; at.vcache = r.vcache; at.kv16 = kv16; at.pos = pos_dev; at.slot0 = slot0; at.rpb = rpb;
        at.M = M; at.d = d; at.heads = heads; at.cap = cap; at.window = window;
        launch_attention(at, st);
        GemmArgs o;
        o.A = b.att; o.lda = d; o.W = w.out.p; o.wtype = w.out.wtype; o.C = X; o.ldc = d; o.R = X; o.ldr = d; o.colscale = w.ls1;
        o.M = M; o.N = d; o.K = d; o.split_a = split_a;
        if (w.out.lo) w.out.bind(o, M);
        if (thin) thin_linear(w.out, b.att, d, X, d, d, d, ACT_NONE, w.ls1, X, d);
        else launch_gemm(o, st);
        ln.g = w.ln2g; ln.b = w.ln2b;
        launch_layernorm(ln, st);
        GemmArgs f1;
        f1.A = b.h; f1.lda = d; f1.W = w.ff1.p; f1.wtype = w.ff1.wtype; f1.C = b.ff; f1.ldc = ffn; f1.act_pre = ACT_GELU;
        f1.M = M; f1.N = ffn; f1.K = d; f1.split_a = split_a; f1.a_bf16 = bfr; f1.c_bf16 = bfr;
        if (w.ff1.lo) w.ff1.bind(f1, M);
        if (thin) thin_linear(w.ff1, b.h, d, b.ff, ffn, ffn, d, ACT_GELU, nullptr, nullptr, 0);
        else launch_gemm(f1, st);
        GemmArgs f2;
        f2.A = b.ff; f2.lda = ffn; f2.W = w.ff2.p; f2.wtype = w.ff2.wtype; f2.R = X; f2.ldr = d; f2.colscale = w.ls2;
        f2.M = M; f2.N = d; f2.K = ffn; f2.split_a = split_a; f2.a_bf16 = bfr;
        if (w.ff2.lo) w.ff2.bind(f2, M);
        if (last_out && l + 1 == layers.size()) {
            f2.C = last_out; f2.c_bs = last_bs; f2.ldc = last_ld; f2.rpb = rpb;
            f2.r_bs = (long long)rpb * d; f2.a_bs = (long long)rpb * ffn;
        }
        else { f2.C = X; f2.ldc = d; }
        if (thin) thin_linear(w.ff2, b.ff, ffn, f2.C, f2.ldc, d, ffn, ACT_NONE, w.ls2, X, d);   // one utterance: no row remapping
        else launch_gemm(f2, st);
    }
    check_launches("the transformer pass");
}

// The dynamic uint8 scheme of an int8 product: weight row sums for the product's slice count.
static void bind_q8_act(SkinnyArgs& k, int scheme, const int* const* wsums) {
    if (scheme != 1 || k.wtype != W_I8) return;
    if (k.splits < 1 || k.splits > 16 || !wsums[k.splits]) throw std::runtime_error("int8 product: no weight row sums for this K-slice count");
    k.q8_act = 1;
    k.wsum = wsums[k.splits];
}

// K-slices for a split-K product: about 256 deep each, at most 8, dividing K into k-step multiples.
static int pick_splits(int K, int kstep) {
    int s = K / 256;
    if (s < 1) s = 1;
    if (s > 8) s = 8;
    while (s > 1 && (K % s != 0 || (K / s) % kstep != 0)) --s;
    return s;
}

// adv_dev / adv_host (optional): per-slot position advance when the nb x rpb rows are padded to a common length.
void Engine::lm_rows(Batch& b, int slot0, int nb, int rpb, float* X, const int* adv_dev, const int* adv_host) {
    const Spec& s = spec();
    for (int i = 0; i < nb; ++i)
        if (b.lm_pos_host[slot0 + i] + rpb > b.lm_cap)
            throw std::runtime_error("LM context exceeds the KV-cache capacity (voice + text + frames too long)");
    b.ensure_rows(static_cast<size_t>(nb) * rpb, s.d_model, s.lm_ffn);
    // int8 matrices exist only in the skinny kernels' layout; float and bf16 passes of up to 128 rows (one utterance's
    // voice and text) go the same way, because a tiled GEMM of that height fills 16 to 48 of the 256 CUs
    if (model_->lm_wtype == W_I8 || (nb * rpb <= 128 && !std::getenv("POCKET_TTS_NO_SKINNY_PREFILL"))) {
        lm_rows_skinny(b, slot0, nb, rpb, X);
        if (adv_dev) launch_advance_pos_vec(b.lm_pos.as<int>() + slot0, adv_dev, nb, b.st);
        else launch_advance_pos(b.lm_pos.as<int>() + slot0, nb, rpb, b.st);
        for (int i = 0; i < nb; ++i) b.lm_pos_host[slot0 + i] += adv_host ? adv_host[i] : rpb;
        return;
    }
    const auto& layers = hd_.lm;
    const long long layer_stride = static_cast<long long>(b.B) * b.lm_cap * s.d_model;
    transformer(layers, b.lm_k.p, b.lm_v.p, b.kv16, layer_stride, b.lm_cap, b.lm_pos.as<int>() + slot0, slot0, nb, rpb, X, s.d_model,
                s.lm_heads, s.lm_ffn, s.lm_eps, 0, model_->lm_cos, model_->lm_sin, model_->lm_tab_len, b.scratch(), nullptr, 0, 0, b.st, true);
    if (adv_dev) launch_advance_pos_vec(b.lm_pos.as<int>() + slot0, adv_dev, nb, b.st);
    else launch_advance_pos(b.lm_pos.as<int>() + slot0, nb, rpb, b.st);
    for (int i = 0; i < nb; ++i) b.lm_pos_host[slot0 + i] += adv_host ? adv_host[i] : rpb;
}

// Multi-row LM pass on the decode step's kernels (skinny product, add & norm) over M = nb * rpb rows, with the
// multi-row rotate/append and tiled attention in place of the one-row kernel.  Short passes cut the output and
// FFN-2 products into K-slices so that the launch has enough workgroups; the add & norm that follows sums them.
void Engine::lm_rows_skinny(Batch& b, int slot0, int nb, int rpb, float* X) {
    const Spec& s = spec();
    const int D = s.d_model, F = s.lm_ffn, M = nb * rpb;
    const int wt = model_->lm_wtype, kstep = wt == W_I8 ? 64 : wt == W_BF16 ? 32 : 16;
    const int s_out = M <= 128 ? pick_splits(D, kstep) : 1, s_ff2 = M <= 128 ? pick_splits(F, kstep) : 1;
    hipStream_t st = b.st;
    const size_t need = static_cast<size_t>(std::max(s_out, s_ff2)) * M * D * 4;
    if (b.pre_partial.bytes < need) {
        PTTS_HIP(hipStreamSynchronize(st));
        b.pre_partial.alloc(need);
    }
    const long long layer_stride = static_cast<long long>(b.B) * b.lm_cap * D;
    AddNormArgs an;
    an.X = X; an.ldx = D; an.M = M; an.D = D; an.eps = s.lm_eps;
    an.g = hd_.lm[0].ln1g; an.b = hd_.lm[0].ln1b; an.H = b.h.f();
    launch_add_norm(an, st);
    for (int l = 0; l < s.lm_layers; ++l) {
        const TfLayer& w = hd_.lm[l];
        const Model::PackedLayer& pw = model_->lm_packed[l];
        SkinnyArgs q;
        q.A = b.h.f(); q.lda = D; q.Wp = pw.qkv; q.Wp2 = pw.qkv_lo; q.wtype = wt; q.wscale = pw.s_qkv; q.M = M; q.N = 3 * D; q.K = D; q.C = b.qkv.f(); q.ldc = 3 * D;
        bind_q8_act(q, model_->q8_act, pw.ws_qkv);
        launch_skinny_gemm(q, st);
        RopeAppendArgs r;
        const long long kv_off = static_cast<long long>(l) * layer_stride * (b.kv16 ? 2 : 4);
        r.qkv = b.qkv.f(); r.q_out = b.q.f(); r.kcache = static_cast<char*>(b.lm_k.p) + kv_off; r.vcache = static_cast<char*>(b.lm_v.p) + kv_off; r.kv16 = b.kv16;
        r.pos = b.lm_pos.as<int>() + slot0; r.slot0 = slot0; r.rpb = rpb; r.M = M; r.d = D; r.heads = s.lm_heads; r.cap = b.lm_cap;
        r.cos_tab = model_->lm_cos; r.sin_tab = model_->lm_sin; r.tab_len = model_->lm_tab_len;
        launch_rope_append(r, st);
        AttentionArgs at;
        at.q = b.q.f(); at.out = b.att.f(); at.kcache = r.kcache; at.vcache = r.vcache; at.kv16 = b.kv16; at.pos = r.pos; at.slot0 = slot0; at.rpb = rpb;
        at.M = M; at.d = D; at.heads = s.lm_heads; at.cap = b.lm_cap; at.window = 0;
        launch_attention(at, st);
        SkinnyArgs o;
        o.A = b.att.f(); o.lda = D; o.Wp = pw.out; o.Wp2 = pw.out_lo; o.wtype = wt; o.wscale = pw.s_out; o.M = M; o.N = D; o.K = D; o.splits = s_out; o.P = b.pre_partial.f();
        bind_q8_act(o, model_->q8_act, pw.ws_out);
        launch_skinny_gemm(o, st);
        AddNormArgs a1 = an;
        a1.P = b.pre_partial.f(); a1.splits = s_out; a1.colscale = w.ls1; a1.g = w.ln2g; a1.b = w.ln2b;
        launch_add_norm(a1, st);
        SkinnyArgs f1;
        f1.A = b.h.f(); f1.lda = D; f1.Wp = pw.ff1; f1.wtype = wt; f1.wscale = pw.s_ff1; f1.M = M; f1.N = F; f1.K = D; f1.act = ACT_GELU;
        f1.C = b.ff.f(); f1.ldc = F;
        bind_q8_act(f1, model_->q8_act, pw.ws_ff1);
        launch_skinny_gemm(f1, st);
        SkinnyArgs f2;
        f2.A = b.ff.f(); f2.lda = F; f2.Wp = pw.ff2; f2.Wp2 = pw.ff2_lo; f2.wtype = wt; f2.wscale = pw.s_ff2; f2.M = M; f2.N = D; f2.K = F; f2.splits = s_ff2; f2.P = b.pre_partial.f();
        bind_q8_act(f2, model_->q8_act, pw.ws_ff2);
        launch_skinny_gemm(f2, st);
        AddNormArgs a2 = an;
        a2.P = b.pre_partial.f(); a2.splits = s_ff2; a2.colscale = w.ls2;
        if (l + 1 < s.lm_layers) { a2.g = hd_.lm[l + 1].ln1g; a2.b = hd_.lm[l + 1].ln1b; }
        else { a2.g = nullptr; a2.b = nullptr; a2.H = nullptr; }
        launch_add_norm(a2, st);
    }
    check_launches("the short prefill pass");
}

// sequence[B][latent] (NaN = BOS) -> conditioning[B][d_model], eos[B]   (src/pocket_tts.cpp:323-411, decode shape)
// One new row per utterance.  Per layer: QKV product -> rotate/append/attend -> output product as K-slices ->
// add & norm -> FFN-1 product with GELU -> FFN-2 product as K-slices -> add & norm (which also produces the
// next layer's normalised input, or after the last layer the conditioning vector and the EOS logit).
void Engine::lm_step_dev(Batch& b, const float* seq_dev, float* cond_dev, float* eos_dev, bool prepped) {
    const Spec& s = spec();
    const int B = b.Bw, D = s.d_model, F = s.lm_ffn;   // rows launched; b.B stays the slot stride of the caches
    const int wt = model_->lm_wtype, kstep = wt == W_I8 ? 64 : wt == W_BF16 ? 32 : 16;
    const float* seq_rows = seq_dev;
    if (!prepped) {
        launch_prep_latent(seq_dev, b.seq_in.f(), B, s.latent, 0, hd_.bos, nullptr, b.st);
        seq_rows = b.seq_in.f();
    }
    // the input projection (K = latent_dim, no bias) runs inside the first add & norm launch when its shape allows
    const DevWeight& win = hd_.lm_in;
    const bool proj_in_norm = s.latent % 8 == 0 && D % 4 == 0 && D <= 1024 && (win.wtype == W_BF16 || win.wtype == W_F32);
    if (!proj_in_norm) {
        GemmArgs g;
        g.A = seq_rows; g.lda = s.latent; g.W = win.p; g.wtype = win.wtype; g.C = b.x.f(); g.ldc = D;
        g.M = B; g.N = D; g.K = s.latent; g.split_a = 1;
        launch_gemm(g, b.st);
    }
    const int s_out = pick_splits(D, kstep), s_ff2 = pick_splits(F, kstep);
    int s_qkv = 1;
    if (const char* e = std::getenv("POCKET_TTS_QKV_SPLITS")) s_qkv = std::atoi(e);
    else if (B <= 64 && D % (4 * kstep) == 0) s_qkv = 4;   // measured at B = 64: 1 -> 0.568 ms per step, 2 -> 0.528, 4 -> 0.518
    if (s_qkv < 1 || D % s_qkv || (D / s_qkv) % kstep) s_qkv = 1;
    if (b.qkv.bytes < static_cast<size_t>(s_qkv) * B * 3 * D * 4) throw std::runtime_error("QKV slab buffer too small");
    if (b.partial.bytes < static_cast<size_t>(std::max(s_out, s_ff2)) * B * D * 4) throw std::runtime_error("split-K slab buffer too small");
    // int8 engine: every matrix input of the step is quantised by the kernel that produces it (add & norm, attention,
    // the FFN-1 epilogue), so the products read int8 rows and stage nothing
    // (block scheme only: in the dynamic uint8 scheme the products quantise their rows themselves)
    const bool preq = wt == W_I8 && model_->q8_act == 0 && b.q_rows > 0 && !std::getenv("POCKET_TTS_Q8_STAGED");
    // dynamic uint8 scheme: the producers leave the (min, max) of the rows they write, so that the products need not scan
    // their rows over the full K before quantising them (POCKET_TTS_Q8_STAGED=1: the products scan)
    const bool dyn_mm = wt == W_I8 && model_->q8_act == 1 && b.mm_h.p && !std::getenv("POCKET_TTS_Q8_STAGED");
    AddNormArgs an;
    an.X = b.x.f(); an.ldx = D; an.M = B; an.D = D; an.eps = s.lm_eps;
    an.g = hd_.lm[0].ln1g; an.b = hd_.lm[0].ln1b; an.H = b.h.f();
    if (dyn_mm) an.H_mm = b.mm_h.f();
    if (preq) { an.H = nullptr; an.Hq = b.hq.as<signed char>(); an.Hd = b.hd.f(); an.Mq = b.q_rows; }
    {
        AddNormArgs a0 = an;
        if (proj_in_norm) { a0.seq = seq_rows; a0.w_in = win.p; a0.w_in_type = win.wtype; a0.L = s.latent; }
        launch_add_norm(a0, b.st);
    }
    const long long layer_stride = static_cast<long long>(b.B) * b.lm_cap * D;
    for (int l = 0; l < s.lm_layers; ++l) {
        const TfLayer& w = hd_.lm[l];
        const Model::PackedLayer& pw = model_->lm_packed[l];
        SkinnyArgs q;
        q.A = b.h.f(); q.lda = D; q.Wp = pw.qkv; q.Wp2 = pw.qkv_lo; q.wtype = wt; q.wscale = pw.s_qkv; q.M = B; q.N = 3 * D; q.K = D;
        q.splits = s_qkv; q.P = b.qkv.f();   // K-slice slabs, summed where the attention kernel reads them
        if (preq) { q.A = nullptr; q.Aq = b.hq.as<signed char>(); q.ldaq = D; q.Ad = b.hd.f(); q.Mq = b.q_rows; }
        bind_q8_act(q, model_->q8_act, pw.ws_qkv);
        if (dyn_mm) { q.a_mm = b.mm_h.f(); q.mm_parts = 1; }
        launch_skinny_gemm(q, b.st);
        AttnDecodeArgs at;
        const long long kv_off = static_cast<long long>(l) * layer_stride * (b.kv16 ? 2 : 4);
        at.qkv = b.qkv.f(); at.qkv_splits = s_qkv; at.kcache = static_cast<char*>(b.lm_k.p) + kv_off; at.vcache = static_cast<char*>(b.lm_v.p) + kv_off; at.kv16 = b.kv16;
        at.pos = b.lm_pos.as<int>(); at.M = B; at.d = D; at.heads = s.lm_heads; at.cap = b.lm_cap; at.window = 0;
        at.cos_tab = model_->lm_cos; at.sin_tab = model_->lm_sin; at.tab_len = model_->lm_tab_len;
        at.out = b.att.f();
        if (preq) { at.out = nullptr; at.out_q = b.attq.as<signed char>(); at.out_d = b.attd.f(); at.Mq = b.q_rows; }
        if (dyn_mm) at.out_mm = b.mm_att.f();
        launch_attn_decode(at, b.st);
        SkinnyArgs o;
        o.A = b.att.f(); o.lda = D; o.Wp = pw.out; o.Wp2 = pw.out_lo; o.wtype = wt; o.wscale = pw.s_out; o.M = B; o.N = D; o.K = D; o.splits = s_out; o.P = b.partial.f();
        if (preq) { o.A = nullptr; o.Aq = b.attq.as<signed char>(); o.ldaq = D; o.Ad = b.attd.f(); o.Mq = b.q_rows; }
        bind_q8_act(o, model_->q8_act, pw.ws_out);
        if (dyn_mm) { o.a_mm = b.mm_att.f(); o.mm_parts = s.lm_heads; }
        launch_skinny_gemm(o, b.st);
        AddNormArgs a1 = an;
        a1.P = b.partial.f(); a1.splits = s_out; a1.colscale = w.ls1; a1.g = w.ln2g; a1.b = w.ln2b;
        launch_add_norm(a1, b.st);
        SkinnyArgs f1;
        f1.A = b.h.f(); f1.lda = D; f1.Wp = pw.ff1; f1.wtype = wt; f1.wscale = pw.s_ff1; f1.M = B; f1.N = F; f1.K = D; f1.act = ACT_GELU;
        f1.C = b.ff.f(); f1.ldc = F;
        if (preq) {
            f1.A = nullptr; f1.Aq = b.hq.as<signed char>(); f1.ldaq = D; f1.Ad = b.hd.f(); f1.Mq = b.q_rows;
            f1.C = nullptr; f1.Cq = b.ffq.as<signed char>(); f1.Cd = b.ffd.f();
        }
        bind_q8_act(f1, model_->q8_act, pw.ws_ff1);
        if (dyn_mm) { f1.a_mm = b.mm_h.f(); f1.mm_parts = 1; f1.c_mm = b.mm_ff.f(); }
        launch_skinny_gemm(f1, b.st);
        SkinnyArgs f2;
        f2.A = b.ff.f(); f2.lda = F; f2.Wp = pw.ff2; f2.Wp2 = pw.ff2_lo; f2.wtype = wt; f2.wscale = pw.s_ff2; f2.M = B; f2.N = D; f2.K = F; f2.splits = s_ff2; f2.P = b.partial.f();
        if (preq) { f2.A = nullptr; f2.Aq = b.ffq.as<signed char>(); f2.ldaq = F; f2.Ad = b.ffd.f(); f2.Mq = b.q_rows; }
        bind_q8_act(f2, model_->q8_act, pw.ws_ff2);
        if (dyn_mm) { f2.a_mm = b.mm_ff.f(); f2.mm_parts = F / 64; }
        launch_skinny_gemm(f2, b.st);
        AddNormArgs a2 = an;
        a2.P = b.partial.f(); a2.splits = s_ff2; a2.colscale = w.ls2;
        if (l + 1 < s.lm_layers) {
            a2.g = hd_.lm[l + 1].ln1g; a2.b = hd_.lm[l + 1].ln1b;
        } else {
            a2.g = hd_.out_g; a2.b = hd_.out_b; a2.H32 = cond_dev; a2.Hq = nullptr; a2.H = preq ? nullptr : a2.H; a2.H_mm = nullptr;
            a2.dot_w = hd_.eos_w; a2.dot_b = hd_.eos_b; a2.dot_out = eos_dev;
            a2.pos = b.lm_pos.as<int>(); a2.pos_delta = 1;
        }
        launch_add_norm(a2, b.st);
    }
    check_launches("the LM frame step");
}

// Host mirror of the positions one decode step consumes; call once per step actually run.
void Engine::lm_bookkeeping(Batch& b) {
    for (int i = 0; i < b.Bw; ++i) {
        if (b.lm_pos_host[i] + 1 > b.lm_cap) throw std::runtime_error("LM context exceeds the KV-cache capacity (voice + text + frames too long)");
        b.lm_pos_host[i] += 1;
    }
}

// lsd_steps Euler iterations for B rows: 3 small launches for the adaLN vectors, then the head's per-block chain.
void Engine::flow_dev(Batch& b, const float* cond_dev, const float* x0_dev, long long x0_stride, float* x_out_dev, int B,
                      const int* x0_step, long long x0_step_stride, hipEvent_t probe_a, hipEvent_t probe_b, bool frame_mode) {
    FlowScratch fs{b.cemb.f(), b.ymod.f(), b.mods.f(), b.flow_ws.f(), b.flow_ctl.as<unsigned>()};
    flow_core(fs, b.st, cond_dev, x0_dev, x0_stride, x_out_dev, B, x0_step, x0_step_stride, probe_a, probe_b, frame_mode ? &b : nullptr);
}

// Bytes of the five scratch areas of a flow phase over `rows` rows, each rounded up to 256 bytes:
// conditioning slabs, adaLN input rows, adaLN vectors, head workspace, the persistent head's control words.
void Engine::flow_scratch_layout(int rows, size_t off[6]) const {
    const Spec& s = spec();
    const int C = s.flow_dim, N = opt_.lsd_steps;
    auto r256 = [](size_t n) { return (n + 255) & ~static_cast<size_t>(255); };
    off[0] = 0;
    off[1] = off[0] + r256(static_cast<size_t>(16) * rows * C * 4);
    off[2] = off[1] + r256(static_cast<size_t>(N) * rows * C * 4);
    off[3] = off[2] + r256(static_cast<size_t>(N) * rows * model_->mod_width * 4);
    off[4] = off[3] + r256(flow_head_workspace_bytes(rows, C, model_->flow_wtype, s.flow_blocks));
    off[5] = off[4] + r256(flow_ring_ctl_bytes(rows, s.flow_blocks));
}

void Engine::check_flow_ring(const unsigned* ctl, int rows) {
    if (!ctl || !(model_->flow_wtype == W_BF16 || model_->flow_ring_split) || spec().flow_dim != 512) return;
    unsigned err = 0;
    sync_copy(&err, ctl + flow_ring_err_word(rows, spec().flow_blocks), 4, hipMemcpyDeviceToHost);
    if (err) {
        sync_fill(const_cast<unsigned*>(ctl), 0, 4);
        ring_failed();
        throw std::runtime_error("flow head: a hand-off between workgroups timed out (the persistent launch was not fully resident); "
                                 "this engine runs the head as a launch chain from now on");
    }
}

// After a failed hand-off the engine stops using the persistent head: every later call runs the launch chain.  The captured
// frame graphs of pooled batches hold the persistent launch and are dropped (the streams are idle: the caller has drained them).
void Engine::ring_failed() {
    owns_ring_ = false;
    ring_gave_up_ = true;
    for (auto& kv : batch_pool_) kv.second->drop_graphs();
}

// frame (optional): the batch whose frame bookkeeping (EOS publication, frame counter, latent filing) rides in the launches
void Engine::flow_core(const FlowScratch& fs, hipStream_t st, const float* cond_dev, const float* x0_dev, long long x0_stride,
                       float* x_out_dev, int B, const int* x0_step, long long x0_step_stride, hipEvent_t probe_a, hipEvent_t probe_b,
                       Batch* frame) {
    const Spec& s = spec();
    const int C = s.flow_dim, N = opt_.lsd_steps;
    // conditioning embedding as K-slice slabs (skinny product), summed inside the next kernel
    const int wt = model_->flow_wtype;
    const int cs = pick_splits(s.d_model, wt == W_BF16 ? 32 : 16);
    SkinnyArgs g;
    g.A = cond_dev; g.lda = s.d_model; g.Wp = model_->flow_cond_packed; g.Wp2 = model_->flow_cond_packed_lo; g.wtype = wt; g.M = B; g.N = C; g.K = s.d_model;
    g.splits = cs; g.P = fs.cemb;
    launch_skinny_gemm(g, st);
    FramePublish pub;
    if (frame) {
        pub.eos = frame->eos.f(); pub.step = frame->step_dev.as<int>(); pub.h_eos_all = frame->h_eos_all; pub.h_flag = frame->h_flag;
        pub.ring = frame->eos_ring; pub.stride = frame->B;
    }
    // the persistent head's counters start every call at zero; this launch (two boundaries ahead of the head) clears them
    FlowHeadArgs probe;
    probe.C = C; probe.L = s.latent; probe.blocks = s.flow_blocks; probe.steps = N; probe.B = B; probe.wtype = model_->flow_wtype;
    probe.host_tabs = &model_->flow_tabs;
    probe.block_wtype = model_->flow_block_wtype; probe.ring_split = model_->flow_ring_split; probe.ring_ctl = fs.ctl; probe.ring_wg_budget = 224 / std::max(lanes_, 1);
    const bool ring = owns_ring_ && fs.ctl && flow_ring_supported(probe);
    const int cw = static_cast<int>(flow_ring_counter_word());
    if (frame && ring) {   // h_flag[1] mirrors the head's error word (see FramePublish)
        pub.ring_err = fs.ctl + flow_ring_err_word(B, s.flow_blocks);
        pub.h_ring_err = reinterpret_cast<unsigned*>(frame->h_flag + 1);
    }
    // the adaLN product: where its rows-stationary kernel applies (640 x 10240 x 512 at 64 rows), the kernel in front of it
    // leaves the rows as two-term bf16 fragments in the same buffer and the product loads them as they are
    // (POCKET_TTS_ADA_F32_ROWS=1: f32 rows, converted and reordered by the product)
    SkinnyArgs m;
    if (model_->ada_packed) {
        m.A = fs.ymod; m.lda = C; m.Wp = model_->ada_packed; m.Wp2 = model_->ada_packed_lo; m.wtype = wt; m.M = N * B; m.N = model_->mod_width; m.K = C;
        m.bias = model_->ada_all_b; m.C = fs.mods; m.ldc = model_->mod_width;
        if (m.M > 128) { m.mtiles = 2; m.kchunk = 128; }   // measured at 640 x 10240 x 512: 32 us (64-row groups 39 us, tiled GEMM 66 us)
        const char* f32_rows = std::getenv("POCKET_TTS_ADA_F32_ROWS");
        if (skinny_wide_rows_applies(m) && !(f32_rows && std::atoi(f32_rows))) m.a_frag = 1;
    }
    launch_flow_mod_input(flow_te_.f(), fs.cemb, cs, hd_.flow_cond_b, fs.ymod, N, B, C, st, frame ? &pub : nullptr, ring ? fs.ctl + cw : nullptr,
                          ring ? static_cast<int>(flow_ring_ctl_bytes(B, s.flow_blocks) / 4) - cw : 0, m.a_frag);
    if (model_->ada_packed) {
        launch_skinny_gemm(m, st);
    } else {
        GemmArgs m;
        m.A = fs.ymod; m.lda = C; m.W = model_->ada_all.p; m.wtype = model_->ada_all.wtype; m.C = fs.mods; m.ldc = model_->mod_width;
        m.bias = model_->ada_all_b; m.M = N * B; m.N = model_->mod_width; m.K = C; m.split_a = 1;
        launch_gemm(m, st);
    }
    FlowHeadArgs fa;
    fa.x0 = x0_dev; fa.x0_stride = x0_stride; fa.x_out = x_out_dev; fa.mods = fs.mods; fa.B = B; fa.L = s.latent; fa.C = C; fa.blocks = s.flow_blocks;
    fa.steps = N; fa.mod_width = model_->mod_width;
    fa.w_in = model_->flow_in_packed; fa.b_in = hd_.flow_in_b; fa.w_out = model_->flow_out_packed; fa.b_out = hd_.flow_out_b;
    fa.host_tabs = &model_->flow_tabs; fa.workspace = fs.ws; fa.wtype = model_->flow_wtype; fa.eps = s.flow_eps;
    fa.block_wtype = model_->flow_block_wtype;
    fa.ring_ctl = ring ? fs.ctl : nullptr;
    fa.ring_split = model_->flow_ring_split;
    fa.ring_wg_budget = probe.ring_wg_budget;   // concurrent lanes share the chip: every ring in flight must be fully resident
    fa.x0_step = x0_step; fa.x0_step_stride = x0_step_stride; fa.probe_a = probe_a; fa.probe_b = probe_b;
    if (frame) {   // the counter was advanced when the EOS logits were published: the frame index is *step - 1
        fa.x0_step = frame->step_dev.as<int>(); fa.x0_step_bias = -1;
        fa.latents = frame->latents.f(); fa.max_frames = frame->lat_frames; fa.step_ring = frame->step_ring;
    }
    launch_flow_head(fa, st);
    check_launches("the flow phase");
}

// latent rows (batched mapping: utterance b at lat_dev + b*lat_bs, n rows of `latent` floats)
// -> audio[b][n*frame]   (src/pocket_tts.cpp:454-546 for one chunk of n <= 15 frames)
void Engine::mimi_decode_dev(Batch& b, const float* lat_dev, long long lat_bs, int n, float* audio_dev, long long audio_bs, hipStream_t st,
                             int slot0, int nb, int ring_start, unsigned long long due_mask) {
    const Spec& s = spec();
    if (n <= 0 || n > kDecChunk) throw std::runtime_error("decoder chunk must hold 1..15 frames");
    if (nb < 0) nb = b.B - slot0;
    if (slot0 < 0 || nb <= 0 || slot0 + nb > b.B || b.lm_only) throw std::runtime_error("decoder chunk: slot range out of bounds");
    const int B = nb, L = s.latent, M = s.mimi_dim, U = s.upsample;
    const int T0 = n * U, T0max = kDecChunk * U;
    for (int i = slot0; i < slot0 + nb; ++i)
        if ((due_mask == ~0ull || ((due_mask >> i) & 1ull)) && b.dec_pos_host[i] + T0 > model_->mimi_tab_len) throw std::runtime_error("decoder position exceeds the rotary table");
    phase_begin(2, st);
    // the chunk's latent rows, de-normalised
    if (ring_start >= 0) {
        launch_gather_latents(b.latents.f(), b.lat_frames, slot0, B, ring_start, n, L, hd_.emb_std, hd_.emb_mean, b.zbuf.f(), st);
    } else {
        PTTS_HIP(hipMemcpy2DAsync(b.zbuf.p, static_cast<size_t>(n) * L * 4, lat_dev, static_cast<size_t>(lat_bs) * 4,
                                  static_cast<size_t>(n) * L * 4, B, hipMemcpyDeviceToDevice, st));
        launch_prep_latent(b.zbuf.f(), b.zbuf.f(), B * n, L, 1, hd_.emb_std, hd_.emb_mean, st);
    }
    // state buffers of the range start at slot0; scratch buffers are used from their beginning
    const long long qbs = static_cast<long long>(1 + kDecChunk) * M;
    float* qbuf0 = b.qbuf.f() + slot0 * qbs;
    const DevWeight& wi = hd_.dec_in;
    GemmArgs g;
    g.A = b.zbuf.f(); g.lda = L; g.W = wi.p; g.wtype = wi.wtype; g.M = B * n; g.N = M; g.K = L; g.rpb = n;
    g.a_bs = static_cast<long long>(n) * L;
    g.C = qbuf0 + M; g.c_bs = qbs; g.ldc = M;
    launch_gemm(g, st);
    launch_upsample_dw(qbuf0, qbs, n, M, U, hd_.dec_up_w, b.dx.f(), B, st);
    const int H0 = s.kernel - 1;
    const auto& layers = hd_.dec;
    const long long ybs = static_cast<long long>(H0 + T0max) * M;
    float* ybuf0 = b.ybuf.f() + slot0 * ybs;
    transformer(layers, b.dec_k.p, b.dec_v.p, b.kv16, static_cast<long long>(b.B) * b.dec_cap * M, b.dec_cap, b.dec_pos.as<int>() + slot0, slot0, B, T0,
                b.dx.f(), M, s.mimi_heads, s.mimi_ffn, s.mimi_eps, s.mimi_context, model_->mimi_cos, model_->mimi_sin,
                model_->mimi_tab_len, b.dec_scratch(), ybuf0 + static_cast<long long>(H0) * M, ybs, M, st, false,
                std::getenv("POCKET_TTS_DECODER_F32_ROWS") == nullptr);
    // SEANet decoder
    int c = 8 * s.n_filters;
    float* e0buf0 = b.e0buf.f() + slot0 * static_cast<long long>(1 + T0max) * c;
    {
        const DevWeight& w = hd_.dec_conv0;
        GemmArgs a;
        a.A = ybuf0; a.a_bs = ybs; a.lda = M; a.rpb = T0; a.W = w.p; a.wtype = w.wtype; a.bias = hd_.dec_conv0_b;
        a.M = B * T0; a.N = c; a.K = s.kernel * M;
        a.C = e0buf0 + c; a.c_bs = static_cast<long long>(1 + T0max) * c; a.ldc = c; a.c_act = ACT_ELU;
        if (B == 1 && T0 <= 256 && w.frag && !std::getenv("POCKET_TTS_NO_THIN_PASS")) {
            SkinnyArgs k;   // 80 rows against K = 7 * 512: 16 workgroups as a tiled GEMM
            k.A = a.A; k.lda = a.lda; k.Wp = w.frag; k.wtype = w.wtype; k.M = T0; k.N = c; k.K = a.K; k.bias = a.bias; k.act = ACT_ELU;
            k.C = a.C; k.ldc = c;
            launch_skinny_gemm(k, st);
        } else {
            launch_gemm(a, st);
        }
    }
    const float* prev = e0buf0;
    long long prev_bs = static_cast<long long>(1 + T0max) * c;
    int prev_bf16 = 0;   // the rows of a fused stage are bf16 in memory
    int T = T0, Tmax = T0max;
    for (int i = 0; i < 3; ++i) {
        const int r = s.ratios[i], co = c / 2, T2 = T * r, T2max = Tmax * r, rk = s.res_kernel;
        const int hn = i < 2 ? 1 : s.last_kernel - 1;
        const DevWeight &wu = hd_.dec_up[i], &w1 = hd_.dec_c1[i], &w2 = hd_.dec_c2[i];
        GemmArgs up;  // transposed conv: rows = input steps, N = r * co output samples x channels
        up.A = prev; up.a_bs = prev_bs; up.lda = c; up.rpb = T; up.W = wu.p; up.wtype = wu.wtype; up.a_bf16 = prev_bf16;
        up.bias = hd_.dec_up_b[i]; up.bias_mod = co; up.M = B * T; up.N = r * co; up.K = 2 * c;
        if (dec_fused_[i]) {
            // fused residual unit: the up-sampled rows are kept raw (bf16), behind the stream's two previous rows, in
            // the buffer that otherwise holds their ELU; one kernel does ELU -> conv -> ELU -> conv -> + skip -> ELU
            up.c_bs = static_cast<long long>(rk - 1 + T2max) * co; up.ldc = r * co;
            __bf16* ybuf = reinterpret_cast<__bf16*>(b.uebuf[i].p) + slot0 * up.c_bs;
            up.C = reinterpret_cast<float*>(ybuf + static_cast<long long>(rk - 1) * co); up.c_bf16 = 1;
            launch_gemm(up, st);
            ResblockArgs rb;
            rb.Y = ybuf; rb.y_bs = up.c_bs;
            rb.e_bs = static_cast<long long>(hn + T2max) * co; rb.e_off = hn;
            rb.E = reinterpret_cast<__bf16*>(b.ebuf[i].p) + slot0 * rb.e_bs;
            rb.W1 = w1.p; rb.b1 = hd_.dec_c1_b[i]; rb.W2 = w2.p; rb.b2 = hd_.dec_c2_b[i];
            rb.T = T2; rb.B = B; rb.co = co;
            launch_resblock(rb, st);
            prev = reinterpret_cast<const float*>(rb.E); prev_bs = rb.e_bs; prev_bf16 = 1;
            c = co; T = T2; Tmax = T2max;
            continue;
        }
        up.C = b.ubuf[i].f(); up.c_bs = static_cast<long long>(T2max) * co; up.ldc = r * co;
        up.c2_bs = static_cast<long long>(rk - 1 + T2max) * co;
        float* ue0 = b.uebuf[i].f() + slot0 * up.c2_bs;
        up.C2 = ue0 + static_cast<long long>(rk - 1) * co;
        up.ldc2 = r * co; up.c2_act = ACT_ELU;
        launch_gemm(up, st);
        GemmArgs c1;
        c1.A = ue0; c1.a_bs = up.c2_bs; c1.lda = co; c1.rpb = T2; c1.W = w1.p; c1.wtype = w1.wtype;
        c1.bias = hd_.dec_c1_b[i]; c1.M = B * T2; c1.N = co / 2; c1.K = rk * co;
        c1.C = b.h1buf[i].f(); c1.c_bs = static_cast<long long>(T2max) * (co / 2); c1.ldc = co / 2; c1.c_act = ACT_ELU;
        launch_gemm(c1, st);
        GemmArgs c2;
        c2.A = b.h1buf[i].f(); c2.a_bs = c1.c_bs; c2.lda = co / 2; c2.rpb = T2; c2.W = w2.p; c2.wtype = w2.wtype;
        c2.bias = hd_.dec_c2_b[i]; c2.M = B * T2; c2.N = co; c2.K = co / 2;
        c2.R = b.ubuf[i].f(); c2.r_bs = up.c_bs; c2.ldr = co;
        c2.c_bs = static_cast<long long>(hn + T2max) * co; c2.ldc = co;
        float* e0 = b.ebuf[i].f() + slot0 * c2.c_bs;
        c2.C = e0 + static_cast<long long>(hn) * co;
        c2.c_act = ACT_ELU;
        launch_gemm(c2, st);
        prev = e0; prev_bs = c2.c_bs; prev_bf16 = 0;
        c = co; T = T2; Tmax = T2max;
    }
    launch_conv_to_mono(prev, prev_bf16, prev_bs, T, s.last_kernel * c, c, hd_.dec_final_w, hd_.dec_final_b, audio_dev, audio_bs, B, st);
    // carry the streaming histories and positions to the next chunk
    launch_shift_history(qbuf0, qbs, n, 1, M, B, st, due_mask, slot0);
    launch_shift_history(ybuf0, ybs, T0, H0, M, B, st, due_mask, slot0);
    c = 8 * s.n_filters;
    launch_shift_history(e0buf0, static_cast<long long>(1 + T0max) * c, T0, 1, c, B, st, due_mask, slot0);
    T = T0; Tmax = T0max;
    for (int i = 0; i < 3; ++i) {
        const int co = c / 2, rk = s.res_kernel, hn = i < 2 ? 1 : s.last_kernel - 1;
        T *= s.ratios[i]; Tmax *= s.ratios[i];
        // bf16 rows of a fused stage: the same bytes seen as co/2 floats per row
        const int div = dec_fused_[i] ? 2 : 1;
        const long long ubs = static_cast<long long>(rk - 1 + Tmax) * co / div, ebs = static_cast<long long>(hn + Tmax) * co / div;
        launch_shift_history(b.uebuf[i].f() + slot0 * ubs, ubs, T, rk - 1, co / div, B, st, due_mask, slot0);
        launch_shift_history(b.ebuf[i].f() + slot0 * ebs, ebs, T, hn, co / div, B, st, due_mask, slot0);
        c = co;
    }
    launch_advance_pos(b.dec_pos.as<int>() + slot0, B, T0, st, due_mask, slot0);
    for (int i = slot0; i < slot0 + nb; ++i)
        if (due_mask == ~0ull || ((due_mask >> i) & 1ull)) b.dec_pos_host[i] += T0;
    phase_end(2, st);
    check_launches("the decoder chunk");
}

}  // namespace ptts
