// Reader for the weights inside an ONNX file: the graph's initializers and Constant-node tensors, with no
// dependency on protobuf or onnx.  The reference hands its five .onnx files to ONNX Runtime
// (/root/reference/src/pocket_tts.cpp:90-113); this engine only needs the tensors they carry, so the file is
// walked at the protobuf wire level (onnx.proto3: ModelProto.graph = 7; GraphProto.node = 1, .initializer = 5;
// NodeProto.input = 1, .output = 2, .name = 3, .op_type = 4, .attribute = 5; AttributeProto.name = 1, .t = 5;
// TensorProto.dims = 1, .data_type = 2, .float_data = 4, .int32_data = 5, .int64_data = 7, .name = 8,
// .raw_data = 9, .double_data = 10, .external_data = 13, .data_location = 14).
//
// Quantised graphs made by ONNX Runtime's quantize_dynamic / quantize_static keep a matrix W as three
// initializers <base>_quantized (int8 or uint8), <base>_scale and <base>_zero_point; dequantize() returns
// (q - zero_point) * scale with a per-tensor or per-axis scale.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>

#include "onnx_reader.h"

namespace ptts {

namespace {

struct Cursor {
    const uint8_t* p; const uint8_t* end;
    bool done() const { return p >= end; }
    uint64_t varint() {
        uint64_t v = 0; int shift = 0;
        while (true) {
            if (p >= end || shift > 63) throw std::runtime_error("onnx: truncated varint");
            const uint8_t b = *p++;
            v |= static_cast<uint64_t>(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
            shift += 7;
        }
    }
    Cursor bytes() {
        const uint64_t n = varint();
        if (n > static_cast<uint64_t>(end - p)) throw std::runtime_error("onnx: length-delimited field runs past the end");
        Cursor c{p, p + n};
        p += n;
        return c;
    }
    void skip(int wire) {
        switch (wire) {
            case 0: varint(); break;
            case 1: if (end - p < 8) throw std::runtime_error("onnx: truncated fixed64"); p += 8; break;
            case 2: bytes(); break;
            case 5: if (end - p < 4) throw std::runtime_error("onnx: truncated fixed32"); p += 4; break;
            default: throw std::runtime_error("onnx: unsupported wire type " + std::to_string(wire));
        }
    }
};

std::string str_of(Cursor c) { return std::string(reinterpret_cast<const char*>(c.p), static_cast<size_t>(c.end - c.p)); }

// repeated scalar field, packed (wire 2) or not (wire 0 / 1 / 5)
template <typename F>
void repeated(Cursor& c, int wire, int scalar_wire, F&& take) {
    if (wire == 2) {
        Cursor in = c.bytes();
        while (!in.done()) take(in);
    } else if (wire == scalar_wire) {
        take(c);
    } else {
        throw std::runtime_error("onnx: unexpected wire type in a repeated scalar field");
    }
}

float half_to_float(uint16_t h) {
    const uint32_t sign = static_cast<uint32_t>(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff, bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {   // subnormal half -> normal float
            int e = -1;
            do { ++e; man <<= 1; } while (!(man & 0x400));
            bits = sign | static_cast<uint32_t>(127 - 15 - e) << 23 | (man & 0x3ff) << 13;
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | man << 13;
    } else {
        bits = sign | (exp + 127 - 15) << 23 | man << 13;
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

size_t elem_size(int dtype) {
    switch (dtype) {
        case ONNX_FLOAT: case ONNX_INT32: return 4;
        case ONNX_UINT8: case ONNX_INT8: case ONNX_BOOL: return 1;
        case ONNX_UINT16: case ONNX_INT16: case ONNX_FLOAT16: case ONNX_BFLOAT16: return 2;
        case ONNX_INT64: case ONNX_DOUBLE: return 8;
        default: return 0;
    }
}

OnnxTensor parse_tensor(Cursor c) {
    OnnxTensor t;
    while (!c.done()) {
        const uint64_t key = c.varint();
        const int field = static_cast<int>(key >> 3), wire = static_cast<int>(key & 7);
        switch (field) {
            case 1: repeated(c, wire, 0, [&](Cursor& s) { t.dims.push_back(static_cast<int64_t>(s.varint())); }); break;
            case 2: t.dtype = static_cast<int>(c.varint()); break;
            case 4:
                repeated(c, wire, 5, [&](Cursor& s) {
                    if (s.end - s.p < 4) throw std::runtime_error("onnx: truncated float_data");
                    float f; std::memcpy(&f, s.p, 4); s.p += 4; t.typed_f32.push_back(f);
                });
                break;
            case 5: repeated(c, wire, 0, [&](Cursor& s) { t.typed_i64.push_back(static_cast<int32_t>(s.varint())); }); break;
            case 7: repeated(c, wire, 0, [&](Cursor& s) { t.typed_i64.push_back(static_cast<int64_t>(s.varint())); }); break;
            case 10:
                repeated(c, wire, 1, [&](Cursor& s) {
                    if (s.end - s.p < 8) throw std::runtime_error("onnx: truncated double_data");
                    double d; std::memcpy(&d, s.p, 8); s.p += 8; t.typed_f32.push_back(static_cast<float>(d));
                });
                break;
            case 8: t.name = str_of(c.bytes()); break;
            case 9: { Cursor b = c.bytes(); t.raw.assign(b.p, b.end); t.has_raw = true; break; }
            case 13: {   // StringStringEntryProto { key = 1, value = 2 }
                Cursor e = c.bytes();
                std::string k, v;
                while (!e.done()) {
                    const uint64_t ek = e.varint();
                    if ((ek >> 3) == 1 && (ek & 7) == 2) k = str_of(e.bytes());
                    else if ((ek >> 3) == 2 && (ek & 7) == 2) v = str_of(e.bytes());
                    else e.skip(static_cast<int>(ek & 7));
                }
                if (k == "location") t.ext_location = v;
                else if (k == "offset") t.ext_offset = std::stoll(v);
                else if (k == "length") t.ext_length = std::stoll(v);
                break;
            }
            case 14: t.external = c.varint() == 1; break;
            default: c.skip(wire);
        }
    }
    return t;
}

}  // namespace

int64_t OnnxTensor::numel() const {
    int64_t n = 1;
    for (int64_t d : dims) n *= d;
    return n;
}

OnnxFile::OnnxFile(const std::string& path) : path_(path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("onnx: cannot open " + path);
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<uint8_t> buf(static_cast<size_t>(n));
    if (n && !f.read(reinterpret_cast<char*>(buf.data()), n)) throw std::runtime_error("onnx: cannot read " + path);
    Cursor model{buf.data(), buf.data() + buf.size()};
    bool saw_graph = false;
    while (!model.done()) {
        const uint64_t key = model.varint();
        if ((key >> 3) == 7 && (key & 7) == 2) {
            saw_graph = true;
            Cursor g = model.bytes();
            while (!g.done()) {
                const uint64_t gk = g.varint();
                const int field = static_cast<int>(gk >> 3), wire = static_cast<int>(gk & 7);
                if (field == 5 && wire == 2) {
                    tensors_.push_back(parse_tensor(g.bytes()));
                } else if (field == 1 && wire == 2) {
                    Cursor nc = g.bytes();
                    OnnxNode node;
                    std::vector<OnnxTensor> consts;
                    while (!nc.done()) {
                        const uint64_t nk = nc.varint();
                        const int nf = static_cast<int>(nk >> 3), nw = static_cast<int>(nk & 7);
                        if (nf == 1 && nw == 2) node.inputs.push_back(str_of(nc.bytes()));
                        else if (nf == 2 && nw == 2) node.outputs.push_back(str_of(nc.bytes()));
                        else if (nf == 3 && nw == 2) node.name = str_of(nc.bytes());
                        else if (nf == 4 && nw == 2) node.op_type = str_of(nc.bytes());
                        else if (nf == 5 && nw == 2) {
                            Cursor a = nc.bytes();
                            std::string an;
                            while (!a.done()) {
                                const uint64_t ak = a.varint();
                                if ((ak >> 3) == 1 && (ak & 7) == 2) an = str_of(a.bytes());
                                else if ((ak >> 3) == 5 && (ak & 7) == 2) consts.push_back(parse_tensor(a.bytes()));
                                else a.skip(static_cast<int>(ak & 7));
                            }
                            (void)an;
                        } else {
                            nc.skip(nw);
                        }
                    }
                    if (node.op_type == "Constant" && !node.outputs.empty())
                        for (auto& t : consts) { t.name = node.outputs[0]; tensors_.push_back(std::move(t)); }
                    nodes_.push_back(std::move(node));
                } else {
                    g.skip(wire);
                }
            }
        } else {
            model.skip(static_cast<int>(key & 7));
        }
    }
    if (!saw_graph) throw std::runtime_error("onnx: " + path + " holds no graph (not an ONNX model?)");
    for (size_t i = 0; i < tensors_.size(); ++i) index_[tensors_[i].name] = static_cast<int>(i);
}

int OnnxFile::find(const std::string& name) const {
    auto it = index_.find(name);
    return it == index_.end() ? -1 : it->second;
}

// Element i of tensor t as a double, whatever container the exporter chose.
std::vector<float> OnnxFile::read_f32(int i) const {
    if (i < 0 || i >= static_cast<int>(tensors_.size())) throw std::runtime_error("onnx: tensor index out of range");
    const OnnxTensor& t = tensors_[i];
    const int64_t n = t.numel();
    if (n < 0) throw std::runtime_error("onnx: tensor " + t.name + " has a negative dimension");
    std::vector<float> out(static_cast<size_t>(n));
    std::vector<uint8_t> ext;
    const uint8_t* raw = nullptr;
    size_t raw_bytes = 0;
    if (t.external) {
        if (t.ext_location.empty()) throw std::runtime_error("onnx: external tensor " + t.name + " names no file");
        std::string dir = path_;
        const size_t slash = dir.find_last_of('/');
        dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
        std::ifstream f(dir + "/" + t.ext_location, std::ios::binary);
        if (!f) throw std::runtime_error("onnx: cannot open external data file " + t.ext_location);
        const size_t need = t.ext_length >= 0 ? static_cast<size_t>(t.ext_length) : static_cast<size_t>(n) * elem_size(t.dtype);
        ext.resize(need);
        f.seekg(t.ext_offset);
        if (need && !f.read(reinterpret_cast<char*>(ext.data()), static_cast<std::streamsize>(need)))
            throw std::runtime_error("onnx: external data of " + t.name + " is shorter than its tensor");
        raw = ext.data(); raw_bytes = ext.size();
    } else if (t.has_raw) {
        raw = t.raw.data(); raw_bytes = t.raw.size();
    }
    if (raw) {
        const size_t es = elem_size(t.dtype);
        if (!es) throw std::runtime_error("onnx: tensor " + t.name + " has unsupported data type " + std::to_string(t.dtype));
        if (raw_bytes != static_cast<size_t>(n) * es) throw std::runtime_error("onnx: raw data of " + t.name + " does not match its shape");
        for (int64_t k = 0; k < n; ++k) {
            const uint8_t* e = raw + static_cast<size_t>(k) * es;
            float v = 0.f;
            switch (t.dtype) {
                case ONNX_FLOAT: std::memcpy(&v, e, 4); break;
                case ONNX_DOUBLE: { double d; std::memcpy(&d, e, 8); v = static_cast<float>(d); break; }
                case ONNX_FLOAT16: { uint16_t h; std::memcpy(&h, e, 2); v = half_to_float(h); break; }
                case ONNX_BFLOAT16: { uint16_t h; std::memcpy(&h, e, 2); const uint32_t b = static_cast<uint32_t>(h) << 16; std::memcpy(&v, &b, 4); break; }
                case ONNX_INT8: v = static_cast<float>(*reinterpret_cast<const int8_t*>(e)); break;
                case ONNX_UINT8: case ONNX_BOOL: v = static_cast<float>(*e); break;
                case ONNX_INT16: { int16_t x; std::memcpy(&x, e, 2); v = static_cast<float>(x); break; }
                case ONNX_UINT16: { uint16_t x; std::memcpy(&x, e, 2); v = static_cast<float>(x); break; }
                case ONNX_INT32: { int32_t x; std::memcpy(&x, e, 4); v = static_cast<float>(x); break; }
                case ONNX_INT64: { int64_t x; std::memcpy(&x, e, 8); v = static_cast<float>(x); break; }
            }
            out[static_cast<size_t>(k)] = v;
        }
        return out;
    }
    // typed repeated fields
    if (t.dtype == ONNX_FLOAT || t.dtype == ONNX_DOUBLE) {
        if (static_cast<int64_t>(t.typed_f32.size()) != n) throw std::runtime_error("onnx: float data of " + t.name + " does not match its shape");
        return t.typed_f32;
    }
    if (static_cast<int64_t>(t.typed_i64.size()) != n) throw std::runtime_error("onnx: integer data of " + t.name + " does not match its shape");
    for (int64_t k = 0; k < n; ++k) {
        const int64_t x = t.typed_i64[static_cast<size_t>(k)];
        if (t.dtype == ONNX_FLOAT16) out[static_cast<size_t>(k)] = half_to_float(static_cast<uint16_t>(x));   // int32_data carries the bits
        else if (t.dtype == ONNX_BFLOAT16) { const uint32_t b = static_cast<uint32_t>(x & 0xffff) << 16; float v; std::memcpy(&v, &b, 4); out[static_cast<size_t>(k)] = v; }
        else out[static_cast<size_t>(k)] = static_cast<float>(x);
    }
    return out;
}

std::vector<float> OnnxFile::dequantize(const std::string& base, int axis, std::vector<int64_t>* dims_out) const {
    const int iq = find(base + "_quantized"), is = find(base + "_scale"), iz = find(base + "_zero_point");
    if (iq < 0 || is < 0) throw std::runtime_error("onnx: no " + base + "_quantized / " + base + "_scale pair in " + path_);
    const OnnxTensor& q = tensors_[iq];
    if (q.dtype != ONNX_INT8 && q.dtype != ONNX_UINT8) throw std::runtime_error("onnx: " + q.name + " is not an int8 / uint8 tensor");
    const std::vector<float> qv = read_f32(iq), sv = read_f32(is);
    std::vector<float> zv(sv.size(), 0.f);
    if (iz >= 0) {
        zv = read_f32(iz);
        if (zv.size() != sv.size()) throw std::runtime_error("onnx: scale and zero point of " + base + " differ in length");
    }
    if (dims_out) *dims_out = q.dims;
    std::vector<float> out(qv.size());
    if (sv.size() == 1) {
        for (size_t k = 0; k < qv.size(); ++k) out[k] = (qv[k] - zv[0]) * sv[0];
        return out;
    }
    const int rank = static_cast<int>(q.dims.size());
    if (axis < 0) {   // the one axis whose extent equals the number of scales
        int hit = -1, hits = 0;
        for (int d = 0; d < rank; ++d)
            if (q.dims[d] == static_cast<int64_t>(sv.size())) { hit = d; ++hits; }
        if (hits != 1) throw std::runtime_error("onnx: cannot tell which axis of " + q.name + " its " + std::to_string(sv.size()) + " scales run along; name the axis");
        axis = hit;
    }
    if (axis >= rank || q.dims[axis] != static_cast<int64_t>(sv.size())) throw std::runtime_error("onnx: scales of " + base + " do not match axis " + std::to_string(axis));
    int64_t inner = 1;
    for (int d = axis + 1; d < rank; ++d) inner *= q.dims[d];
    for (size_t k = 0; k < qv.size(); ++k) {
        const size_t c = static_cast<size_t>((static_cast<int64_t>(k) / inner) % q.dims[axis]);
        out[k] = (qv[k] - zv[c]) * sv[c];
    }
    return out;
}

}  // namespace ptts
