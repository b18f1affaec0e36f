// Tensors carried by an ONNX file (initializers and Constant nodes); see onnx_reader.cpp.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace ptts {

enum OnnxDType { ONNX_FLOAT = 1, ONNX_UINT8 = 2, ONNX_INT8 = 3, ONNX_UINT16 = 4, ONNX_INT16 = 5, ONNX_INT32 = 6, ONNX_INT64 = 7,
                 ONNX_BOOL = 9, ONNX_FLOAT16 = 10, ONNX_DOUBLE = 11, ONNX_BFLOAT16 = 16 };

struct OnnxTensor {
    std::string name;
    int dtype = 0;
    std::vector<int64_t> dims;
    bool has_raw = false, external = false;
    std::vector<uint8_t> raw;
    std::vector<float> typed_f32;     // float_data / double_data
    std::vector<int64_t> typed_i64;   // int32_data / int64_data
    std::string ext_location; long long ext_offset = 0, ext_length = -1;
    int64_t numel() const;
};

struct OnnxNode {
    std::string name, op_type;
    std::vector<std::string> inputs, outputs;
};

class OnnxFile {
  public:
    explicit OnnxFile(const std::string& path);
    const std::vector<OnnxTensor>& tensors() const { return tensors_; }
    const std::vector<OnnxNode>& nodes() const { return nodes_; }
    int find(const std::string& name) const;                 // -1 when absent
    std::vector<float> read_f32(int i) const;                // any supported element type -> float
    // (q - zero_point) * scale of <base>_quantized / _scale / _zero_point; axis -1 = the one axis the scales fit
    std::vector<float> dequantize(const std::string& base, int axis, std::vector<int64_t>* dims_out) const;

  private:
    std::string path_;
    std::vector<OnnxTensor> tensors_;
    std::vector<OnnxNode> nodes_;
    std::unordered_map<std::string, int> index_;
};

}  // namespace ptts
