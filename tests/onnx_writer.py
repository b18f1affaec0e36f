"""Minimal ONNX file writer for the tests (protobuf wire format by hand; the `onnx` package is not installed).

Writes ModelProto{ir_version, graph{node*, initializer*}} with TensorProto payloads in the forms exporters use:
raw_data, float_data / int32_data / int64_data, external data files, and Constant nodes.
"""
import struct

import numpy as np

DT = {np.dtype(np.float32): 1, np.dtype(np.uint8): 2, np.dtype(np.int8): 3, np.dtype(np.int32): 6, np.dtype(np.int64): 7,
      np.dtype(np.float16): 10, np.dtype(np.float64): 11}


def varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def key(field: int, wire: int) -> bytes:
    return varint((field << 3) | wire)


def ld(field: int, payload: bytes) -> bytes:
    return key(field, 2) + varint(len(payload)) + payload


def tensor(name, a: np.ndarray, form="raw", dtype_code=None, ext=None) -> bytes:
    """form: raw | typed | typed_unpacked | external (ext = (file name, offset, length))"""
    shape = np.shape(a)
    a = np.ascontiguousarray(a).reshape(shape)
    code = dtype_code if dtype_code is not None else DT[a.dtype]
    out = b""
    if form == "typed_unpacked":
        for d in a.shape:
            out += key(1, 0) + varint(int(d))
    else:
        out += ld(1, b"".join(varint(int(d)) for d in a.shape))
    out += key(2, 0) + varint(code)
    if form == "raw":
        out += ld(9, a.tobytes())
    elif form in ("typed", "typed_unpacked"):
        if a.dtype == np.float32:
            out += ld(4, a.tobytes()) if form == "typed" else b"".join(key(4, 5) + struct.pack("<f", float(v)) for v in a.ravel())
        elif a.dtype == np.float64:
            out += ld(10, a.tobytes())
        elif a.dtype == np.int64:
            out += ld(7, b"".join(varint(int(v)) for v in a.ravel()))
        elif a.dtype == np.float16:   # the bits travel in int32_data
            out += ld(5, b"".join(varint(int(v)) for v in a.view(np.uint16).ravel()))
        else:
            out += ld(5, b"".join(varint(int(v)) for v in a.ravel()))
    elif form == "external":
        fn, off, length = ext
        for k, v in (("location", fn), ("offset", str(off)), ("length", str(length))):
            out += ld(13, ld(1, k.encode()) + ld(2, v.encode()))
        out += key(14, 0) + varint(1)
    else:
        raise ValueError(form)
    out += ld(8, name.encode())
    return out


def node(op_type, inputs, outputs, name="", const=None) -> bytes:
    out = b"".join(ld(1, i.encode()) for i in inputs) + b"".join(ld(2, o.encode()) for o in outputs)
    out += ld(3, name.encode()) + ld(4, op_type.encode())
    if const is not None:   # AttributeProto{name = "value", t = tensor, type = TENSOR(4)}
        out += ld(5, ld(1, b"value") + ld(5, const) + key(20, 0) + varint(4))
    return out


def model(initializers, nodes=()) -> bytes:
    g = b"".join(ld(1, n) for n in nodes) + ld(2, b"g") + b"".join(ld(5, t) for t in initializers)
    return key(1, 0) + varint(8) + ld(2, b"tests/onnx_writer.py") + ld(7, g) + ld(8, ld(1, b"") + key(2, 0) + varint(17))


def quantize_per_axis(w: np.ndarray, axis: int, unsigned=False):
    """ONNX Runtime style weight quantisation: returns (q, scale, zero_point) with w ~ (q - zp) * scale."""
    mv = np.moveaxis(w, axis, 0).reshape(w.shape[axis], -1)
    if unsigned:
        lo, hi = np.minimum(mv.min(1), 0), np.maximum(mv.max(1), 0)
        scale = np.maximum((hi - lo) / 255.0, 1e-12).astype(np.float32)
        zp = np.clip(np.rint(-lo / scale), 0, 255).astype(np.uint8)
        shp = [1] * w.ndim
        shp[axis] = -1
        q = np.clip(np.rint(w / scale.reshape(shp)) + zp.reshape(shp).astype(np.float32), 0, 255).astype(np.uint8)
        return q, scale, zp
    scale = np.maximum(np.abs(mv).max(1) / 127.0, 1e-12).astype(np.float32)
    shp = [1] * w.ndim
    shp[axis] = -1
    q = np.clip(np.rint(w / scale.reshape(shp)), -127, 127).astype(np.int8)
    return q, scale, np.zeros(w.shape[axis], np.int8)
