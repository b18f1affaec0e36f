"""ctypes mirror of the ptts_onnx_* entry points: the tensors an ONNX file carries, without onnx / protobuf.

The reference's model directory holds five .onnx graphs (src/pocket_tts.cpp:90-96); `OnnxWeights` lists and reads
their initializers so that tools/onnx_to_ptts.py can build the engine's weight pack from them.  Host only.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import numpy as np

import ptts_lib

DTYPE_NAMES = {1: "float32", 2: "uint8", 3: "int8", 4: "uint16", 5: "int16", 6: "int32", 7: "int64", 9: "bool", 10: "float16",
               11: "float64", 16: "bfloat16"}


def _lib():
    L = ptts_lib.load()
    if not getattr(L, "_onnx_ready", False):
        L.ptts_onnx_open.restype = C.c_void_p
        L.ptts_onnx_open.argtypes = [C.c_char_p]
        L.ptts_onnx_close.argtypes = [C.c_void_p]
        L.ptts_onnx_tensor_count.argtypes = [C.c_void_p]
        L.ptts_onnx_tensor_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                            C.POINTER(C.c_longlong)]
        L.ptts_onnx_find.argtypes = [C.c_void_p, C.c_char_p]
        L.ptts_onnx_read_f32.restype = C.c_longlong
        L.ptts_onnx_read_f32.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_longlong]
        L.ptts_onnx_dequantize.restype = C.c_longlong
        L.ptts_onnx_dequantize.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_float), C.c_longlong, C.POINTER(C.c_int),
                                           C.POINTER(C.c_longlong)]
        L.ptts_onnx_node_count.argtypes = [C.c_void_p]
        L.ptts_onnx_node_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.pocket_tts_get_last_error.restype = C.c_char_p
        L._onnx_ready = True
    return L


class OnnxWeights:
    def __init__(self, path: str):
        self._L = _lib()
        self._h = self._L.ptts_onnx_open(path.encode())
        if not self._h:
            raise RuntimeError((self._L.pocket_tts_get_last_error() or b"").decode())
        self.path = path
        self.info: Dict[str, Tuple[int, str, Tuple[int, ...]]] = {}   # name -> (index, dtype name, shape)
        for i in range(self._L.ptts_onnx_tensor_count(self._h)):
            name, dt, rank = C.c_char_p(), C.c_int(), C.c_int()
            dims = (C.c_longlong * 8)()
            if self._L.ptts_onnx_tensor_info(self._h, i, C.byref(name), C.byref(dt), C.byref(rank), dims) != 0:
                raise RuntimeError((self._L.pocket_tts_get_last_error() or b"").decode())
            self.info[name.value.decode()] = (i, DTYPE_NAMES.get(dt.value, str(dt.value)), tuple(dims[d] for d in range(rank.value)))

    def close(self):
        if self._h:
            self._L.ptts_onnx_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def names(self) -> List[str]:
        return list(self.info)

    def read(self, name: str) -> np.ndarray:
        """Tensor `name` as float32 (any stored element type)."""
        if name not in self.info:
            raise KeyError(f"{name} is not a tensor of {self.path}")
        i, _, shape = self.info[name]
        n = int(np.prod(shape)) if shape else 1
        out = np.empty(max(n, 1), np.float32)
        got = self._L.ptts_onnx_read_f32(self._h, i, out.ctypes.data_as(C.POINTER(C.c_float)), out.size)
        if got < 0:
            raise RuntimeError((self._L.pocket_tts_get_last_error() or b"").decode())
        return out[:got].reshape(shape)

    def has_quantized(self, base: str) -> bool:
        return base + "_quantized" in self.info and base + "_scale" in self.info

    def dequantize(self, base: str, axis: int = -1) -> np.ndarray:
        """(q - zero_point) * scale of the <base>_quantized / _scale / _zero_point triplet."""
        if not self.has_quantized(base):
            raise KeyError(f"{base}_quantized / {base}_scale are not tensors of {self.path}")
        shape = self.info[base + "_quantized"][2]
        out = np.empty(max(int(np.prod(shape)), 1), np.float32)
        rank = C.c_int()
        dims = (C.c_longlong * 8)()
        got = self._L.ptts_onnx_dequantize(self._h, base.encode(), axis, out.ctypes.data_as(C.POINTER(C.c_float)), out.size, C.byref(rank), dims)
        if got < 0:
            raise RuntimeError((self._L.pocket_tts_get_last_error() or b"").decode())
        return out[:got].reshape(tuple(dims[d] for d in range(rank.value)))

    def nodes(self) -> List[Tuple[str, List[str], List[str]]]:
        """(op type, input names, output names) of every node, in graph order."""
        out = []
        cap = 4096
        for i in range(self._L.ptts_onnx_node_count(self._h)):
            op, ins, outs = C.create_string_buffer(cap), C.create_string_buffer(cap), C.create_string_buffer(cap)
            if self._L.ptts_onnx_node_info(self._h, i, op, ins, outs, cap) != 0:
                raise RuntimeError((self._L.pocket_tts_get_last_error() or b"").decode())
            out.append((op.value.decode(), [s for s in ins.value.decode().split(",") if s], [s for s in outs.value.decode().split(",") if s]))
        return out
