#!/usr/bin/env python3
"""Builds the engine's weight pack (.ptts) from the tensors inside ONNX files.

The reference loads `<models_dir>/{mimi_encoder,text_conditioner}.onnx` and
`{flow_lm_main,flow_lm_flow,mimi_decoder}{,_int8}.onnx` (src/pocket_tts.cpp:90-96).  Exporters rename weights
(`onnx::MatMul_123`), store Linear matrices transposed and, in the `_int8` graphs, as int8 with scale and zero
point, so which initializer feeds which pack tensor is written down in a map file rather than guessed:

    {"spec": {"d_model": 1024, ...},                       # ptts_pack.Spec fields that differ from the defaults
     "tensors": {
        "lm.l0.qkv.w":  {"file": "flow_lm_main.onnx", "tensor": "onnx::MatMul_812", "transpose": true},
        "lm.l0.ff1.w":  {"file": "flow_lm_main_int8.onnx", "tensor": "onnx::MatMul_813", "dequantize": true, "transpose": true},
        "dec.conv0.w":  {"file": "mimi_decoder.onnx", "tensor": "decoder.model.0.conv.weight", "conv": true},
        "lm.l0.ln1.g":  {"file": "flow_lm_main.onnx", "tensor": "transformer.layers.0.norm1.weight"},
        "flow.b0.ada.w": {"concat": [ {...}, {...} ]}      # rows of several tensors stacked along axis 0
     }}

Per entry: `transpose` swaps a 2-D matrix to [out][in]; `conv` turns a Conv1d weight [Cout][Cin][k] into the pack's
[Cout][k*Cin] (K index = tap*Cin + ci); `dequantize` (default: automatic when `<tensor>_quantized` exists) applies
(q - zero_point) * scale with `axis` (default: the axis the scales fit); `rows: [a, b]` keeps rows a..b-1;
`reshape` forces the final shape.

`--list` prints every tensor (name, type, shape) and, for MatMul / Gemm / Conv / ConvTranspose / Gather nodes, which
initializer each consumes: the raw material for writing the map.  `--auto` maps by equal names (files whose
initializers are already named like pack tensors), transposing 2-D tensors whose shape is the reverse of the pack's.
"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pyhost"))

import ptts_onnx  # noqa: E402
import ptts_pack as pp  # noqa: E402

REFERENCE_FILES = ["mimi_encoder.onnx", "text_conditioner.onnx", "flow_lm_main.onnx", "flow_lm_flow.onnx", "mimi_decoder.onnx",
                   "flow_lm_main_int8.onnx", "flow_lm_flow_int8.onnx", "mimi_decoder_int8.onnx"]


def open_all(models_dir, only=None):
    files = {}
    for fn in sorted(os.listdir(models_dir)):
        if fn.endswith(".onnx") and (only is None or fn in only):
            files[fn] = ptts_onnx.OnnxWeights(os.path.join(models_dir, fn))
    return files


def list_models(models_dir):
    for fn, ow in open_all(models_dir).items():
        print(f"== {fn}: {len(ow.info)} tensors")
        for name, (_, dt, shape) in ow.info.items():
            print(f"   {name}  {dt}  {list(shape)}")
        for op, ins, outs in ow.nodes():
            if op in ("MatMul", "Gemm", "Conv", "ConvTranspose", "Gather", "MatMulInteger", "QLinearMatMul", "DequantizeLinear"):
                fed = [i for i in ins if i in ow.info]
                if fed:
                    print(f"   node {op} -> {outs[0] if outs else '?'} uses {fed}")


def fetch(files, entry, want_shape):
    if "concat" in entry:
        parts = [fetch(files, e, None) for e in entry["concat"]]
        a = np.concatenate(parts, axis=0)
    else:
        ow = files.get(entry["file"])
        if ow is None:
            raise KeyError(f"{entry['file']} is not in the models directory")
        name = entry["tensor"]
        deq = entry.get("dequantize", None)
        if deq is None:
            deq = name not in ow.info and ow.has_quantized(name)
        a = ow.dequantize(name, int(entry.get("axis", -1))) if deq else ow.read(name)
        if entry.get("conv"):
            if a.ndim != 3:
                raise ValueError(f"{name}: conv weights are rank 3, got {a.shape}")
            a = np.ascontiguousarray(a.transpose(0, 2, 1)).reshape(a.shape[0], -1)   # [Cout][Cin][k] -> [Cout][k*Cin]
        if entry.get("transpose"):
            a = np.ascontiguousarray(a.T)
        if "rows" in entry:
            a = a[int(entry["rows"][0]):int(entry["rows"][1])]
    if "reshape" in entry:
        a = a.reshape(entry["reshape"])
    a = np.ascontiguousarray(a, np.float32)
    if want_shape is not None and tuple(a.shape) != tuple(want_shape):
        if int(np.prod(a.shape)) == int(np.prod(want_shape)) and sum(d != 1 for d in a.shape) == sum(d != 1 for d in want_shape):
            a = a.reshape(want_shape)   # only unit axes differ
        else:
            raise ValueError(f"{entry}: shape {a.shape}, the pack wants {tuple(want_shape)}")
    return a


def auto_map(files, table):
    tensors = {}
    for name, shape, _ in table:
        for fn, ow in files.items():
            if name in ow.info or ow.has_quantized(name):
                src_shape = ow.info[name][2] if name in ow.info else ow.info[name + "_quantized"][2]
                e = {"file": fn, "tensor": name}
                if len(shape) == 2 and tuple(src_shape) == tuple(reversed(shape)) and shape[0] != shape[1]:
                    e["transpose"] = True
                tensors[name] = e
                break
    return tensors


def convert(models_dir, out_path, map_path=None, auto=False, spec_over=None):
    m = json.load(open(map_path)) if map_path else {}
    spec_kw = dict(m.get("spec", {}))
    spec_kw.update(spec_over or {})
    spec = pp.Spec(**spec_kw)
    table = pp.tensor_table(spec)
    files = open_all(models_dir)
    tensors = auto_map(files, table) if auto else {}
    tensors.update(m.get("tensors", {}))
    missing = [name for name, _, _ in table if name not in tensors]
    if missing:
        raise SystemExit(f"{len(missing)} pack tensors have no source, e.g. {missing[:8]}; add them to the map (see --list)")
    weights = {name: fetch(files, tensors[name], shape) for name, shape, _ in table}
    pp.write_pack(out_path, spec, weights)
    for ow in files.values():
        ow.close()
    return spec, weights


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("models_dir")
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--map")
    ap.add_argument("--auto", action="store_true")
    ap.add_argument("--out", default=None, help="default: <models_dir>/pocket_tts.ptts")
    args = ap.parse_args()
    if args.list:
        list_models(args.models_dir)
        return
    if not args.map and not args.auto:
        ap.error("give --map FILE and/or --auto (or --list to see what the files hold)")
    out = args.out or os.path.join(args.models_dir, "pocket_tts.ptts")
    spec, w = convert(args.models_dir, out, args.map, args.auto)
    print(f"wrote {out}: {len(w)} tensors, {sum(a.nbytes for a in w.values()) / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
