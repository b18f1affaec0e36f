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

Matrices that the int8 engine multiplies in int8 (`ptts_pack.is_int8_weight`: the LM layers' qkv / out / ff1 / ff2 and, for
engines started with POCKET_TTS_Q8_FLOW=1, the flow blocks' fc1 / fc2) and that arrive as a `_quantized` / `_scale` / `_zero_point` triplet are carried into the pack as
they are — int8 values (q - zero_point) and one scale per output row, next to the dequantised f32 tensor — whenever the
triplet allows it: scale per tensor or per output channel, q - zero_point within int8.  The int8 engine then multiplies
with the source's own integers; other triplets are only dequantised.

How the int8 engine quantises ACTIVATIONS is a field of the pack header (`q8_act_scheme`).  A graph file that holds
DynamicQuantizeLinear / MatMulInteger nodes quantises every MatMul input per call as uint8 with one scale and one zero
point; when an LM-layer matrix comes from such a file the pack is written with `q8_act_scheme` = 1 and the engine does
the same per row (flow blocks then stay bf16 and are not carried as int8).  `"q8_act_scheme"` in the map's "spec" overrides.

`--list` prints every tensor (name, type, shape) and, for MatMul / Gemm / Conv / ConvTranspose / Gather nodes, which
initializer each consumes: the raw material for writing the map.  `--auto` maps by equal names (files whose
initializers are already named like pack tensors), transposing 2-D tensors whose shape is the reverse of the pack's.
`--guess OUT.json` writes a first draft of the map for files whose initializers carry exporter names: per graph file it
walks the nodes in graph order and gives every pack tensor of that graph (in the pack's own layer order) the first use,
not yet taken, of a tensor whose shape fits (as stored, transposed, or as a Conv1d weight).  The consuming node settles
what shapes cannot: a MatMul's second input is [in][out] and is transposed even when square.  Separate q / k / v matrices
become one "concat" entry, Constant-node tensors count like initializers, a tensor consumed by two nodes may serve two pack
tensors ("shared_with"), a square quantised matrix gets its scale axis from its consumer ("axis").  Guesses the file does not settle carry a "note"; what cannot be placed is listed under
"unresolved" with the shapes still free in that file.  The draft is for a person to review against `--list`: shapes
repeat from layer to layer, and only the order of the graph tells them apart.

What the reference itself pins are the names of the graphs' inputs and outputs, not of their weights:
mimi_encoder `audio` -> `latents` (src/pocket_tts.cpp:266-267); text_conditioner `token_ids` -> `embeddings` (:305-306);
flow_lm_main `sequence`, `text_embeddings`, `state_<i>` -> conditioning, EOS logit, `out_state_<i>` (:340-345, :400-402);
flow_lm_flow `c`, `s`, `t`, `x` -> `flow_dir` (:439-440); mimi_decoder `latent`, `state_<i>` -> audio, `out_state_<i>`
(:483, :534-536).  GRAPH_IO below holds them.
"""
import argparse
import dataclasses
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pyhost"))

import ptts_onnx  # noqa: E402
import ptts_pack as pp  # noqa: E402

GRAPH_IO = {
    "mimi_encoder": (["audio"], ["latents"]),
    "text_conditioner": (["token_ids"], ["embeddings"]),
    "flow_lm_main": (["sequence", "text_embeddings", "state_<i>"], ["<conditioning>", "<eos logit>", "out_state_<i>"]),
    "flow_lm_flow": (["c", "s", "t", "x"], ["flow_dir"]),
    "mimi_decoder": (["latent", "state_<i>"], ["<audio>", "out_state_<i>"]),
}


def graph_file(name, int8=False):
    """The file of the reference's model directory that holds pack tensor `name` (src/pocket_tts.cpp:90-96)."""
    suffix = "_int8" if int8 else ""
    if name.startswith("enc."):
        return "mimi_encoder.onnx"
    if name.startswith("text."):
        return "text_conditioner.onnx"
    if name.startswith("lm."):
        return f"flow_lm_main{suffix}.onnx"
    if name.startswith("flow."):
        return f"flow_lm_flow{suffix}.onnx"
    return f"mimi_decoder{suffix}.onnx"


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


def fetch_q8(files, entry, want_shape):
    """The int8 form (q [N][K] int8, scale [N] f32) of a map entry whose source is a quantised triplet, or None."""
    if "concat" in entry or "rows" in entry or "reshape" in entry or entry.get("conv") or want_shape is None or len(want_shape) != 2:
        return None
    ow = files.get(entry["file"])
    name = entry["tensor"]
    if ow is None or not ow.has_quantized(name) or entry.get("dequantize") is False:
        return None
    q = ow.read(name + "_quantized")
    sc = ow.read(name + "_scale").reshape(-1)
    zp = ow.read(name + "_zero_point").reshape(-1) if name + "_zero_point" in ow.info else np.zeros(1, np.float32)
    if q.ndim != 2:
        return None
    if entry.get("transpose"):
        q = q.T                      # now [out][in]; a per-column scale of the stored matrix is per output row here
        out_axis_src = 1
    else:
        out_axis_src = 0
    axis = int(entry.get("axis", -1))
    N = q.shape[0]
    if sc.size == 1:
        sc = np.full(N, sc[0], np.float32)
    elif sc.size == N and (axis in (-1, out_axis_src)):
        pass
    else:
        return None                  # scales along the input axis: no per-row form
    if zp.size == 1:
        zp = np.full(N, zp[0], np.float32)
    elif zp.size != N:
        return None
    qi = np.rint(q - zp[:, None])
    if tuple(qi.shape) != tuple(want_shape) or qi.min() < -128 or qi.max() > 127:
        return None
    return np.ascontiguousarray(qi.astype(np.int8)), np.ascontiguousarray(sc, np.float32)


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


def consumption_order(ow):
    """Every use of a weight tensor by a node, in graph order: (base name, stored shape, op type, input position), a
    tensor consumed by two nodes appearing twice.  A Constant node's tensor carries its output name, so it is seen where
    a later node consumes that name.  Tensors that no node consumes follow in file order (op None)."""
    def base_of(nm):
        if nm.endswith(("_scale", "_zero_point")):
            return None
        base = nm[:-len("_quantized")] if nm.endswith("_quantized") else nm
        key = base + "_quantized" if base + "_quantized" in ow.info else base
        return (base, tuple(ow.info[key][2])) if key in ow.info else None

    events, seen = [], set()
    for op, ins, _ in ow.nodes():
        if op in ("DynamicQuantizeLinear", "Constant"):
            continue
        for pos, nm in enumerate(ins):
            hit = base_of(nm)
            if hit:
                events.append((hit[0], hit[1], op, pos))
                seen.add(hit[0])
    for nm in ow.info:
        hit = base_of(nm)
        if hit and hit[0] not in seen:
            seen.add(hit[0])
            events.append((hit[0], hit[1], None, -1))
    return events


def guess_map(models_dir, spec, int8=False):
    """First draft of a name map by shape and graph order; returns (tensors, unresolved).

    Per graph file the pack's tensors are taken in the pack's own order and each gets the first use, not yet taken, of a
    tensor whose shape fits.  What the shapes cannot tell, the consuming node does: the second input of a MatMul /
    MatMulInteger is stored [in][out] and is transposed even when square.  A fused projection that the file holds as
    separate matrices (q, k, v) is drafted as a "concat" of consecutive uses.  A tensor consumed by two nodes may serve two
    pack tensors; the second entry says so ("shared_with").  Entries that rest on a guess the shapes do not settle carry a
    "note"; what cannot be placed is listed under unresolved, with the shapes still free in that file."""
    files = open_all(models_dir)
    order = {fn: consumption_order(ow) for fn, ow in files.items()}
    tensors, unresolved = {}, []
    nscales = {fn: {nm[:-len("_scale")]: int(np.prod(info[2])) if info[2] else 1 for nm, info in ow.info.items() if nm.endswith("_scale")}
               for fn, ow in files.items()}     # quantised triplets: how many scales each has
    used = {fn: set() for fn in files}          # indices into order[fn]
    owner = {fn: {} for fn in files}            # tensor name -> pack tensor that took it first
    squeeze = lambda t: tuple(d for d in t if d != 1)

    def fit(fn, ev, want):
        """Map entry if this use can be the pack tensor of shape `want`, else None."""
        base, have, op, pos = ev
        ent = {"file": fn, "tensor": base, "guessed": True}
        matmul_b = op in ("MatMul", "MatMulInteger") and pos == 1
        if len(have) == 2 and have[0] == have[1] and nscales[fn].get(base, 1) == have[0]:
            # a square quantised matrix with one scale per row or column: the shapes do not say which.  Per-channel
            # quantisation of a MatMul weight [in][out] runs along its output axis (1); a Gemm weight [out][in] along 0.
            ent["axis"] = 1 if matmul_b else 0
            if not matmul_b and op != "Gemm":
                ent["note"] = "square quantised matrix and no MatMul/Gemm consumer: scale axis and orientation unknown"
        if len(want) == 2 and len(have) == 2 and have[0] == have[1] == want[0] == want[1]:
            if matmul_b:
                ent["transpose"] = True
            elif op not in ("Gemm",) and "note" not in ent:
                ent["note"] = "square matrix and no MatMul/Gemm consumer: stored orientation unknown"
            return ent
        if have == want or squeeze(have) == squeeze(want):
            if matmul_b and len(want) == 2 and len(squeeze(want)) == 2:
                return None                     # a MatMul weight stored [out][in] would not multiply: another tensor
            return ent
        if len(want) == 2 and len(have) == 2 and have == want[::-1]:
            ent["transpose"] = True
            return ent
        if len(want) == 2 and len(have) == 3 and have[0] == want[0] and have[1] * have[2] == want[1]:
            ent["conv"] = True
            return ent
        return None

    def take(fn, idx, name, ent):
        used[fn].add(idx)
        first = owner[fn].setdefault(ent["tensor"], name)
        if first != name:
            ent["shared_with"] = first
        return ent

    for name, shape, _ in pp.tensor_table(spec):
        fn = graph_file(name, int8)
        if fn not in files and int8:
            fn = graph_file(name, False)
        want = tuple(shape)
        evs = order.get(fn, [])
        hit = None
        for idx, ev in enumerate(evs):
            if idx in used[fn]:
                continue
            ent = fit(fn, ev, want)
            if ent:
                hit = take(fn, idx, name, ent)
                break
        if hit is None and len(want) == 2 and name.endswith(".qkv.w") and want[0] == 3 * want[1]:
            # separate q / k / v projections: the next three free uses of a [D][D] matrix, in graph order
            part = (want[1], want[1])
            free = [(idx, ev) for idx, ev in enumerate(evs) if idx not in used[fn] and fit(fn, ev, part)]
            if len(free) >= 3:
                parts = [take(fn, idx, name, fit(fn, ev, part)) for idx, ev in free[:3]]
                hit = {"concat": parts, "guessed": True, "note": "q, k, v taken as the next three free [D][D] matrices in graph order"}
        if hit:
            tensors[name] = hit
        else:
            left = sorted({ev[1] for idx, ev in enumerate(evs) if idx not in used[fn]})
            unresolved.append({"pack_tensor": name, "shape": list(want), "file": fn, "free_shapes_in_file": [list(t) for t in left][:12]})
    for ow in files.values():
        ow.close()
    return tensors, unresolved


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
    if "q8_act_scheme" not in spec_kw:
        dynamic = {fn for fn, ow in files.items() if any(op in ("DynamicQuantizeLinear", "MatMulInteger") for op, _, _ in ow.nodes())}
        lm_sources = {tensors[name].get("file") for name, _, _ in table if name.startswith("lm.l") and pp.is_int8_weight(name, spec.flow_dim)}
        if dynamic & lm_sources:
            spec = dataclasses.replace(spec, q8_act_scheme=1)
    q8 = {}
    for name, shape, _ in table:
        if pp.is_int8_weight(name, spec.flow_dim, spec.q8_act_scheme, flow_blocks=True):   # flow blocks: for POCKET_TTS_Q8_FLOW=1
            got = fetch_q8(files, tensors[name], shape)
            if got is not None:
                q8[name] = got
    pp.write_pack(out_path, spec, weights, q8=q8)
    for ow in files.values():
        ow.close()
    convert.last_q8 = sorted(q8)
    return spec, weights


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("models_dir")
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--map")
    ap.add_argument("--auto", action="store_true")
    ap.add_argument("--out", default=None, help="default: <models_dir>/pocket_tts.ptts")
    ap.add_argument("--guess", metavar="OUT.json", help="write a draft map by shape and graph order (review it against --list)")
    ap.add_argument("--int8", action="store_true", help="--guess: take the LM, flow and decoder tensors from the _int8 graphs")
    args = ap.parse_args()
    if args.list:
        list_models(args.models_dir)
        return
    if args.guess:
        spec_kw = json.load(open(args.map)).get("spec", {}) if args.map else {}
        spec = pp.Spec(**spec_kw)
        tensors, unresolved = guess_map(args.models_dir, spec, args.int8)
        full_kw = {f: getattr(spec, f) for f in pp.SPEC_INT_FIELDS + pp.SPEC_FLOAT_FIELDS}
        json.dump({"spec": full_kw, "tensors": tensors, "unresolved": unresolved, "graph_io": GRAPH_IO}, open(args.guess, "w"), indent=1)
        print(f"wrote {args.guess}: {len(tensors)} tensors placed, {len(unresolved)} unresolved")
        return
    if not args.map and not args.auto:
        ap.error("give --map FILE and/or --auto (or --list to see what the files hold)")
    out = args.out or os.path.join(args.models_dir, "pocket_tts.ptts")
    spec, w = convert(args.models_dir, out, args.map, args.auto)
    print(f"wrote {out}: {len(w)} tensors, {sum(a.nbytes for a in w.values()) / 1e6:.1f} MB; {len(convert.last_q8)} matrices carried in int8")


if __name__ == "__main__":
    main()
