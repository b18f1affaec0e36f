"""The ONNX weight reader and the .onnx -> .ptts converter (host only).

No real Pocket TTS ONNX file exists offline, so these tests check the reader against files written by
tests/onnx_writer.py in every storage form an exporter may choose, and the converter by a round trip: a pack is
exported as five ONNX files (Linear matrices transposed like torch's MatMul export, conv weights in
[Cout][Cin][k], some matrices int8 with scales), converted back and compared.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pocket-tts.cpp_amd", "tools"))

import onnx_writer as ow


@pytest.fixture(scope="module")
def ponnx():
    import ptts_onnx
    return ptts_onnx


def test_reader_storage_forms(tmp_path, ponnx):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((5, 7)).astype(np.float32)
    b = rng.standard_normal((3, 2, 4)).astype(np.float32)
    h = rng.standard_normal((6,)).astype(np.float16)
    i64 = np.array([[1, -2, 3], [40000000000, -5, 6]], np.int64)
    d = rng.standard_normal((4,)).astype(np.float64)
    bf = rng.standard_normal((8,)).astype(np.float32)
    bf_bits = (bf.view(np.uint32) >> 16).astype(np.uint16)
    ext = rng.standard_normal((9, 3)).astype(np.float32)
    with open(tmp_path / "weights.bin", "wb") as f:
        f.write(b"\0" * 64)
        f.write(ext.tobytes())
    c = rng.standard_normal((2, 2)).astype(np.float32)
    inits = [ow.tensor("a_raw", a), ow.tensor("b_typed", b, "typed"), ow.tensor("b_unpacked", b, "typed_unpacked"),
             ow.tensor("h_raw", h), ow.tensor("h_typed", h, "typed"), ow.tensor("i64_typed", i64, "typed"), ow.tensor("i64_raw", i64),
             ow.tensor("d_typed", d, "typed"), ow.tensor("bf_raw", bf_bits, dtype_code=16),
             ow.tensor("ext", ext, "external", ext=("weights.bin", 64, ext.nbytes)),
             ow.tensor("scalar", np.float32(2.5).reshape(()))]
    nodes = [ow.node("Constant", [], ["c_out"], "c0", const=ow.tensor("", c)), ow.node("MatMul", ["x", "a_raw"], ["y"], "mm0")]
    path = str(tmp_path / "m.onnx")
    open(path, "wb").write(ow.model(inits, nodes))
    with ponnx.OnnxWeights(path) as w:
        assert w.info["a_raw"][1:] == ("float32", (5, 7)) and w.info["h_raw"][1] == "float16" and w.info["bf_raw"][1] == "bfloat16"
        np.testing.assert_array_equal(w.read("a_raw"), a)
        np.testing.assert_array_equal(w.read("b_typed"), b)
        np.testing.assert_array_equal(w.read("b_unpacked"), b)
        np.testing.assert_array_equal(w.read("h_raw"), h.astype(np.float32))
        np.testing.assert_array_equal(w.read("h_typed"), h.astype(np.float32))
        np.testing.assert_array_equal(w.read("i64_typed"), i64.astype(np.float32))
        np.testing.assert_array_equal(w.read("i64_raw"), i64.astype(np.float32))
        np.testing.assert_array_equal(w.read("d_typed"), d.astype(np.float32))
        np.testing.assert_array_equal(w.read("bf_raw"), (bf_bits.astype(np.uint32) << 16).view(np.float32))
        np.testing.assert_array_equal(w.read("ext"), ext)
        np.testing.assert_array_equal(w.read("c_out"), c)
        assert w.read("scalar").shape == () and float(w.read("scalar")) == 2.5
        assert ("MatMul", ["x", "a_raw"], ["y"]) in w.nodes() and w.nodes()[0][0] == "Constant"
        with pytest.raises(KeyError):
            w.read("nope")


def test_reader_dequantize(tmp_path, ponnx):
    rng = np.random.default_rng(1)
    w0 = rng.standard_normal((12, 8)).astype(np.float32)
    q1, s1, z1 = ow.quantize_per_axis(w0, 1)                    # per column (MatMul B, axis 1), int8 symmetric
    q2, s2, z2 = ow.quantize_per_axis(w0, 0, unsigned=True)     # per row, uint8 with zero points
    sq = rng.standard_normal((6, 6)).astype(np.float32)         # square: the axis cannot be told from the shapes
    q3, s3, z3 = ow.quantize_per_axis(sq, 0)
    qt = np.clip(np.rint(w0 / 0.05), -127, 127).astype(np.int8)
    inits = []
    for base, q, s, z in (("w1", q1, s1, z1), ("w2", q2, s2, z2), ("w3", q3, s3, z3)):
        inits += [ow.tensor(base + "_quantized", q), ow.tensor(base + "_scale", s), ow.tensor(base + "_zero_point", z)]
    inits += [ow.tensor("wt_quantized", qt), ow.tensor("wt_scale", np.float32(0.05).reshape(())), ow.tensor("wt_zero_point", np.int8(0).reshape(()))]
    inits += [ow.tensor("nozp_quantized", q1), ow.tensor("nozp_scale", s1)]
    path = str(tmp_path / "q.onnx")
    open(path, "wb").write(ow.model(inits))
    with ponnx.OnnxWeights(path) as w:
        np.testing.assert_allclose(w.dequantize("w1"), q1.astype(np.float32) * s1[None, :], rtol=0, atol=0)
        np.testing.assert_allclose(w.dequantize("w2"), (q2.astype(np.float32) - z2[:, None].astype(np.float32)) * s2[:, None], rtol=0, atol=0)
        assert np.abs(w.dequantize("w2") - w0).max() <= s2.max() * 0.51 + 1e-6
        with pytest.raises(RuntimeError, match="which axis"):
            w.dequantize("w3")
        np.testing.assert_allclose(w.dequantize("w3", axis=0), q3.astype(np.float32) * s3[:, None], rtol=0, atol=0)
        np.testing.assert_allclose(w.dequantize("wt"), qt.astype(np.float32) * np.float32(0.05), rtol=0, atol=0)
        np.testing.assert_allclose(w.dequantize("nozp"), q1.astype(np.float32) * s1[None, :], rtol=0, atol=0)


def test_reader_rejects_garbage(tmp_path, ponnx):
    p = tmp_path / "bad.onnx"
    p.write_bytes(b"\x3a\xff\xff\xff\x0f" + b"\0" * 10)      # graph field whose length runs past the end
    with pytest.raises(RuntimeError, match="Failed to read ONNX file"):
        ponnx.OnnxWeights(str(p))
    p.write_bytes(b"\x08\x08")                                  # a model with no graph
    with pytest.raises(RuntimeError, match="no graph"):
        ponnx.OnnxWeights(str(p))
    with pytest.raises(RuntimeError):
        ponnx.OnnxWeights(str(tmp_path / "missing.onnx"))


def _graph_of(name):
    if name.startswith("text."):
        return "text_conditioner.onnx"
    if name.startswith("lm."):
        return "flow_lm_main.onnx"
    if name.startswith("flow."):
        return "flow_lm_flow.onnx"
    if name.startswith("enc."):
        return "mimi_encoder.onnx"
    return "mimi_decoder.onnx"


def test_converter_round_trip(tmp_path, packs, ponnx):
    import onnx_to_ptts
    import ptts_pack as pp
    path, spec, w = packs["tiny"]
    table = pp.tensor_table(spec)
    per_file = {}
    mapping = {}
    for name, shape, _ in table:
        a = w[name]
        fn = _graph_of(name)
        ent = {"file": fn, "tensor": "t::" + name}
        if name.endswith(".w") and a.ndim == 2 and "conv" not in name and name.startswith("lm.l0."):
            # int8 MatMul weight stored [K][N] with per-column scales, like a dynamically quantised graph
            q, s, z = ow.quantize_per_axis(np.ascontiguousarray(a.T), 1)
            per_file.setdefault(fn, []).extend([ow.tensor(ent["tensor"] + "_quantized", q), ow.tensor(ent["tensor"] + "_scale", s),
                                                ow.tensor(ent["tensor"] + "_zero_point", z)])
            ent["transpose"] = True
            ent["axis"] = 1          # square matrices: the axis cannot be told from the shapes
        elif name.endswith(".w") and a.ndim == 2 and name.startswith(("lm.", "flow.")):
            per_file.setdefault(fn, []).append(ow.tensor(ent["tensor"], np.ascontiguousarray(a.T)))   # torch MatMul export: [in][out]
            ent["transpose"] = True
        elif name in ("dec.conv0.w", "enc.conv0.w"):
            cout = a.shape[0]
            k = spec.kernel
            a3 = np.ascontiguousarray(a.reshape(cout, k, -1).transpose(0, 2, 1))                      # [Cout][Cin][k]
            per_file.setdefault(fn, []).append(ow.tensor(ent["tensor"], a3, "typed"))
            ent["conv"] = True
        else:
            per_file.setdefault(fn, []).append(ow.tensor(ent["tensor"], a))
        mapping[name] = ent
    mdir = tmp_path / "onnx"
    mdir.mkdir()
    for fn, inits in per_file.items():
        (mdir / fn).write_bytes(ow.model(inits))
    assert sorted(per_file) == sorted(["text_conditioner.onnx", "flow_lm_main.onnx", "flow_lm_flow.onnx", "mimi_encoder.onnx", "mimi_decoder.onnx"])
    spec_kw = {f: getattr(spec, f) for f in pp.SPEC_INT_FIELDS + pp.SPEC_FLOAT_FIELDS}
    (tmp_path / "map.json").write_text(json.dumps({"spec": spec_kw, "tensors": mapping}))
    out = str(tmp_path / "pocket_tts.ptts")
    onnx_to_ptts.convert(str(mdir), out, str(tmp_path / "map.json"))
    spec2, w2 = pp.read_pack(out)
    assert all(getattr(spec2, f) == getattr(spec, f) for f in pp.SPEC_INT_FIELDS)
    assert all(getattr(spec2, f) == pytest.approx(getattr(spec, f)) for f in pp.SPEC_FLOAT_FIELDS)   # stored as f32
    for name, _, _ in table:
        if name.startswith("lm.l0.") and name.endswith(".w") and w[name].ndim == 2:
            s_max = np.abs(w[name]).max(1) / 127.0                      # int8 round trip: within half a step of each row's scale
            assert np.all(np.abs(w2[name] - w[name]).max(1) <= s_max * 0.51 + 1e-7), name
        else:
            np.testing.assert_array_equal(w2[name], w[name], err_msg=name)
    # the quantised matrices travel as int8: the integers and per-row scales of the source, not a second quantisation
    q8 = pp.read_pack_q8(out)
    carried = [n for n, _, _ in table if n.startswith("lm.l0.") and n.endswith(".w") and w[n].ndim == 2]
    assert sorted(q8) == sorted(carried) and carried
    for name in carried:
        q, sc, _ = ow.quantize_per_axis(np.ascontiguousarray(w[name].T), 1)
        np.testing.assert_array_equal(q8[name][0], q.T, err_msg=name)
        np.testing.assert_array_equal(q8[name][1], sc, err_msg=name)
        np.testing.assert_array_equal(w2[name], q8[name][0].astype(np.float32) * q8[name][1][:, None], err_msg=name)
    # a map that misses tensors is refused with their names
    (tmp_path / "map2.json").write_text(json.dumps({"spec": spec_kw, "tensors": {k: v for k, v in mapping.items() if not k.startswith("enc.")}}))
    with pytest.raises(SystemExit, match="no source"):
        onnx_to_ptts.convert(str(mdir), out, str(tmp_path / "map2.json"))


def test_converter_reads_the_activation_scheme_off_the_graph(tmp_path, packs, ponnx):
    """An LM graph with DynamicQuantizeLinear + MatMulInteger nodes in front of its int8 matrices (what ONNX Runtime's
    dynamic quantisation writes) gives a pack whose header asks the int8 engine for per-row uint8 activations; the flow
    blocks, int8 in the file too, are then only dequantised.  Without those nodes, or with the field given in the map,
    the header keeps what the map says."""
    import onnx_to_ptts
    import ptts_pack as pp
    import dataclasses
    spec = dataclasses.replace(pp.Spec.tiny(), flow_dim=256)     # wide enough for int8 flow blocks
    w = pp.synth_weights(spec, seed=5)
    spec_kw = {f: getattr(spec, f) for f in pp.SPEC_INT_FIELDS + pp.SPEC_FLOAT_FIELDS if f != "q8_act_scheme"}

    def export(tag, with_nodes):
        mdir = tmp_path / tag
        mdir.mkdir()
        per_file, nodes, mapping = {}, {}, {}
        for k, (name, shape, _) in enumerate(pp.tensor_table(spec)):
            a = w[name]
            fn = _graph_of(name)
            ent = {"file": fn, "tensor": f"t{k}"}
            if pp.is_int8_weight(name, spec.flow_dim, flow_blocks=True):
                q, sc, z = ow.quantize_per_axis(np.ascontiguousarray(a.T), 1)
                per_file.setdefault(fn, []).extend([ow.tensor(f"t{k}_quantized", q), ow.tensor(f"t{k}_scale", sc), ow.tensor(f"t{k}_zero_point", z)])
                ent.update(transpose=True, axis=1)
                if with_nodes:
                    nodes.setdefault(fn, []).extend([
                        ow.node("DynamicQuantizeLinear", [f"x{k}"], [f"x{k}_q", f"x{k}_s", f"x{k}_z"]),
                        ow.node("MatMulInteger", [f"x{k}_q", f"t{k}_quantized", f"x{k}_z", f"t{k}_zero_point"], [f"y{k}_i"])])
            elif a.ndim == 2 and name.endswith(".w") and name.startswith(("lm.", "flow.")):
                per_file.setdefault(fn, []).append(ow.tensor(f"t{k}", np.ascontiguousarray(a.T)))
                ent["transpose"] = True
            elif name in ("dec.conv0.w", "enc.conv0.w"):
                per_file.setdefault(fn, []).append(ow.tensor(f"t{k}", np.ascontiguousarray(a.reshape(a.shape[0], spec.kernel, -1).transpose(0, 2, 1))))
                ent["conv"] = True
            else:
                per_file.setdefault(fn, []).append(ow.tensor(f"t{k}", a))
            mapping[name] = ent
        for fn, inits in per_file.items():
            (mdir / fn).write_bytes(ow.model(inits, nodes.get(fn, ())))
        return str(mdir), mapping

    lm_q8 = sorted(n for n, _, _ in pp.tensor_table(spec) if n.startswith("lm.l") and pp.is_int8_weight(n, spec.flow_dim))
    all_q8 = sorted(n for n, _, _ in pp.tensor_table(spec) if pp.is_int8_weight(n, spec.flow_dim, flow_blocks=True))
    assert len(all_q8) > len(lm_q8) > 0
    out = str(tmp_path / "out.ptts")
    for tag, with_nodes, over, scheme, carried in [("dyn", True, {}, 1, lm_q8), ("plain", False, {}, 0, all_q8),
                                                   ("forced", True, {"q8_act_scheme": 0}, 0, all_q8)]:
        mdir, mapping = export(tag, with_nodes)
        mp = tmp_path / f"{tag}.json"
        mp.write_text(json.dumps({"spec": dict(spec_kw, **over), "tensors": mapping}))
        got_spec, _ = onnx_to_ptts.convert(mdir, out, str(mp))
        assert got_spec.q8_act_scheme == scheme == pp.read_pack(out)[0].q8_act_scheme, tag
        assert sorted(pp.read_pack_q8(out)) == carried, tag


def test_converter_auto_map(tmp_path, packs, ponnx):
    """Files whose initializers carry the pack's own names convert without a map (2-D matrices may be transposed)."""
    import onnx_to_ptts
    import ptts_pack as pp
    path, spec, w = packs["tiny"]
    mdir = tmp_path / "onnx"
    mdir.mkdir()
    per_file = {}
    for name, shape, _ in pp.tensor_table(spec):
        a = w[name]
        if a.ndim == 2 and a.shape[0] != a.shape[1] and name.startswith("flow."):
            a = np.ascontiguousarray(a.T)
        per_file.setdefault(_graph_of(name), []).append(ow.tensor(name, a))
    for fn, inits in per_file.items():
        (mdir / fn).write_bytes(ow.model(inits))
    spec_kw = {f: getattr(spec, f) for f in pp.SPEC_INT_FIELDS + pp.SPEC_FLOAT_FIELDS}
    (tmp_path / "map.json").write_text(json.dumps({"spec": spec_kw}))
    out = str(tmp_path / "auto.ptts")
    onnx_to_ptts.convert(str(mdir), out, str(tmp_path / "map.json"), auto=True)
    _, w2 = pp.read_pack(out)
    for name in w:
        np.testing.assert_array_equal(w2[name], w[name], err_msg=name)


def _exporter_files(mdir, spec, w, drop=None):
    """Writes the five graphs the way an exporter would: opaque names, initializers in another order than the nodes,
    every Linear matrix stored [in][out] behind a MatMul (square ones too), the LM layers' fused qkv as three separate
    MatMuls, the Mimi transformers' matrices as Constant nodes, conv weights [Cout][Cin][k] behind a Conv, and one norm
    gain stored once but consumed by the two layers that share its values.  Returns {pack tensor: stored names}."""
    import onnx_to_ptts
    import ptts_pack as pp
    per_file, nodes, stored = {}, {}, {}
    rng = np.random.default_rng(0)
    D = spec.d_model
    for k, (name, shape, _) in enumerate(pp.tensor_table(spec)):
        a = w[name]
        fn = onnx_to_ptts.graph_file(name)
        inits, nds = per_file.setdefault(fn, []), nodes.setdefault(fn, [])
        t = f"onnx::T_{7000 + 3 * k}"
        linear = a.ndim == 2 and name.endswith(".w") and (name.startswith(("lm.", "flow.")) or ".t" in name)
        if name == drop:
            continue
        if name == "lm.l1.ln1.g":                                   # same values as layer 0's: the file holds one tensor
            nds.append(ow.node("Mul", [f"act_{k}", stored["lm.l0.ln1.g"][0]], [f"act_{k + 1}"]))
            stored[name] = stored["lm.l0.ln1.g"]
        elif name.startswith("lm.l") and name.endswith(".qkv.w"):
            names = []
            for part in range(3):
                tp = f"{t}_{part}"
                inits.append(ow.tensor(tp, np.ascontiguousarray(a[part * D:(part + 1) * D].T)))
                nds.append(ow.node("MatMul", [f"act_{k}", tp], [f"act_{k}_{part}"]))
                names.append(tp)
            stored[name] = names
        elif linear and ".t" in name:                               # Constant node: the tensor is named by the node's output
            nds.append(ow.node("Constant", [], [t], const=ow.tensor("", np.ascontiguousarray(a.T))))
            nds.append(ow.node("MatMul", [f"act_{k}", t], [f"act_{k + 1}"]))
            stored[name] = [t]
        elif linear:
            inits.append(ow.tensor(t, np.ascontiguousarray(a.T)))
            nds.append(ow.node("MatMul", [f"act_{k}", t], [f"act_{k + 1}"]))
            stored[name] = [t]
        elif name in ("dec.conv0.w", "enc.conv0.w"):
            inits.append(ow.tensor(t, np.ascontiguousarray(a.reshape(a.shape[0], spec.kernel, -1).transpose(0, 2, 1))))
            nds.append(ow.node("Conv", [f"act_{k}", t], [f"act_{k + 1}"]))
            stored[name] = [t]
        else:
            inits.append(ow.tensor(t, a))
            nds.append(ow.node("Conv" if a.ndim == 2 else "Add", [f"act_{k}", t], [f"act_{k + 1}"]))
            stored[name] = [t]
    for fn in per_file:
        inits = per_file[fn]
        (mdir / fn).write_bytes(ow.model([inits[i] for i in rng.permutation(len(inits))], nodes[fn]))
    return stored


def test_guess_map_by_shape_and_graph_order(tmp_path, packs, ponnx):
    """The draft map for exporter-style files (see _exporter_files) places every tensor, and the pack built from it is the
    model: the order of the nodes, not of the initializers, tells layers apart; a MatMul's weight is transposed even when
    square; separate q / k / v become one concat entry; Constant-node tensors are found; a tensor consumed twice serves
    two pack tensors and the draft says so."""
    import onnx_to_ptts
    import ptts_pack as pp
    path, spec, w = packs["tiny"]
    w = dict(w)
    w["lm.l1.ln1.g"] = w["lm.l0.ln1.g"].copy()
    mdir = tmp_path / "onnx"
    mdir.mkdir()
    stored = _exporter_files(mdir, spec, w)
    spec_kw = {f: getattr(spec, f) for f in pp.SPEC_INT_FIELDS + pp.SPEC_FLOAT_FIELDS}
    tensors, unresolved = onnx_to_ptts.guess_map(str(mdir), pp.Spec(**spec_kw))
    assert not unresolved and len(tensors) == len(pp.tensor_table(spec))
    for name, names in stored.items():
        ent = tensors[name]
        got = [e["tensor"] for e in ent["concat"]] if "concat" in ent else [ent["tensor"]]
        assert got == names, name
    assert tensors["lm.l1.ln1.g"]["shared_with"] == "lm.l0.ln1.g" and "shared_with" not in tensors["lm.l0.ln1.g"]
    assert all(e.get("transpose") for e in tensors["lm.l0.qkv.w"]["concat"]) and tensors["lm.l0.out.w"].get("transpose")
    (tmp_path / "draft.json").write_text(json.dumps({"spec": spec_kw, "tensors": tensors}))
    out = str(tmp_path / "guessed.ptts")
    onnx_to_ptts.convert(str(mdir), out, str(tmp_path / "draft.json"))
    _, w2 = pp.read_pack(out)
    for name in w:
        np.testing.assert_array_equal(w2[name], w[name], err_msg=name)
    assert set(onnx_to_ptts.GRAPH_IO) == {"mimi_encoder", "text_conditioner", "flow_lm_main", "flow_lm_flow", "mimi_decoder"}


def test_guess_map_lists_what_it_cannot_place(tmp_path, packs, ponnx):
    """A file that lacks a tensor: the draft does not invent one.  The last pack tensor of that shape stays unresolved,
    with the shapes still free in the file beside it, and files that are complete are placed in full."""
    import onnx_to_ptts
    import ptts_pack as pp
    path, spec, w = packs["tiny"]
    w = dict(w)
    w["lm.l1.ln1.g"] = w["lm.l0.ln1.g"].copy()
    mdir = tmp_path / "onnx"
    mdir.mkdir()
    _exporter_files(mdir, spec, w, drop="flow.b1.fc2.w")
    tensors, unresolved = onnx_to_ptts.guess_map(str(mdir), spec)
    assert len(unresolved) == 1 and len(tensors) == len(pp.tensor_table(spec)) - 1
    u = unresolved[0]
    assert u["file"] == "flow_lm_flow.onnx" and u["shape"] == [spec.flow_dim, spec.flow_dim] and u["pack_tensor"].startswith("flow.")
    assert [spec.flow_dim, spec.flow_dim] not in u["free_shapes_in_file"]
    assert all(n in tensors for n, _, _ in pp.tensor_table(spec) if not n.startswith("flow."))


def test_guess_map_on_a_dynamically_quantised_graph(tmp_path, packs, ponnx):
    """The `_int8` graphs as ONNX Runtime's dynamic quantisation writes them: every LM-layer matrix a `_quantized` / `_scale` /
    `_zero_point` triplet stored [in][out] and consumed by a MatMulInteger behind a DynamicQuantizeLinear.  `--guess --int8`
    places the triplets by their base names (transposed, being a MatMulInteger's second input), and converting with the
    draft carries the integers into the pack and marks it for per-row uint8 activations."""
    import onnx_to_ptts
    import ptts_pack as pp
    path, spec, w = packs["tiny"]
    mdir = tmp_path / "onnx"
    mdir.mkdir()
    per_file, nodes, want_q8 = {}, {}, {}
    for k, (name, shape, _) in enumerate(pp.tensor_table(spec)):
        a = w[name]
        fn = onnx_to_ptts.graph_file(name, int8=True)
        inits, nds = per_file.setdefault(fn, []), nodes.setdefault(fn, [])
        t = f"onnx::MatMul_{900 + k}"
        if pp.is_int8_weight(name, spec.flow_dim, scheme=1):
            q, sc, z = ow.quantize_per_axis(np.ascontiguousarray(a.T), 1)
            inits += [ow.tensor(t + "_quantized", q), ow.tensor(t + "_scale", sc), ow.tensor(t + "_zero_point", z)]
            nds += [ow.node("DynamicQuantizeLinear", [f"act_{k}"], [f"act_{k}_q", f"act_{k}_s", f"act_{k}_z"]),
                    ow.node("MatMulInteger", [f"act_{k}_q", t + "_quantized", f"act_{k}_z", t + "_zero_point"], [f"acc_{k}"])]
            want_q8[name] = (np.ascontiguousarray(q.T), sc)
        elif a.ndim == 2 and name.endswith(".w") and name.startswith(("lm.", "flow.")):
            inits.append(ow.tensor(t, np.ascontiguousarray(a.T)))
            nds.append(ow.node("MatMul", [f"act_{k}", t], [f"act_{k + 1}"]))
        elif name in ("dec.conv0.w", "enc.conv0.w"):
            inits.append(ow.tensor(t, np.ascontiguousarray(a.reshape(a.shape[0], spec.kernel, -1).transpose(0, 2, 1))))
            nds.append(ow.node("Conv", [f"act_{k}", t], [f"act_{k + 1}"]))
        else:
            inits.append(ow.tensor(t, a))
            nds.append(ow.node("Conv" if a.ndim == 2 else "Add", [f"act_{k}", t], [f"act_{k + 1}"]))
    for fn in per_file:
        (mdir / fn).write_bytes(ow.model(per_file[fn][::-1], nodes[fn]))
    assert "flow_lm_main_int8.onnx" in per_file and want_q8
    tensors, unresolved = onnx_to_ptts.guess_map(str(mdir), spec, int8=True)
    assert not unresolved
    for name in want_q8:
        assert tensors[name].get("transpose") and tensors[name]["file"] == "flow_lm_main_int8.onnx", name
    spec_kw = {f: getattr(spec, f) for f in pp.SPEC_INT_FIELDS + pp.SPEC_FLOAT_FIELDS if f != "q8_act_scheme"}
    (tmp_path / "draft.json").write_text(json.dumps({"spec": spec_kw, "tensors": tensors}))
    out = str(tmp_path / "int8.ptts")
    got_spec, w2 = onnx_to_ptts.convert(str(mdir), out, str(tmp_path / "draft.json"))
    assert got_spec.q8_act_scheme == 1
    carried = pp.read_pack_q8(out)
    assert sorted(carried) == sorted(want_q8)
    for name, (q, sc) in want_q8.items():
        np.testing.assert_array_equal(carried[name][0], q, err_msg=name)
        np.testing.assert_array_equal(carried[name][1], sc, err_msg=name)
        np.testing.assert_array_equal(w2[name], q.astype(np.float32) * sc[:, None], err_msg=name)
    for name in w:
        if name not in want_q8:
            np.testing.assert_array_equal(w2[name], w[name], err_msg=name)
