"""Real-model ingestion on the GPU: ONNX files (made by tests/onnx_writer.py in the shapes exporters produce: renamed
initializers, MatMul weights stored [in][out], Conv1d weights [Cout][Cin][k], int8 graphs with `_quantized` / `_scale` /
`_zero_point` triplets) -> tools/onnx_to_ptts.py -> weight pack -> engine, against the engine on the pack written directly.
The reference opens the same five files per precision (src/pocket_tts.cpp:90-96)."""
import json
import os
import sys

import numpy as np
import pytest

import onnx_writer as ow

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pocket-tts.cpp_amd", "tools"))

pytestmark = pytest.mark.gpu


def _graph_of(name, int8=False):
    if name.startswith("enc."):
        return "mimi_encoder.onnx"
    if name.startswith("text."):
        return "text_conditioner.onnx"
    if name.startswith("lm."):
        return "flow_lm_main_int8.onnx" if int8 else "flow_lm_main.onnx"
    if name.startswith("flow."):
        return "flow_lm_flow_int8.onnx" if int8 else "flow_lm_flow.onnx"
    return "mimi_decoder_int8.onnx" if int8 else "mimi_decoder.onnx"


def _export(tmp, spec, w, int8):
    """Writes the five graphs of one precision and their map; returns (models dir, map path, {name: (q, scale)})."""
    import ptts_pack as pp
    mdir = tmp / ("onnx_i8" if int8 else "onnx_f32")
    mdir.mkdir()
    per_file, mapping, q8 = {}, {}, {}
    for k, (name, shape, _) in enumerate(pp.tensor_table(spec)):
        a = w[name]
        fn = _graph_of(name, int8)
        ent = {"file": fn, "tensor": f"onnx::MatMul_{1000 + k}" if a.ndim == 2 else f"model.{name.replace('.', '_')}"}
        if int8 and pp.is_int8_weight(name, spec.flow_dim):
            # dynamically quantised MatMul: int8 [in][out], one scale per output column, chosen so that |q| <= 100 (a second
            # quantisation of the dequantised matrix would pick other integers)
            sc = (np.abs(a).max(axis=1) / 100.0).astype(np.float32)
            q = np.clip(np.rint(a / sc[:, None]), -127, 127).astype(np.int8)
            per_file.setdefault(fn, []).extend([ow.tensor(ent["tensor"] + "_quantized", np.ascontiguousarray(q.T)),
                                                ow.tensor(ent["tensor"] + "_scale", sc),
                                                ow.tensor(ent["tensor"] + "_zero_point", np.zeros(len(sc), np.int8))])
            ent["transpose"] = True
            ent["axis"] = 1
            q8[name] = (q, sc)
        elif a.ndim == 2 and name.endswith(".w") and name.startswith(("lm.", "flow.")):
            per_file.setdefault(fn, []).append(ow.tensor(ent["tensor"], np.ascontiguousarray(a.T)))
            ent["transpose"] = True
        elif name in ("dec.conv0.w", "enc.conv0.w"):
            a3 = np.ascontiguousarray(a.reshape(a.shape[0], spec.kernel, -1).transpose(0, 2, 1))
            per_file.setdefault(fn, []).append(ow.tensor(ent["tensor"], a3, "typed"))
            ent["conv"] = True
        else:
            per_file.setdefault(fn, []).append(ow.tensor(ent["tensor"], a))
        mapping[name] = ent
    for fn, inits in per_file.items():
        (mdir / fn).write_bytes(ow.model(inits))
    spec_kw = {f: getattr(spec, f) for f in pp.SPEC_INT_FIELDS + pp.SPEC_FLOAT_FIELDS}
    mp = tmp / ("map_i8.json" if int8 else "map_f32.json")
    mp.write_text(json.dumps({"spec": spec_kw, "tensors": mapping}))
    return str(mdir), str(mp), q8


def _run(ptts, path, precision, spec, env=None):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        eng = ptts.Engine(path, precision=precision, temperature=0.0, lsd_steps=4, max_frames=6)
        rng = np.random.default_rng(3)
        ids = [rng.integers(0, spec.vocab, n).astype(np.int64) for n in (9, 5)]
        voices = [(rng.standard_normal((v, spec.d_model)) * 0.5).astype(np.float32) for v in (6, 11)]
        res = eng.generate_ids(ids, voices)
        clip = (0.3 * np.sin(np.arange(9600) / 7.0)).astype(np.float32)
        emb = eng.mimi_encode(clip)
        eng.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return res, emb


def test_fp32_pack_from_onnx_equals_direct_pack(tmp_path, packs):
    import onnx_to_ptts
    import ptts_lib as ptts
    path, spec, w = packs["small"]
    mdir, mp, _ = _export(tmp_path, spec, w, int8=False)
    out = str(tmp_path / "from_onnx.ptts")
    onnx_to_ptts.convert(mdir, out, mp)
    a, ea = _run(ptts, path, "fp32", spec)
    b, eb = _run(ptts, out, "fp32", spec)
    for x, y in zip(a, b):
        assert x["frames"] == y["frames"] == 6
        np.testing.assert_array_equal(x["latents"], y["latents"])
        np.testing.assert_array_equal(x["audio"], y["audio"])
    np.testing.assert_array_equal(ea, eb)


def test_int8_triplets_are_carried_into_the_engine(tmp_path, packs):
    import onnx_to_ptts
    import ptts_lib as ptts
    import ptts_pack as pp
    path, spec, w = packs["small"]
    mdir, mp, q8 = _export(tmp_path, spec, w, int8=True)
    assert q8, "the small model has int8-eligible matrices"
    p1 = str(tmp_path / "from_onnx_i8.ptts")
    onnx_to_ptts.convert(mdir, p1, mp)
    carried = pp.read_pack_q8(p1)
    assert sorted(carried) == sorted(q8)
    for name, (q, sc) in q8.items():
        np.testing.assert_array_equal(carried[name][0], q, err_msg=name)
        np.testing.assert_array_equal(carried[name][1], sc, err_msg=name)
    # the same model written directly: dequantised f32 tensors plus the int8 forms / without them
    wd = dict(w)
    for name, (q, sc) in q8.items():
        wd[name] = (q.astype(np.float32) * sc[:, None]).astype(np.float32)
    p2, p3 = str(tmp_path / "direct_i8.ptts"), str(tmp_path / "direct_deq.ptts")
    pp.write_pack(p2, spec, wd, q8=q8)
    pp.write_pack(p3, spec, wd)
    r1, _ = _run(ptts, p1, "int8", spec)
    r2, _ = _run(ptts, p2, "int8", spec)
    r3, _ = _run(ptts, p3, "int8", spec)
    r4, _ = _run(ptts, p1, "int8", spec, env={"POCKET_TTS_REQUANTIZE": "1"})
    differs = False
    for a, b, c, d in zip(r1, r2, r3, r4):
        np.testing.assert_array_equal(a["latents"], b["latents"])      # converter == direct: the engine holds the same integers
        np.testing.assert_array_equal(c["latents"], d["latents"])      # the switch reproduces "quantise the f32 tensor again"
        differs |= not np.array_equal(a["latents"], c["latents"])
        assert np.abs(a["latents"][0] - c["latents"][0]).max() < 0.3   # first frame only: later frames drift apart freely
    assert differs, "re-quantising picked the same integers: the test does not show that the triplets are carried"
    # the int8 kernel on carried values against a numpy restatement of the scheme: integer block sums, f32 scaling
    name = sorted(q8)[0]
    q, sc = q8[name]
    eng = ptts.Engine(p1, precision="int8", temperature=0.0, lsd_steps=1, max_frames=1)
    rng = np.random.default_rng(5)
    M, (N, K) = 7, q.shape
    A = (rng.standard_normal((M, K)) * rng.uniform(0.1, 3.0, (M, 1))).astype(np.float32)
    Ab = A.reshape(M, K // 64, 64)
    ma = np.abs(Ab).max(axis=2)
    inva = np.where(ma > 0, np.float32(127.0) / np.where(ma > 0, ma, 1), 0).astype(np.float32)
    qa = np.clip(np.rint(Ab * inva[:, :, None]), -127, 127).astype(np.int32)
    da = (ma / np.float32(127.0)).astype(np.float32)
    acc = np.zeros((M, N), np.float32)
    qw = q.astype(np.int32)
    for kb in range(K // 64):
        acc = (acc + (qa[:, kb, :] @ qw[:, kb * 64:(kb + 1) * 64].T).astype(np.float32) * da[:, kb:kb + 1]).astype(np.float32)
    want = acc * sc[None, :]
    got = eng.q8_linear_q(A, q, sc)
    assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
    eng.close()
