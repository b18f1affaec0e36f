"""Which part of the int8 scheme owns the int8 engine's distance from the float32 model?  CPU only (uses the checker
under oracle/: a diagnostic beside the tests, never part of the product path).

The int8 engine = the bf16 engine with the LM layers' qkv / out / ff1 / ff2 (and, with POCKET_TTS_Q8_FLOW=1, the flow blocks'
fc1 / fc2) as int8 rows (one scale per output row) multiplied with activations quantised per row and 64-wide block.  The float32 checker can
apply that scheme to any subset of the six families and to either side alone (weights only / activations only), so each
row below is the float32 model with ONE such quantity quantised, against the float32 model, per-frame latents on the same
history (the inputs of tests/test_gpu_configs.py::test_scaled_pack_int8_vs_float32_oracle).

    python tools/int8_attribution.py [--scaled 0|1] [--act block|dyn] [--out profiles/r03_int8_attribution.txt]

--act dyn: the second activation scheme (per row, uint8 with a zero point over the whole row, as ONNX
DynamicQuantizeLinear does it; LM layers only, the flow blocks stay bf16 in that scheme).
"""
import argparse
import ctypes as C
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "pocket-tts.cpp_amd", "pyhost"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import oracle_py  # noqa: E402
import ptts_pack as pp  # noqa: E402

BITS = {"lm.qkv": 1, "lm.out": 2, "lm.ff1": 4, "lm.ff2": 8, "flow.fc1": 16, "flow.fc2": 32}


def frames_of(o, ids, voice, history, F, lsd):
    st = o.lm_state(len(voice) + len(ids) + F + 8)
    o.lm_main(st, None, voice)
    o.lm_main(st, None, o.text_conditioner(ids))
    cur = np.full((1, o.latent), np.nan, np.float32)
    out, conds = [], []
    for k in range(F):
        cond, _ = o.lm_main(st, cur, None)
        x = o.flow_euler(cond, np.zeros(o.latent, np.float32), lsd)
        out.append(x)
        conds.append(cond)
        cur = (x if history is None else history[k]).reshape(1, -1)
    return np.array(out), np.array(conds)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scaled", type=int, default=1)
    ap.add_argument("--out", default="")
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--act", choices=["block", "dyn"], default="block")
    a = ap.parse_args()
    oracle_py.build()
    L = oracle_py.lib()
    L.ptts_oracle_enable_q8_some.argtypes = [C.c_void_p, C.c_int]
    L.ptts_oracle_set_q8_parts.argtypes = [C.c_int]
    L.ptts_oracle_set_lm_kv_bf16.argtypes = [C.c_int]
    L.ptts_oracle_set_q8_act.argtypes = [C.c_int]
    dyn = a.act == "dyn"
    spec = pp.Spec.full()
    if dyn:
        import dataclasses
        spec = dataclasses.replace(spec, q8_act_scheme=1)
    kw = dict(residual_gain=1.0 / np.sqrt(2.0 * spec.lm_layers), gate_gain=0.25) if a.scaled else {}
    w = pp.synth_weights(spec, seed=1234, eos_bias=-20.0, **kw)
    rng = np.random.default_rng(77)
    ids = [rng.integers(0, spec.vocab, t).astype(np.int64) for t in (32, 11, 20)]
    voices = [(rng.standard_normal((v, spec.d_model)) * 0.5).astype(np.float32) for v in (62, 30, 45)]
    F, lsd = a.frames, 10
    tmp = tempfile.mkdtemp(prefix="ptts_attr8_")
    path = os.path.join(tmp, "f32.ptts")
    pp.write_pack(path, spec, w)
    path_e = os.path.join(tmp, "engine.ptts")
    pp.write_pack(path_e, spec, pp.int8_model(w, scheme=spec.q8_act_scheme))   # what the int8 engine holds beside its int8 matrices
    path_ef = os.path.join(tmp, "engine_flow.ptts")
    pp.write_pack(path_ef, spec, pp.int8_model(w, scheme=spec.q8_act_scheme, flow_blocks=True))   # ... with POCKET_TTS_Q8_FLOW=1

    def run(pack, mask, parts, kv=False, base=None):
        o = oracle_py.Oracle(pack)
        if mask:
            L.ptts_oracle_set_q8_act(1 if dyn else 0)
            L.ptts_oracle_enable_q8_some(o.h, mask)
            L.ptts_oracle_set_q8_act(0)
        L.ptts_oracle_set_q8_parts(parts)
        L.ptts_oracle_set_lm_kv_bf16(1 if kv else 0)
        res = [frames_of(o, ids[i], voices[i], None if base is None else base[i][0], F, lsd) for i in range(3)]
        L.ptts_oracle_set_q8_parts(3)
        L.ptts_oracle_set_lm_kv_bf16(0)
        o.close()
        return res

    base = run(path, 0, 3)
    lines = []

    def report(name, res):
        lat = max(float(np.abs(res[i][0] - base[i][0]).max()) for i in range(3))
        cond = max(float(np.abs(res[i][1] - base[i][1]).max()) for i in range(3))
        line = f"{name:58s} latent max-abs {lat:.3e}   LM conditioning max-abs {cond:.3e}"
        print(line, flush=True)
        lines.append(line)

    hdr = (f"# int8 attribution, full-size {'scaled' if a.scaled else 'unscaled'} synthetic pack (seed 1234), float32 checker, "
           f"3 utterances x {F} frames, same history on both sides\n"
           f"# each row: the float32 model with the named quantity in the engine's int8 scheme, against the float32 model")
    print(hdr)
    lines.append(hdr)
    if dyn:
        lines[-1] += "\n# activation scheme: per row, uint8 with one scale and zero point over the whole row (LM layers only)"
        for name, bit in list(BITS.items())[:4]:
            report(f"{name}: uint8 activation rows only", run(path, bit, 2, base=base))
        report("LM families: int8 weight rows only", run(path, 15, 1, base=base))
        report("LM families: uint8 activation rows only", run(path, 15, 2, base=base))
        report("LM families: weights and activations", run(path, 15, 3, base=base))
        report("the int8 engine's whole model (+ bf16 rest, bf16 K/V)", run(path_e, 15, 3, kv=True, base=base))
    else:
        for name, bit in BITS.items():
            report(f"{name}: int8 weight rows only", run(path, bit, 1, base=base))
            report(f"{name}: int8 activation blocks only", run(path, bit, 2, base=base))
        report("all six families: int8 weight rows only", run(path, 63, 1, base=base))
        report("all six families: int8 activation blocks only", run(path, 63, 2, base=base))
        report("all six families: weights and activations", run(path, 63, 3, base=base))
        report("LM families only (flow blocks float32): both sides", run(path, 15, 3, base=base))
        report("flow blocks only (LM float32): both sides", run(path, 48, 3, base=base))
        report("the int8 engine's whole model (int8 LM layers + bf16 rest, bf16 K/V)", run(path_e, 15, 3, kv=True, base=base))
        report("... with int8 flow blocks as well (POCKET_TTS_Q8_FLOW=1)", run(path_ef, 63, 3, kv=True, base=base))
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")
    for p in (path, path_e, path_ef):
        os.remove(p)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
