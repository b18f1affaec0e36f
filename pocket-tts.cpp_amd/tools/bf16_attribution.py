"""Which stored-in-bf16 quantity owns the bf16 engine's distance from the float32 model?  CPU only (uses the checker
under oracle/, so this is a diagnostic beside the tests, never part of the product path).

The bf16 engine differs from float32 arithmetic in exactly two places: matrix weights are stored as bf16, and the LM's
K/V rows are stored as bf16.  Everything else (activations as two bf16 terms, f32 accumulation) is float-grade.  So the
float32 checker evaluating a pack in which ONE family of matrices is rounded to bf16 (or with its K/V rows rounded)
reproduces that family's share of the engine's error.  Per-frame latents are compared with both sides on the same
history (the float32 model's own frames), 3 utterances x 5 frames, the inputs of
tests/test_gpu_configs.py::test_scaled_pack_*_vs_float32_oracle.

    python tools/bf16_attribution.py [--scaled 0|1] [--out profiles/r03_bf16_attribution.txt]
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

FAMILIES = {
    "lm.in_proj": lambda k: k == "lm.in_proj.w",
    "lm.qkv": lambda k: k.startswith("lm.l") and k.endswith(".qkv.w"),
    "lm.out": lambda k: k.startswith("lm.l") and k.endswith(".out.w"),
    "lm.ff1": lambda k: k.startswith("lm.l") and k.endswith(".ff1.w"),
    "lm.ff2": lambda k: k.startswith("lm.l") and k.endswith(".ff2.w"),
    "flow.cond": lambda k: k == "flow.cond.w",
    "flow.ada": lambda k: k.startswith("flow.") and k.endswith(".ada.w"),
    "flow.in_proj": lambda k: k == "flow.in_proj.w",
    "flow.fc1": lambda k: k.startswith("flow.b") and k.endswith(".fc1.w"),
    "flow.fc2": lambda k: k.startswith("flow.b") and k.endswith(".fc2.w"),
    "flow.final.lin": lambda k: k == "flow.final.lin.w",
}


def rounded_pack(w, select):
    return {k: (pp.bf16_round(v) if select(k) else v) for k, v in w.items()}


def frames_of(o, ids, voice, history, F, lsd):
    """Per-frame latents of oracle `o`, frame k computed from history[0..k-1] (history None: free running)."""
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
    ap.add_argument("--cand", action="append", default=[], help="extra candidate: families joined by '+' kept as two bf16 terms")
    ap.add_argument("--only-cand", type=int, default=0, help="1: skip the per-family rows and the built-in candidates")
    a = ap.parse_args()
    oracle_py.build()
    L = oracle_py.lib()
    L.ptts_oracle_set_lm_kv_bf16.argtypes = [C.c_int]
    spec = pp.Spec.full()
    kw = dict(residual_gain=1.0 / np.sqrt(2.0 * spec.lm_layers), gate_gain=0.25) if a.scaled else {}
    w = pp.synth_weights(spec, seed=1234, eos_bias=-20.0, **kw)
    rng = np.random.default_rng(77)
    ids = [rng.integers(0, spec.vocab, t).astype(np.int64) for t in (32, 11, 20)]
    voices = [(rng.standard_normal((v, spec.d_model)) * 0.5).astype(np.float32) for v in (62, 30, 45)]
    F, lsd = a.frames, 10
    tmp = tempfile.mkdtemp(prefix="ptts_attr_")
    path = os.path.join(tmp, "v.ptts")

    def run(weights, kv=False, base=None):
        pp.write_pack(path, spec, weights)
        o = oracle_py.Oracle(path)
        L.ptts_oracle_set_lm_kv_bf16(1 if kv else 0)
        res = [frames_of(o, ids[i], voices[i], None if base is None else base[i][0], F, lsd) for i in range(3)]
        L.ptts_oracle_set_lm_kv_bf16(0)
        o.close()
        return res

    base = run(w)
    lines = []

    def report(name, res):
        lat = max(float(np.abs(res[i][0] - base[i][0]).max()) for i in range(3))
        cond = max(float(np.abs(res[i][1] - base[i][1]).max()) for i in range(3))
        line = f"{name:34s} latent max-abs {lat:.3e}   LM conditioning max-abs {cond:.3e}"
        print(line, flush=True)
        lines.append(line)
        return lat

    hdr = (f"# bf16 attribution, full-size {'scaled' if a.scaled else 'unscaled'} synthetic pack (seed 1234), float32 checker, "
           f"3 utterances x {F} frames, same history on both sides\n"
           f"# each row: the float32 model with ONE quantity stored as bf16, against the float32 model")
    print(hdr)
    lines.append(hdr)
    allw = lambda k: any(f(k) for f in FAMILIES.values())  # noqa: E731
    for name, sel in FAMILIES.items():
        if not a.only_cand:
            report("only " + name + " in bf16", run(rounded_pack(w, sel), base=base))
    if not a.only_cand:
        report("only LM K/V rows in bf16", run(w, kv=True, base=base))
    report("all matrices in bf16, K/V f32", run(rounded_pack(w, allw), base=base))
    report("all matrices + K/V in bf16", run(rounded_pack(w, allw), kv=True, base=base))
    lines.append("# candidates: the matrices named kept as two bf16 terms (float-grade), the rest and K/V as above")
    lmsel = lambda k: k.startswith("lm.")  # noqa: E731
    builtin = () if a.only_cand else (("lm.qkv",), ("lm.ff1",), ("lm.ff2",), ("lm.qkv", "lm.ff1"), ("lm.qkv", "lm.ff1", "lm.ff2"),
                 ("lm.qkv", "lm.out", "lm.ff1", "lm.ff2", "lm.in_proj"), ("flow.ada",), ("flow.fc1", "flow.fc2"),
                 ("flow.cond", "flow.ada"), ("flow.cond", "flow.ada", "flow.fc1", "flow.fc2", "flow.in_proj", "flow.final.lin"))
    for keep in builtin + tuple(tuple(c.split("+")) for c in a.cand):
        sel = lambda k, keep=keep: allw(k) and not any(FAMILIES[f](k) for f in keep)  # noqa: E731
        report("two-term: " + "+".join(keep) + "; K/V bf16", run(rounded_pack(w, sel), kv=True, base=base))
        report("two-term: " + "+".join(keep) + "; K/V f32", run(rounded_pack(w, sel), kv=False, base=base))
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")
    os.remove(path)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
