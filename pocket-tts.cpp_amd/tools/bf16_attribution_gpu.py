"""Per-frame latent error of the bf16 (or int8) ENGINE against the float32 checker on unrounded weights, with the set of
matrix families held as two bf16 terms varied (POCKET_TTS_BF16_TWO_TERM), plus the K/V cache type.  Runs on the GPU box;
the inputs are those of tests/test_gpu_configs.py::test_scaled_pack_*_vs_float32_oracle (3 utterances x 5 frames, both
sides on the engine's own history).  The checker under oracle/ is used as the reference only.

    python tools/bf16_attribution_gpu.py [--scaled 1] [--precision bf16] [--out gpurun_out/attribution.txt]
"""
import argparse
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "pocket-tts.cpp_amd", "pyhost"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import oracle_py  # noqa: E402
import ptts_lib  # noqa: E402
import ptts_pack as pp  # noqa: E402

ALL = ["lm.in_proj", "lm.qkv", "lm.out", "lm.ff2", "flow.cond", "flow.ada", "flow.in_proj", "flow.final.lin"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scaled", type=int, default=1)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    ptts_lib.load()
    spec = pp.Spec.full()
    kw = dict(residual_gain=1.0 / np.sqrt(2.0 * spec.lm_layers), gate_gain=0.25) if a.scaled else {}
    path = os.path.join(tempfile.mkdtemp(prefix="ptts_attr_"), "full.ptts")
    pp.make_pack(path, spec, seed=1234, eos_bias=-20.0, **kw)
    o = oracle_py.Oracle(path)
    rng = np.random.default_rng(77)
    ids = [rng.integers(0, spec.vocab, t).astype(np.int64) for t in (32, 11, 20)]
    voices = [(rng.standard_normal((v, spec.d_model)) * 0.5).astype(np.float32) for v in (62, 30, 45)]
    F = a.frames
    lines = [f"# {a.precision} engine on MI355X against the float32 checker on the unrounded weights, full-size "
             f"{'scaled' if a.scaled else 'unscaled'} synthetic pack, 3 utterances x {F} frames, per-frame latent max-abs (and rms)"]
    print(lines[0], flush=True)

    def run(label, two_term, kv_f32=False):
        os.environ["POCKET_TTS_BF16_TWO_TERM"] = two_term
        if kv_f32:
            os.environ["POCKET_TTS_KV_F32"] = "1"
        else:
            os.environ.pop("POCKET_TTS_KV_F32", None)
        eng = ptts_lib.Engine(path, precision=a.precision, temperature=0.0, lsd_steps=10, max_frames=F)
        res = eng.generate_ids(ids, voices)
        eng.close()
        worst, sq, n = 0.0, 0.0, 0
        for i in range(3):
            st = o.lm_state(len(voices[i]) + len(ids[i]) + F + 8)
            o.lm_main(st, None, voices[i])
            o.lm_main(st, None, o.text_conditioner(ids[i]))
            cur = np.full((1, o.latent), np.nan, np.float32)
            for k in range(F):
                cond, _ = o.lm_main(st, cur, None)
                want = o.flow_euler(cond, np.zeros(o.latent, np.float32), 10)
                d = np.abs(want - res[i]["latents"][k]).astype(np.float64)
                worst = max(worst, float(d.max()))
                sq += float((d ** 2).sum())
                n += d.size
                cur = res[i]["latents"][k:k + 1]
        line = f"{label:72s} max-abs {worst:.3e}   rms {np.sqrt(sq / n):.3e}"
        print(line, flush=True)
        lines.append(line)

    run("no family two-term (every matrix one bf16 term), K/V bf16", "none")
    run("no family two-term, K/V f32", "none", kv_f32=True)
    for fam in ALL:
        run(f"two-term: {fam}", fam)
    run("two-term: lm.in_proj,flow.cond,flow.in_proj,flow.final.lin", "lm.in_proj,flow.cond,flow.in_proj,flow.final.lin")
    run("two-term: ... + flow.ada", "lm.in_proj,flow.cond,flow.in_proj,flow.final.lin,flow.ada")
    run("two-term: ... + lm.ff2", "lm.in_proj,flow.cond,flow.in_proj,flow.final.lin,lm.ff2")
    run("two-term: ... + flow.ada + lm.ff2", "lm.in_proj,flow.cond,flow.in_proj,flow.final.lin,flow.ada,lm.ff2")
    run("two-term: default (= ... + flow.ada)", "default")
    run("two-term: default, K/V f32", "default", kv_f32=True)
    run("two-term: all eight families", ",".join(ALL))
    os.environ.pop("POCKET_TTS_BF16_TWO_TERM", None)
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
