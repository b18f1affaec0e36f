"""How the bf16 engine's distance from the float32 checker (scaled full-size pack, per-frame latent max-abs and rms on the
engine's own history, as tests/test_gpu_configs.py::test_scaled_pack_bf16_vs_float32_oracle measures it) moves with the
seed of the synthetic weights and of the inputs.  Runs on the GPU box; the checker under oracle/ is the reference only.

    python tools/bf16_seed_sweep.py [--seeds 1234,1,2,3] [--frames 5] [--two-term lm.in_proj,...] [--out gpurun_out/bf16_seed_sweep.txt]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="1234,1,2,3")
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--two-term", default="", help="POCKET_TTS_BF16_TWO_TERM for the engine (default: the engine's own default set)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    if a.two_term:
        os.environ["POCKET_TTS_BF16_TWO_TERM"] = a.two_term
    ptts_lib.load()
    spec = pp.Spec.full()
    F = a.frames
    lines = [f"# {a.precision} engine against the float32 checker on the unrounded weights, full-size scaled synthetic pack, 3 utterances x {F} frames per seed, two-term families: {a.two_term or 'default'}"]
    print(lines[0], flush=True)
    tmp = tempfile.mkdtemp(prefix="ptts_sweep_")
    for seed in [int(x) for x in a.seeds.split(",")]:
        path = os.path.join(tmp, f"full_{seed}.ptts")
        pp.make_pack(path, spec, seed=seed, eos_bias=-20.0, residual_gain=1.0 / np.sqrt(2.0 * spec.lm_layers), gate_gain=0.25)
        o = oracle_py.Oracle(path)
        rng = np.random.default_rng(77 + seed)
        ids = [rng.integers(0, spec.vocab, t).astype(np.int64) for t in (32, 11, 20)]
        voices = [(rng.standard_normal((v, spec.d_model)) * 0.5).astype(np.float32) for v in (62, 30, 45)]
        eng = ptts_lib.Engine(path, precision=a.precision, temperature=0.0, lsd_steps=10, max_frames=F)
        res = eng.generate_ids(ids, voices)
        eng.close()
        worst, sq, n, amp = 0.0, 0.0, 0, 0.0
        for i in range(3):
            st = o.lm_state(len(voices[i]) + len(ids[i]) + F + 8)
            o.lm_main(st, None, voices[i])
            o.lm_main(st, None, o.text_conditioner(ids[i]))
            cur = np.full((1, o.latent), np.nan, np.float32)
            for k in range(F):
                cond, _ = o.lm_main(st, cur, None)
                x = o.flow_euler(cond, np.zeros(o.latent, np.float32), 10)
                d = res[i]["latents"][k] - x
                worst = max(worst, float(np.abs(d).max()))
                amp = max(amp, float(np.abs(x).max()))
                sq += float((d * d).sum())
                n += d.size
                cur = res[i]["latents"][k].reshape(1, -1)
        o.close()
        os.remove(path)
        line = f"weights seed {seed:5d}, inputs seed {77 + seed:5d}: max-abs {worst:.3e}   rms {np.sqrt(sq / n):.3e}   largest latent {amp:.2f}"
        print(line, flush=True)
        lines.append(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
