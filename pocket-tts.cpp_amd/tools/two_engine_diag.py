"""Diagnostic: ragged utterances queued on one handle (set POCKET_TTS_ENGINES_PER_GPU for several slot tables on
the GPU); prints every utterance whose latents or EOS logits differ from the oracle's teacher-forced replay, and
the step at which the difference starts."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for sub in ("tests", os.path.join("pocket-tts.cpp_amd", "pyhost"), "oracle"):
    sys.path.insert(0, os.path.join(ROOT, sub))


def main():
    import oracle_py
    import ptts_pack as pp
    import ptts_lib as ptts
    from conftest import oracle_replay
    oracle_py.build()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    F = 150
    path = os.path.join(tempfile.mkdtemp(), "tiny_spread.ptts")
    spec = pp.Spec.tiny()
    pp.make_pack(path, spec, seed=7, eos_bias=-6.6, layer_scale=0.2)
    eng = ptts.Engine(path, precision="fp32", temperature=0.0, lsd_steps=4, max_frames=F)
    o = oracle_py.Oracle(path)
    rng = np.random.default_rng(2024)
    ids = [rng.integers(0, spec.vocab, int(rng.integers(1, 18))).astype(np.int64) for _ in range(n)]
    voices = [(rng.standard_normal((int(rng.integers(1, 12)), spec.d_model)) * 0.5).astype(np.float32) for _ in range(n)]
    res = eng.generate_ids(ids, voices, queued=True)
    bad = 0
    for i in range(n):
        lat, eos, frames = oracle_replay(o, ids[i], voices[i], res[i]["latents"], 4, F)
        m = min(len(eos), len(res[i]["eos"]))
        de = np.abs(np.asarray(res[i]["eos"][:m]) - np.asarray(eos[:m]))
        k = min(frames, res[i]["frames"])
        dl = np.abs(res[i]["latents"][:k] - lat[:k]).reshape(k, -1).max(axis=1) if k else np.zeros(0)
        if res[i]["frames"] != frames or (de > 1e-3).any() or (dl > 1e-3).any():
            bad += 1
            fe = int(np.argmax(de > 1e-3)) if (de > 1e-3).any() else -1
            fl = int(np.argmax(dl > 1e-3)) if (dl > 1e-3).any() else -1
            print(f"utt {i}: T={len(ids[i])} V={len(voices[i])} frames gpu/oracle {res[i]['frames']}/{frames}; first eos diff at step {fe} "
                  f"(max {de.max():.3g}), first latent diff at frame {fl} (max {dl.max() if k else 0:.3g})", flush=True)
            np.set_printoptions(precision=3, linewidth=200)
            print("  per-frame latent diff against the replay:", dl[:40])
            print("  per-step eos diff against the replay:", de[:40])
            alone = eng.generate_ids([ids[i]], [voices[i]])[0]
            ka = min(k, alone["frames"])
            da = np.abs(res[i]["latents"][:ka] - alone["latents"][:ka]).reshape(ka, -1).max(axis=1)
            print(f"  alone: {alone['frames']} frames; per-frame latent diff queued vs alone:", da[:40])
            print("  queued frame 0:", res[i]["latents"][0][:8], " alone frame 0:", alone["latents"][0][:8], " replay frame 0:", lat[0][:8])
    print(f"{bad} of {n} utterances differ; stats {eng.scheduler_stats()}")
    eng.close()


if __name__ == "__main__":
    main()
