#!/usr/bin/env python3
"""First-chunk latency of the streaming call, one utterance, as a function of the chunk size: the slope is the cost
of one frame, the intercept the prefill plus the chunk decode and its copy.  Prints one JSON line."""
import argparse, json, os, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pocket-tts.cpp_amd", "pyhost"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--chunks", default="1,5,10")
    ap.add_argument("--temperature", type=float, default=0.0)
    args = ap.parse_args()
    import ptts_pack as pp
    import ptts_lib
    spec = pp.Spec.full()
    path = os.path.join(tempfile.gettempdir(), f"ptts_bench_full_{os.getuid()}_0.ptts")
    pp.make_pack(path, spec, seed=1234, eos_bias=-20.0)
    ids = np.random.default_rng(1).integers(0, spec.vocab, 32).astype(np.int64)
    voice = (np.random.default_rng(2).standard_normal((62, spec.d_model)) * 0.5).astype(np.float32)
    out = {}
    for c in [int(x) for x in args.chunks.split(",")]:
        eng = ptts_lib.Engine(path, precision=args.precision, temperature=args.temperature, lsd_steps=10, max_frames=max(c, 10), seed=7)
        ts = []
        for i in range(args.runs + 2):
            _, _, t = eng.streaming_ids(ids, voice, chunk_frames=c)
            if i >= 2:
                ts.append(t * 1e3)
        eng.close()
        ts.sort()
        out[f"chunk{c}_ms"] = round(ts[len(ts) // 2], 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
