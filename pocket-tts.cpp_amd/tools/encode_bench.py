#!/usr/bin/env python3
"""Voice-clone encode throughput: the Mimi encoder over N seeded clips of 5 s (the reference cuts every voice clip to
120000 samples before encoding, src/pocket_tts.cpp:256-259), --batch clips per call (1 = the drop-in call's form).
Host samples in, host embedding rows out.  Prints one JSON line."""
import argparse, json, os, sys, tempfile, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pocket-tts.cpp_amd", "pyhost"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=64)
    ap.add_argument("--samples", type=int, default=120000)
    ap.add_argument("--batch", type=int, default=32, help="clips per call (the library stacks them 8 at a time)")
    args = ap.parse_args()
    import ptts_pack as pp
    import ptts_lib
    spec = pp.Spec.full()
    path = os.path.join(tempfile.gettempdir(), f"ptts_bench_full_{os.getuid()}_0.ptts")
    pp.make_pack(path, spec, seed=1234, eos_bias=-20.0)
    eng = ptts_lib.Engine(path, precision="bf16", max_frames=10)
    rng = np.random.default_rng(5)
    t = np.arange(args.samples) / 24000.0
    clips = [(0.2 * np.sin(2 * np.pi * (110 + 7 * i) * t) + 0.05 * rng.standard_normal(args.samples)).astype(np.float32) for i in range(4)]
    group = [clips[i % 4] for i in range(args.batch)]
    for i in range(2):
        eng.mimi_encode(clips[i]) if args.batch == 1 else eng.mimi_encode_batch(group)
    t0 = time.perf_counter()
    for i in range(0, args.clips, args.batch):
        out = eng.mimi_encode(clips[i % 4]) if args.batch == 1 else eng.mimi_encode_batch(group)[-1]
    dt = time.perf_counter() - t0
    eng.close()
    print(json.dumps(dict(metric="voice encode throughput", clips=args.clips, clips_per_call=args.batch, samples_per_clip=args.samples, rows_out=int(out.shape[0]),
                          ms_per_clip=round(dt / args.clips * 1e3, 3), clips_per_s=round(args.clips / dt, 1),
                          audio_s_per_s=round(args.clips * args.samples / 24000.0 / dt, 1))))


if __name__ == "__main__":
    main()
