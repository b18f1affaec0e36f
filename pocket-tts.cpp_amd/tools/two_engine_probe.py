"""Tuning aid: two engines on one GPU, each running a staged batch of 64 from its own thread; prints the aggregate RTFx."""
import os, sys, time, threading, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyhost"))
sys.path.insert(0, os.path.dirname(ROOT))
import numpy as np
import ptts_pack as pp, ptts_lib

def main():
    n_eng = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    spec = pp.Spec.full()
    path = os.path.join(tempfile.gettempdir(), "probe_full.ptts")
    pp.make_pack(path, spec, seed=1234, eos_bias=-20.0)
    engs = [ptts_lib.Engine(path, precision="bf16", temperature=0.7, max_frames=125, seed=5 + i) for i in range(n_eng)]
    rng = np.random.default_rng(0)
    staged = []
    for e in engs:
        ids = [rng.integers(1, spec.vocab, size=30).astype(np.int64) for _ in range(B)]
        voices = [rng.standard_normal((62, spec.d_model)).astype(np.float32) * 0.1 for _ in range(B)]
        st = e.stage_ids(ids, voices)
        e.run_staged(st)
        staged.append(st)
    frame_s = spec.frame_size / spec.sample_rate
    tot = [0] * n_eng
    def work(i):
        for _ in range(3):
            tot[i] += sum(engs[i].run_staged(staged[i]))
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(n_eng)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print("engines", n_eng, "B", B, "RTFx", round(sum(tot) * frame_s / dt, 1), "ms/step", round(dt / 3 * 1e3, 1))

main()
