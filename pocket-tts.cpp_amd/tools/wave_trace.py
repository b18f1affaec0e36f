"""Tuning aid: 192 queued utterances x 125 frames on one engine with POCKET_TTS_TRACE=1; the trace (stderr) shows what happens
between one table's worth of utterances and the next."""
import os, sys, time, tempfile
os.environ.setdefault("POCKET_TTS_TRACE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyhost"))
import numpy as np
import ptts_pack as pp, ptts_lib

spec = pp.Spec.full()
path = os.path.join(tempfile.gettempdir(), "probe_full.ptts")
pp.make_pack(path, spec, seed=1234, eos_bias=-20.0)
eng = ptts_lib.Engine(path, precision=sys.argv[1] if len(sys.argv) > 1 else "int8", temperature=0.7, max_frames=125, seed=5)
rng = np.random.default_rng(0)
N = 192
ids = [rng.integers(1, spec.vocab, size=int(rng.integers(8, 41))).astype(np.int64) for _ in range(N)]
voices = [(rng.standard_normal((int(rng.integers(20, 63)), spec.d_model)) * 0.1).astype(np.float32) for _ in range(N)]
eng.generate_ids(ids[:64], voices[:64], queued=True, want_latents=False, want_audio=False)
sys.stderr.write("==== timed call\n")
t0 = time.perf_counter()
eng.generate_ids(ids, voices, queued=True, want_latents=False, want_audio=False)
dt = time.perf_counter() - t0
sys.stderr.write("==== end\n")
print("call ms", round(dt * 1e3, 2), "RTFx", round(N * 125 * 0.08 / dt, 1))
eng.close()
