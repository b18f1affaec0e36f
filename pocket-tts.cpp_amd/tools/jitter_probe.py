"""Tuning aid: one utterance streams (5-frame chunks, 125 frames) while other callers keep arriving with offline utterances of
their own; prints the stream's chunk intervals (median, 95th percentile, maximum) and its wall time.  Arrivals make the table
run prefill passes: in line they sit in front of every running slot's next frame, beside the frame loop they do not
(POCKET_TTS_SYNC_PREFILL=1 selects the first)."""
import os, sys, time, threading, tempfile, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyhost"))
import numpy as np
import ptts_pack as pp, ptts_lib


def main():
    arrivals = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    spec = pp.Spec.full()
    path = os.path.join(tempfile.gettempdir(), "probe_full.ptts")
    pp.make_pack(path, spec, seed=1234, eos_bias=-20.0)
    eng = ptts_lib.Engine(path, precision="bf16", temperature=0.7, max_frames=125, seed=5)
    rng = np.random.default_rng(0)
    mk = lambda: (rng.integers(1, spec.vocab, size=int(rng.integers(8, 41))).astype(np.int64),
                  (rng.standard_normal((int(rng.integers(20, 63)), spec.d_model)) * 0.1).astype(np.float32))
    others = [mk() for _ in range(arrivals)]
    sid, sv = mk()
    eng.generate_ids([o[0] for o in others[:4]], [o[1] for o in others[:4]], queued=True, want_latents=False, want_audio=False)   # warm-up
    stamps = []

    def _cb(ptr, n, fin, _u):
        stamps.append(time.perf_counter())
    cb = ptts_lib.CHUNK_CB(_cb)

    def feeder():
        for ids, v in others:
            time.sleep(0.003)
            threading.Thread(target=lambda a=ids, b=v: eng.generate_ids([a], [b], want_latents=False, want_audio=False), daemon=False).start()
    ids = np.ascontiguousarray(sid, np.int64)
    voice = np.ascontiguousarray(sv, np.float32)
    f = threading.Thread(target=feeder)
    t0 = time.perf_counter()
    f.start()
    eng.L.pocket_tts_streaming_ids(eng.h, ids, len(ids), voice, voice.shape[0], None, 5, cb, None)
    wall = time.perf_counter() - t0
    f.join()
    while threading.active_count() > 1:
        time.sleep(0.01)
    gaps = np.diff(np.array(stamps)) * 1e3
    print("arrivals", arrivals, "chunks", len(stamps), "interval ms: median %.2f p95 %.2f max %.2f" % (np.median(gaps), np.percentile(gaps, 95), gaps.max()),
          "stream wall ms %.1f" % (wall * 1e3))
    eng.close()


main()
