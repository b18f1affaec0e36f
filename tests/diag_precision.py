"""Per-graph error of the GPU path (fp32 and bf16) against the oracle on a full-size synthetic pack."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pocket-tts.cpp_amd", "pyhost")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ptts_pack as pp, ptts_lib, oracle_py

size = sys.argv[1] if len(sys.argv) > 1 else "full"
spec = {"full": pp.Spec.full, "small": pp.Spec.small, "tiny": pp.Spec.tiny}[size]()
path = f"/tmp/diag_{size}.ptts"
t0 = time.time(); pp.make_pack(path, spec, seed=1234); print("pack written", time.time() - t0, flush=True)
o = oracle_py.Oracle(path)
rng = np.random.default_rng(0)
B, S = 2, 3
voices = [(rng.standard_normal((nv, spec.d_model)) * 0.5).astype(np.float32) for nv in (20, 9)]
texts = [o.text_conditioner(rng.integers(0, spec.vocab, t)) for t in (12, 7)]
seq = rng.standard_normal((S, B, spec.latent_dim)).astype(np.float32); seq[0] = np.nan
ost = [o.lm_state(256) for _ in range(B)]
oc = np.zeros((S, B, spec.d_model), np.float32); oe = np.zeros((S, B), np.float32)
t0 = time.time()
for b in range(B):
    o.lm_main(ost[b], None, voices[b]); o.lm_main(ost[b], None, texts[b])
    for s in range(S):
        c, e = o.lm_main(ost[b], seq[s, b:b+1], None); oc[s, b] = c; oe[s, b] = e
print("oracle lm done", time.time() - t0, "cond rms", np.sqrt((oc**2).mean()), flush=True)
cond = oc[1]
x0 = np.zeros((B, spec.latent_dim), np.float32)
ox1 = np.stack([o.flow_euler(cond[b], x0[b], 1) for b in range(B)])
ox10 = np.stack([o.flow_euler(cond[b], x0[b], 10) for b in range(B)])
lat = rng.standard_normal((B, 4, spec.latent_dim)).astype(np.float32)
t0 = time.time()
oa = []
for b in range(B):
    ds = o.dec_state(); oa.append(o.mimi_decode(ds, lat[b]))
oa = np.stack(oa); print("oracle dec done", time.time() - t0, "latent rms", np.sqrt((ox10**2).mean()), "audio rms", np.sqrt((oa**2).mean()), flush=True)
for prec in ("fp32", "bf16"):
    for lsd in (1, 10):
        eng = ptts_lib.Engine(path, precision=prec, lsd_steps=lsd, max_frames=16)
        if lsd == 10:
            st = eng.state(B)
            for b in range(B):
                eng.lm_prefill(st, b, voices[b]); eng.lm_prefill(st, b, texts[b])
            ec = 0; ee = 0
            for s in range(S):
                c, e = eng.lm_step(st, seq[s]); ec = max(ec, np.abs(c - oc[s]).max()); ee = max(ee, np.abs(e - oe[s]).max())
            a = eng.mimi_decode(st, lat)
            snr = 10 * np.log10((oa.astype(np.float64)**2).sum() / ((a - oa).astype(np.float64)**2).sum())
            print(f"{prec}: lm cond max-abs err {ec:.3e} eos err {ee:.3e}; decoder SNR {snr:.1f} dB max-abs {np.abs(a-oa).max():.3e}", flush=True)
            st.close()
        x = eng.flow_euler(cond, x0)
        ref = ox1 if lsd == 1 else ox10
        print(f"{prec}: flow euler steps={lsd} max-abs err {np.abs(x - ref).max():.3e} (|x| max {np.abs(ref).max():.2f})", flush=True)
        eng.close()
