#!/usr/bin/env python3
"""Compares two dumps written by tools/ref_dump (one per library build): PCM SNR and max-abs, frame counts,
voice-embedding max-abs, and the streaming chunk layout.

    parity_report.py REF_DIR TEST_DIR [--min-snr 40] [--max-voice 1e-3] [--max-latent 1e-3]

SNR = 10 log10(sum ref^2 / sum (test - ref)^2) over the common length; exit status 1 when a bound fails or the
lengths / chunk layouts differ.  The bounds are the ones the engine is held to against the reference: PCM at
least 40 dB, voice embeddings within 1e-3 (fp32).  When both dumps hold latents.f32 / eos.f32 (ref_dump --latents,
a library that exposes them), the per-frame latent and EOS-logit max-abs are reported frame by frame against
--max-latent (1e-3 for fp32, 2e-2 for bf16 / int8); a free-running random-weight model drifts from frame to frame, so
the table shows where the two runs part.  Per-frame PCM SNR (1920 samples per frame) is reported in any case.
"""
import argparse
import json
import os
import sys

import numpy as np


def load(d):
    meta = json.load(open(os.path.join(d, "meta.json")))
    f = lambda n: np.fromfile(os.path.join(d, n), dtype="<f4")
    return meta, f("pcm.f32"), f("voice.f32"), f("stream_pcm.f32")


def snr_db(ref, test):
    n = min(len(ref), len(test))
    if n == 0:
        return float("nan")
    r, t = ref[:n].astype(np.float64), test[:n].astype(np.float64)
    noise = ((t - r) ** 2).sum()
    return float("inf") if noise == 0 else 10.0 * np.log10(max((r ** 2).sum(), 1e-300) / noise)


def load_latents(d, meta):
    lp, ep = os.path.join(d, "latents.f32"), os.path.join(d, "eos.f32")
    L = int(meta.get("latent_dim", 0) or 0)
    if not (os.path.exists(lp) and os.path.exists(ep) and L > 0):
        return None, None
    return np.fromfile(lp, dtype="<f4").reshape(-1, L), np.fromfile(ep, dtype="<f4")


def per_frame(ref, test, frame):
    """SNR of every whole frame of `frame` samples common to both."""
    n = min(len(ref), len(test)) // frame
    return [snr_db(ref[i * frame:(i + 1) * frame], test[i * frame:(i + 1) * frame]) for i in range(n)]


def report(ref_dir, test_dir, min_snr=40.0, max_voice=1e-3, max_latent=None, frame=1920):
    mr, pr, vr, sr = load(ref_dir)
    mt, pt, vt, st = load(test_dir)
    ok = True
    lines = []
    for k in ("text", "temperature", "lsd_steps", "max_frames", "chunk_frames"):
        if mr.get(k) != mt.get(k):
            lines.append(f"settings differ: {k} = {mr.get(k)!r} vs {mt.get(k)!r}")
            ok = False
    lines.append(f"frames: {mr['frames']} vs {mt['frames']}")
    ok &= mr["frames"] == mt["frames"]
    s = snr_db(pr, pt)
    n = min(len(pr), len(pt))
    lines.append(f"offline PCM: SNR {s:.1f} dB (bound {min_snr}), max-abs diff {np.abs(pr[:n] - pt[:n]).max() if n else 0:.3e}, peak {np.abs(pr).max() if len(pr) else 0:.3f}")
    ok &= bool(s >= min_snr)
    pf = per_frame(pr, pt, frame)
    if pf:
        worst = int(np.argmin(pf))
        lines.append(f"offline PCM per frame: worst SNR {pf[worst]:.1f} dB at frame {worst}; first frame below {min_snr} dB: "
                     f"{next((i for i, v in enumerate(pf) if v < min_snr), None)}")
    lr, er = load_latents(ref_dir, mr)
    lt, et = load_latents(test_dir, mt)
    if lr is not None and lt is not None:
        nf = min(len(lr), len(lt))
        d = np.abs(lr[:nf] - lt[:nf]).max(axis=1) if nf else np.zeros(0)
        ne = min(len(er), len(et))
        de = np.abs(er[:ne] - et[:ne]) if ne else np.zeros(0)
        lines.append(f"latents: {len(lr)} vs {len(lt)} frames, EOS logits: {len(er)} vs {len(et)}")
        ok &= len(lr) == len(lt) and len(er) == len(et)
        for i in range(nf):
            lines.append(f"  frame {i:4d}: latent max-abs {d[i]:.3e}  eos diff {de[i] if i < ne else float('nan'):.3e}")
        if nf:
            lines.append(f"latents: max-abs over frames {d.max():.3e} (frame {int(d.argmax())}), eos {de.max() if ne else 0:.3e}"
                         + (f" (bound {max_latent})" if max_latent is not None else ""))
            if max_latent is not None:
                ok &= bool(d.max() <= max_latent)
    else:
        lines.append("latents: not in both dumps (PCM-only comparison)")
    if len(vr) == len(vt) and len(vr):
        dv = float(np.abs(vr - vt).max())
        lines.append(f"voice embeddings: max-abs diff {dv:.3e} (bound {max_voice})")
        ok &= dv <= max_voice
    else:
        lines.append(f"voice embeddings: {len(vr)} vs {len(vt)} values")
        ok &= len(vr) == len(vt)
    lines.append(f"stream chunks: {mr['chunks']} vs {mt['chunks']}")
    ok &= mr["chunks"] == mt["chunks"]
    ss = snr_db(sr, st)
    lines.append(f"streamed PCM: SNR {ss:.1f} dB")
    ok &= bool(ss >= min_snr) or (len(sr) == 0 and len(st) == 0)
    lines.append("PARITY OK" if ok else "PARITY FAILED")
    return ok, lines


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("ref_dir")
    ap.add_argument("test_dir")
    ap.add_argument("--min-snr", type=float, default=40.0)
    ap.add_argument("--max-voice", type=float, default=1e-3)
    ap.add_argument("--max-latent", type=float, default=None)
    a = ap.parse_args()
    ok, lines = report(a.ref_dir, a.test_dir, a.min_snr, a.max_voice, a.max_latent)
    print("\n".join(lines))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
