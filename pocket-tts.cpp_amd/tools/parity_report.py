#!/usr/bin/env python3
"""Compares two dumps written by tools/ref_dump (one per library build): PCM SNR and max-abs, frame counts,
voice-embedding max-abs, and the streaming chunk layout.

    parity_report.py REF_DIR TEST_DIR [--min-snr 40] [--max-voice 1e-3]

SNR = 10 log10(sum ref^2 / sum (test - ref)^2) over the common length; exit status 1 when a bound fails or the
lengths / chunk layouts differ.  The bounds are the ones the engine is held to against the reference: PCM at
least 40 dB, voice embeddings within 1e-3 (fp32).
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


def report(ref_dir, test_dir, min_snr=40.0, max_voice=1e-3):
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
    a = ap.parse_args()
    ok, lines = report(a.ref_dir, a.test_dir, a.min_snr, a.max_voice)
    print("\n".join(lines))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
