#!/usr/bin/env python3
"""Regenerates the committed golden vectors.  Run from the repo root in the build container:

    python tests/golden/make_golden.py

Sources of truth:
  * tokenizer_ids.json  - ids from the `sentencepiece` Python package (the library the reference links,
    /root/reference/src/tokenizer.cpp:50-55) on the two committed models, which were trained here with
    SentencePieceTrainer on local text (unigram / BPE with byte fallback, vocab 800 / 900).
  * audio_ref.npz       - outputs of the reference's own src/audio_utils.cpp compiled as is
    (oracle/_ref/libaudio_ref.so, built by oracle/Makefile): resample, normalize, WAV decode.
  * tiny_network.npz    - outputs of oracle/ptts_oracle.c on the seeded tiny weight pack.  The network
    arithmetic has no reference output to pin against (parity unpinned); these vectors pin the checker
    against its own history and give the GPU tests a fixture that does not need the checker at run time.
"""
import json
import os
import struct
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pocket-tts.cpp_amd", "pyhost"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

TEXTS = [
    "Hello, this is a test.", "Hello from C! This is a test.", "Hello from Python! This is a test of the Pocket TTS library.",
    "Hello! This is a test of the streaming audio generation feature.", "  multiple   spaces  here ", "a", "A.",
    "The quick brown fox jumps over the lazy dog.", "naïve café résumé — “quotes” …", "日本語のテキスト", "tab\tseparated\nnewline",
    "numbers 12345 and symbols #$%^&*()", "ﬁne ligature ＡＢＣ fullwidth ½", "emoji 😀 test", "x" * 50,
    "Licensed under the Apache License, Version 2.0 (the \"License\");", "hello world", "what's up?", "trailing space   ",
]


def wav_bytes(fmt, channels, rate, bits, payload, extra_chunk=True):
    fmt_chunk = struct.pack("<HHIIHH", fmt, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits)
    chunks = b"fmt " + struct.pack("<I", 16) + fmt_chunk
    if extra_chunk:
        chunks += b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\0"   # odd-sized chunk + pad byte
    chunks += b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks


def main():
    import sentencepiece as spm
    import oracle_py
    import ptts_pack as pp

    # ---- tokenizer
    tok = {}
    for name in ("sp_unigram", "sp_bpe_bytefallback"):
        sp = spm.SentencePieceProcessor(model_file=os.path.join(HERE, name + ".model"))
        tok[name] = {"vocab": sp.get_piece_size(), "ids": [sp.encode(t) for t in TEXTS]}
    with open(os.path.join(HERE, "tokenizer_ids.json"), "w") as f:
        json.dump({"texts": TEXTS, "models": tok}, f, ensure_ascii=False, indent=0)

    # ---- reference audio utils
    oracle_py.build()
    R = oracle_py.ref_audio()
    if R is None:
        raise SystemExit("oracle/_ref/libaudio_ref.so missing: /root/reference is needed to regenerate audio goldens")
    rng = np.random.default_rng(2024)
    t = np.arange(3000) / 44100.0
    sig = (0.5 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3100 * t) + 0.05 * rng.standard_normal(len(t))).astype(np.float32)
    out = {"sig": sig}
    for a, b in [(44100, 24000), (16000, 24000), (48000, 24000), (22050, 24000), (8000, 24000), (24000, 16000)]:
        cap = int(np.ceil(len(sig) * b / a)) + 8
        buf = np.empty(cap, np.float32)
        n = R.audio_ref_resample(sig, len(sig), a, b, buf, cap)
        out[f"resample_{a}_{b}"] = buf[:n].copy()
    loud = (sig * 3.0).astype(np.float32)
    buf = np.empty(len(loud), np.float32)
    R.audio_ref_normalize(loud, len(loud), buf)
    out["normalize_loud"] = buf.copy()
    R.audio_ref_normalize(sig, len(sig), buf)
    out["normalize_quiet"] = buf.copy()
    # WAV decode cases (bytes stored so the test can feed the same file to the product)
    st = np.stack([sig[:1200], sig[300:1500]], axis=1)
    cases = {
        "pcm16_mono_44k": wav_bytes(1, 1, 44100, 16, (np.clip(sig[:1500], -1, 1) * 32767).astype("<i2").tobytes()),
        "pcm16_stereo_24k": wav_bytes(1, 2, 24000, 16, (np.clip(st, -1, 1) * 32767).astype("<i2").tobytes()),
        "f32_stereo_48k": wav_bytes(3, 2, 48000, 32, st.astype("<f4").tobytes(), extra_chunk=False),
        "f32_mono_loud_24k": wav_bytes(3, 1, 24000, 32, loud[:1000].astype("<f4").tobytes()),
    }
    i24 = (np.clip(sig[:900], -1, 1) * 8388607).astype(np.int32)
    b24 = b"".join(struct.pack("<i", int(v))[:3] for v in i24)
    cases["pcm24_mono_22k"] = wav_bytes(1, 1, 22050, 24, b24)
    with tempfile.TemporaryDirectory() as d:
        for name, data in cases.items():
            path = os.path.join(d, name + ".wav")
            with open(path, "wb") as f:
                f.write(data)
            buf = np.empty(1 << 16, np.float32)
            n = R.audio_ref_load_wav(path.encode(), 24000, buf, len(buf))
            assert n > 0, name
            out["wav_" + name] = np.frombuffer(data, np.uint8).copy()
            out["dec_" + name] = buf[:n].copy()
    np.savez_compressed(os.path.join(HERE, "audio_ref.npz"), **out)

    # ---- tiny network vectors from the C oracle
    spec = pp.Spec.tiny()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "tiny.ptts")
        w = pp.make_pack(path, spec, seed=7, eos_bias=-20.0, layer_scale=0.2)
        crc = zlib.crc32(b"".join(w[k].tobytes() for k in sorted(w)))
        o = oracle_py.Oracle(path)
        r = np.random.default_rng(99)
        ids = r.integers(0, spec.vocab, 9).astype(np.int64)
        voice = (r.standard_normal((5, spec.d_model)) * 0.5).astype(np.float32)
        g = o.generate(ids, voice, lsd_steps=10, max_frames=8)
        clip = (r.standard_normal(700) * 0.3).astype(np.float32)
        enc = o.mimi_encode(clip)
        cond = r.standard_normal(spec.d_model).astype(np.float32)
        x0 = (r.standard_normal(spec.latent_dim) * 0.8).astype(np.float32)
        np.savez_compressed(os.path.join(HERE, "tiny_network.npz"), pack_crc32=np.array([crc], np.uint64), ids=ids, voice=voice,
                            latents=g["latents"], eos=g["eos"], audio=g["audio"][: 2 * spec.frame_size], clip=clip, enc=enc,
                            cond=cond, x0=x0, flow10=o.flow_euler(cond, x0, 10), text_emb=o.text_conditioner(ids))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
