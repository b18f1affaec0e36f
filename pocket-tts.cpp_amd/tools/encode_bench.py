#!/usr/bin/env python3
"""Voice-clone encode throughput: the Mimi encoder over N seeded clips of 5 s (the reference cuts every voice clip to
120000 samples before encoding, src/pocket_tts.cpp:256-259), --batch clips per call (1 = the drop-in call's form).
Host samples in, host embedding rows out.  --wavs N: N ten-second WAV files on disk instead (float32 24 kHz mono, or with
--wav-rate 48000 PCM16 stereo that has to be resampled), through pocket_tts_encode_voices: decode, resample, attenuate and
cut on host threads while the GPU encodes.  Prints one JSON line."""
import argparse, json, os, sys, tempfile, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pocket-tts.cpp_amd", "pyhost"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=64)
    ap.add_argument("--samples", type=int, default=120000)
    ap.add_argument("--batch", type=int, default=32, help="clips per call (the library stacks them 8 at a time)")
    ap.add_argument("--wavs", type=int, default=0, help="encode this many 10 s WAV files through the many-files call")
    ap.add_argument("--wav-rate", type=int, default=24000)
    args = ap.parse_args()
    import ptts_pack as pp
    import ptts_lib
    spec = pp.Spec.full()
    path = os.path.join(tempfile.gettempdir(), f"ptts_bench_full_{os.getuid()}_0.ptts")
    pp.make_pack(path, spec, seed=1234, eos_bias=-20.0)
    eng = ptts_lib.Engine(path, precision="bf16", max_frames=10)
    rng = np.random.default_rng(5)
    if args.wavs > 0:
        import wave
        d = tempfile.mkdtemp(prefix="ptts_wavs_")
        n = 10 * args.wav_rate
        tt = np.arange(n) / float(args.wav_rate)
        base = [(0.2 * np.sin(2 * np.pi * (110 + 7 * i) * tt) + 0.05 * rng.standard_normal(n)).astype(np.float32) for i in range(4)]
        paths = []
        for i in range(args.wavs + 8):
            p = os.path.join(d, f"v{i}.wav")
            x = base[i % 4] * np.float32(0.5 + 0.001 * i)
            if args.wav_rate == 24000:
                ptts_lib.audio_save_wav(p, x, 24000)
            else:
                q = np.clip(np.round(x * 32767.0), -32768, 32767).astype("<i2")
                with wave.open(p, "wb") as w:
                    w.setnchannels(2); w.setsampwidth(2); w.setframerate(args.wav_rate)
                    w.writeframes(np.stack([q, q], axis=1).reshape(-1).tobytes())
            paths.append(p)
        eng.encode_voice_paths(paths[args.wavs:])                 # warm-up on files of their own (results are cached by path)
        t0 = time.perf_counter()
        out = eng.encode_voice_paths(paths[:args.wavs])
        dt = time.perf_counter() - t0
        th = time.perf_counter()
        for p in paths[:8]:
            ptts_lib.audio_load_wav(p)
        host_ms = (time.perf_counter() - th) / 8 * 1e3
        eng.close()
        for p in paths:
            os.remove(p)
        os.rmdir(d)
        print(json.dumps(dict(metric="voice encode throughput from WAV files", files=args.wavs, seconds_per_file=10, file_rate=args.wav_rate,
                              file_format="float32 mono" if args.wav_rate == 24000 else "PCM16 stereo", rows_out=int(out[-1].shape[0]),
                              ms_per_file=round(dt / args.wavs * 1e3, 3), files_per_s=round(args.wavs / dt, 1),
                              host_prepare_ms_per_file_one_thread=round(host_ms, 2))))
        return
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
