"""GPU parity of the whole hot path (prefill -> frame loop -> flow head -> decoder) against the oracle's
restatement of the reference orchestration, through the C ABI.  Also the invariants the reference's
own examples imply: sample-count law, chunking invariance, temperature-0 determinism."""
import os

import numpy as np
import pytest

from conftest import oracle_decode, oracle_replay, snr_db

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ptts():
    import ptts_lib
    ptts_lib.load()
    return ptts_lib


def _inputs(spec, rng, n, t_lens, v_lens):
    ids = [rng.integers(0, spec.vocab, t).astype(np.int64) for t in t_lens[:n]]
    voices = [(rng.standard_normal((v, spec.d_model)) * 0.5).astype(np.float32) for v in v_lens[:n]]
    return ids, voices


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_offline_batch_matches_oracle(ptts, packs, oracle_mod, name):
    path, spec, _ = packs[name]
    F = 12 if name == "tiny" else 6
    eng = ptts.Engine(path, precision="fp32", temperature=0.0, lsd_steps=10, max_frames=F)
    o = oracle_mod.Oracle(path)
    rng = np.random.default_rng(21)
    ids, voices = _inputs(spec, rng, 3, [9, 4, 17], [5, 11, 2])
    res = eng.generate_ids(ids, voices)
    for i in range(3):
        lat, eos, frames = oracle_replay(o, ids[i], voices[i], res[i]["latents"], 10, F)
        assert res[i]["frames"] == frames == F
        assert res[i]["audio"].shape[0] == F * spec.frame_size          # sample-count law
        assert res[i]["sample_rate"] == 24000
        assert np.abs(res[i]["latents"] - lat).max() < 1e-3             # per-frame, same history on both sides
        assert np.abs(res[i]["eos"] - eos).max() < 1e-3
        assert snr_db(oracle_decode(o, res[i]["latents"]), res[i]["audio"]) >= 40.0
        free = o.generate(ids[i], voices[i], lsd_steps=10, max_frames=4, want_audio=False)
        assert np.abs(res[i]["latents"][:4] - free["latents"]).max() < 1e-3   # free-running, before drift matters
    # temperature 0 is deterministic: a second run is bit-identical
    res2 = eng.generate_ids(ids, voices)
    for i in range(3):
        np.testing.assert_array_equal(res[i]["latents"], res2[i]["latents"])
        np.testing.assert_array_equal(res[i]["audio"], res2[i]["audio"])
    eng.close()


def test_eos_stop_rule_matches_oracle(ptts, packs, oracle_mod):
    """EOS bias near the -4.0 threshold: utterances stop at different steps; frame counts must match."""
    path, spec, _ = packs["tiny_eos"]
    F = 40
    eng = ptts.Engine(path, precision="fp32", temperature=0.0, lsd_steps=4, max_frames=F)
    o = oracle_mod.Oracle(path)
    rng = np.random.default_rng(33)
    ids, voices = _inputs(spec, rng, 6, [9, 4, 17, 3, 8, 12], [5, 11, 2, 7, 3, 9])
    res = eng.generate_ids(ids, voices)
    counts = []
    for i in range(6):
        lat, eos, frames = oracle_replay(o, ids[i], voices[i], res[i]["latents"], 4, F)
        counts.append(frames)
        assert np.abs(eos + 4.0).min() > 1e-3, "test input sits on the EOS threshold; pick another seed"
        assert res[i]["frames"] == frames, (i, res[i]["frames"], frames)
        assert len(res[i]["eos"]) == len(eos)
        assert np.abs(res[i]["eos"] - eos).max() < 1e-3
        assert res[i]["audio"].shape[0] == frames * spec.frame_size
        if frames:
            assert np.abs(res[i]["latents"] - lat).max() < 1e-3
            assert snr_db(oracle_decode(o, res[i]["latents"]), res[i]["audio"]) >= 40.0
    assert len(set(counts)) > 1 and min(counts) < F, counts   # the test really exercised early stops
    eng.close()


@pytest.mark.parametrize("chunk,F", [(5, 12), (5, 10), (4, 12), (20, 33)])
def test_streaming_schedule_and_chunk_invariance(ptts, packs, oracle_mod, chunk, F):
    path, spec, _ = packs["tiny"]
    eng = ptts.Engine(path, precision="fp32", temperature=0.0, lsd_steps=4, max_frames=F)
    o = oracle_mod.Oracle(path)
    rng = np.random.default_rng(41)
    ids, voices = _inputs(spec, rng, 1, [6], [4])
    total, chunks, _ = eng.streaming_ids(ids[0], voices[0], chunk_frames=chunk)
    want = o.generate(ids[0], voices[0], lsd_steps=4, max_frames=F, chunk_frames=chunk)
    assert [(len(c), f) for c, f in chunks] == [(len(c), f) for c, f in want["chunks"]]
    assert total == sum(len(c) for c, _ in chunks) == F * spec.frame_size
    streamed = np.concatenate([c for c, _ in chunks])
    off = eng.generate_ids(ids, voices)[0]
    # same stateful decoder, different chunking: only the summation order inside the attention tiles differs
    np.testing.assert_allclose(streamed, off["audio"], rtol=0, atol=2e-5)
    assert snr_db(oracle_decode(o, off["latents"]), streamed) >= 40.0
    eng.close()


def test_explicit_noise_start_points(ptts, packs, oracle_mod):
    path, spec, _ = packs["tiny"]
    F = 8
    eng = ptts.Engine(path, precision="fp32", temperature=0.7, lsd_steps=6, max_frames=F, seed=1)
    o = oracle_mod.Oracle(path)
    rng = np.random.default_rng(55)
    ids, voices = _inputs(spec, rng, 2, [6, 10], [4, 6])
    noise = [(rng.standard_normal((F, spec.latent_dim)) * np.sqrt(0.7)).astype(np.float32) for _ in range(2)]
    res = eng.generate_ids(ids, voices, noise=noise)
    for i in range(2):
        lat, _, frames = oracle_replay(o, ids[i], voices[i], res[i]["latents"], 6, F, noise=noise[i])
        assert frames == F and np.abs(res[i]["latents"] - lat).max() < 1e-3
    eng.close()
    # internal generator: equal seeds repeat, different seeds differ
    outs = []
    for seed in (1, 1, 2):
        e2 = ptts.Engine(path, precision="fp32", temperature=0.7, lsd_steps=6, max_frames=F, seed=seed)
        outs.append(e2.generate_ids(ids, voices, want_audio=False)[0]["latents"])
        e2.close()
    np.testing.assert_array_equal(outs[0], outs[1])
    assert np.abs(outs[0] - outs[2]).max() > 1e-3


def test_bf16_latents_within_tolerance(ptts, packs, oracle_mod):
    path, spec, _ = packs["small"]
    F = 6
    eng = ptts.Engine(path, precision="bf16", temperature=0.0, lsd_steps=10, max_frames=F)
    o = oracle_mod.Oracle(path)
    rng = np.random.default_rng(61)
    ids, voices = _inputs(spec, rng, 2, [9, 4], [5, 11])
    res = eng.generate_ids(ids, voices)
    worst = 0.0
    for i in range(2):
        lat, _, frames = oracle_replay(o, ids[i], voices[i], res[i]["latents"], 10, F)
        assert frames == F
        worst = max(worst, np.abs(res[i]["latents"] - lat).max())
        assert snr_db(oracle_decode(o, res[i]["latents"]), res[i]["audio"]) >= 20.0
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/bf16_latent_err.txt", "w") as f:
            f.write(f"bf16 per-frame (teacher-forced) max-abs latent error over {F} frames: {worst}\n")
    # The 2e-2 bound is stated for the full-size model (tests/test_gpu_fullsize.py); relative rounding error
    # shrinks with layer width, so this 128/256-wide model gets a proportionally looser bound.
    assert worst < 1e-1, worst
    eng.close()


def test_frame_graph_survives_scratch_growth(ptts, packs, oracle_mod):
    """A short call captures the frame graph of a 2-utterance batch; the next call on the same engine has a prefill
    of 2 x 330 rows, which outgrows the transformer scratch the graph's decode step runs in.  The graph must be
    captured again on the new buffers: the second call is checked against the oracle frame by frame, and a third,
    short call against the first."""
    path, spec, _ = packs["tiny"]
    F = 5
    eng = ptts.Engine(path, precision="fp32", temperature=0.0, lsd_steps=4, max_frames=F)
    o = oracle_mod.Oracle(path)
    rng = np.random.default_rng(71)
    ids_s, voices_s = _inputs(spec, rng, 2, [6, 9], [4, 7])
    ids_l, voices_l = _inputs(spec, rng, 2, [300, 250], [30, 12])
    first = eng.generate_ids(ids_s, voices_s)
    big = eng.generate_ids(ids_l, voices_l)
    for i in range(2):
        lat, eos, frames = oracle_replay(o, ids_l[i], voices_l[i], big[i]["latents"], 4, F)
        assert big[i]["frames"] == frames == F
        assert np.abs(big[i]["latents"] - lat).max() < 1e-3 and np.abs(big[i]["eos"] - eos).max() < 1e-3
    again = eng.generate_ids(ids_s, voices_s)
    for i in range(2):
        np.testing.assert_array_equal(first[i]["latents"], again[i]["latents"])
        np.testing.assert_array_equal(first[i]["audio"], again[i]["audio"])
    eng.close()


def test_error_conventions(ptts, packs):
    L = ptts.load()
    assert L.pocket_tts_version() == b"1.0.0"
    cfg = ptts.PocketTTSConfig(models_dir=b"/nonexistent/dir", tokenizer_path=None, precision=None, temperature=0, lsd_steps=0, max_frames=0)
    import ctypes as C
    h = L.pocket_tts_create(C.byref(cfg))
    assert not h
    assert b"Failed to create PocketTTS" in L.pocket_tts_get_last_error()
    res = ptts.AudioResult()
    assert L.pocket_tts_generate(None, b"x", None, C.byref(res)) == -1
    assert L.pocket_tts_get_last_error() == b"Invalid parameters"
    L.pocket_tts_free_audio(C.byref(res))      # data == NULL: no-op
    L.pocket_tts_destroy(None)
    L.pocket_tts_free_voice(None)
    L.pocket_tts_cancel_streaming(None)


def test_temperature_override_from_the_environment(ptts, packs):
    """POCKET_TTS_TEMPERATURE replaces the configured temperature when the handle is created: the reference's C API cannot
    pass 0 (fields are applied only when > 0, src/pocket_tts_c.cpp:33-40), so callers of the unchanged bindings get
    deterministic runs this way.  With the override at 0 two handles with different seeds agree bit for bit; without it
    (temperature 0.7) they do not; a value that is not a number fails the creation with a message."""
    import os
    path, spec, _ = packs["tiny"]
    rng = np.random.default_rng(8)
    ids = [rng.integers(0, spec.vocab, 6).astype(np.int64)]
    voices = [(rng.standard_normal((4, spec.d_model)) * 0.5).astype(np.float32)]

    def run(seed):
        eng = ptts.Engine(path, precision="fp32", temperature=0.7, lsd_steps=4, max_frames=5, seed=seed)
        out = eng.generate_ids(ids, voices)[0]["latents"].copy()
        eng.close()
        return out

    a, b = run(1), run(2)
    assert np.abs(a - b).max() > 1e-3
    os.environ["POCKET_TTS_TEMPERATURE"] = "0"
    try:
        c, d = run(1), run(2)
    finally:
        os.environ.pop("POCKET_TTS_TEMPERATURE", None)
    np.testing.assert_array_equal(c, d)
    os.environ["POCKET_TTS_TEMPERATURE"] = "warm"
    try:
        with pytest.raises(ptts.PttsError, match="POCKET_TTS_TEMPERATURE"):
            run(1)
    finally:
        os.environ.pop("POCKET_TTS_TEMPERATURE", None)
