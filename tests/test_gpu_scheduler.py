"""Concurrent callers of the single-utterance entry points share lock-step batches (host scheduler), and the
drop-in C API end to end: pocket_tts_create -> encode_voice_from_samples -> generate / generate_streaming with
a SentencePiece model and a weight pack in the reference's directory layout."""
import ctypes as C
import os
import shutil
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_concurrent_single_calls_are_coalesced(packs):
    import ptts_lib
    path, spec, _ = packs["tiny"]
    F = 10
    eng = ptts_lib.Engine(path, precision="fp32", temperature=0.0, lsd_steps=4, max_frames=F)
    rng = np.random.default_rng(3)
    n = 12
    ids = [rng.integers(0, spec.vocab, int(rng.integers(3, 15))).astype(np.int64) for _ in range(n)]
    voices = [(rng.standard_normal((int(rng.integers(1, 9)), spec.d_model)) * 0.5).astype(np.float32) for _ in range(n)]
    solo = [eng.generate_ids([ids[i]], [voices[i]])[0] for i in range(n)]
    before = eng.scheduler_stats()
    assert before["batches"] == n and before["largest"] == 1
    out = [None] * n
    errs = []
    gate = threading.Barrier(n)

    def work(i):
        try:
            gate.wait()
            out[i] = eng.generate_ids([ids[i]], [voices[i]])[0]
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    after = eng.scheduler_stats()
    assert after["utterances"] - before["utterances"] == n
    assert after["batches"] - before["batches"] < n, "no two concurrent calls shared a batch"
    assert after["largest"] > 1
    for i in range(n):
        assert out[i]["frames"] == solo[i]["frames"] == F
        assert out[i]["audio"].shape == solo[i]["audio"].shape
        # same utterance alone or inside a batch: early frames agree to rounding (later ones drift, the loop is chaotic)
        assert np.abs(out[i]["latents"][:3] - solo[i]["latents"][:3]).max() < 1e-3
    # an invalid request fails alone, before it can join a batch
    with pytest.raises(ptts_lib.PttsError, match="Text cannot be empty"):
        eng.generate_ids([np.zeros(0, np.int64)], [voices[0]])
    eng.close()


def test_drop_in_c_api_end_to_end(packs, tmp_path):
    import ptts_lib
    L = ptts_lib.load()
    path, spec, _ = packs["small"]
    models = tmp_path / "models" / "onnx"
    models.mkdir(parents=True)
    shutil.copy(path, models / "pocket_tts.ptts")
    tok = os.path.join(ROOT, "tests", "golden", "sp_unigram.model")
    cfg = ptts_lib.PocketTTSConfig(models_dir=str(models).encode(), tokenizer_path=tok.encode(), precision=b"fp32",
                                   temperature=0.5, lsd_steps=4, max_frames=7)
    h = L.pocket_tts_create(C.byref(cfg))
    assert h, L.pocket_tts_get_last_error()
    clip = (np.sin(np.arange(16000) / 9.0) * 0.4).astype(np.float32)
    v = L.pocket_tts_encode_voice_from_samples(h, clip, len(clip), 16000)     # resampled to 24 kHz inside
    assert v, L.pocket_tts_get_last_error()
    assert L.pocket_tts_voice_frames(v) == int(np.ceil(24000 / spec.frame_size))
    res = ptts_lib.AudioResult()
    assert L.pocket_tts_generate(h, b"hello from the test", v, C.byref(res)) == 0, L.pocket_tts_get_last_error()
    assert res.sample_rate == 24000 and res.sample_count == 7 * spec.frame_size     # EOS never fires with this pack
    audio = np.ctypeslib.as_array(res.data, shape=(res.sample_count,)).copy()
    assert np.isfinite(audio).all() and np.abs(audio).max() > 0
    L.pocket_tts_free_audio(C.byref(res))
    assert not res.data and res.sample_count == 0
    # streaming: chunk 3 over 7 frames -> 3, 3 and a final flush of 1
    got = []
    cb = ptts_lib.CHUNK_CB(lambda p, n, fin, u: got.append((n, fin)))
    sc = ptts_lib.StreamingConfig(chunk_size_frames=3, user_data=None)
    total = L.pocket_tts_generate_streaming(h, b"hello from the test", v, cb, C.byref(sc))
    assert total == 7 * spec.frame_size
    assert got == [(3 * spec.frame_size, 0), (3 * spec.frame_size, 0), (spec.frame_size, 1)]
    # cancellation from inside the callback (the reference polls the flag before every frame, src/pocket_tts.cpp:745-750,
    # and skips the tail flush, :897): the chunk in hand is the last one, nothing is marked final
    got2 = []

    def on_chunk(p, n, fin, u):
        got2.append((n, fin))
        L.pocket_tts_cancel_streaming(h)

    cb2 = ptts_lib.CHUNK_CB(on_chunk)
    total2 = L.pocket_tts_generate_streaming(h, b"hello from the test", v, cb2, C.byref(sc))
    assert got2 == [(3 * spec.frame_size, 0)] and total2 == 3 * spec.frame_size
    # the flag is cleared at the start of the next call (:698)
    got.clear()
    assert L.pocket_tts_generate_streaming(h, b"hello from the test", v, cb, C.byref(sc)) == 7 * spec.frame_size
    assert [f for _, f in got] == [0, 0, 1]
    # a chunk size beyond the utterance: one final chunk
    got.clear()
    sc_big = ptts_lib.StreamingConfig(chunk_size_frames=50, user_data=None)
    assert L.pocket_tts_generate_streaming(h, b"hello from the test", v, cb, C.byref(sc_big)) == 7 * spec.frame_size
    assert got == [(7 * spec.frame_size, 1)]
    # empty text is an error reported through get_last_error, not a crash
    assert L.pocket_tts_generate(h, b"   ", v, C.byref(res)) == -1
    assert b"Text cannot be empty" in L.pocket_tts_get_last_error()
    L.pocket_tts_free_voice(v)
    L.pocket_tts_destroy(h)


def test_size_limits_and_recovery(packs, oracle_mod):
    """Inputs at and beyond the sizes an engine was created for: too many tokens and a context longer than the KV cache are
    errors that leave the engine usable; no voice rows and a one-token text are valid inputs and match the oracle."""
    import ptts_lib
    from conftest import oracle_replay
    path, spec, _ = packs["small"]
    eng = ptts_lib.Engine(path, precision="fp32", temperature=0.0, lsd_steps=4, max_frames=4, max_text_tokens=8)
    rng = np.random.default_rng(3)
    voice = (rng.standard_normal((6, spec.d_model)) * 0.5).astype(np.float32)
    with pytest.raises(ptts_lib.PttsError, match="more tokens"):
        eng.generate_ids([rng.integers(0, spec.vocab, 9).astype(np.int64)], [voice])
    long_voice = (rng.standard_normal((400, spec.d_model)) * 0.5).astype(np.float32)     # cache holds 64 + 8 + 4 + 8 -> 128 rows
    with pytest.raises(ptts_lib.PttsError, match="KV-cache capacity"):
        eng.generate_ids([rng.integers(0, spec.vocab, 5).astype(np.int64)], [long_voice])
    with pytest.raises(ptts_lib.PttsError, match="KV-cache capacity"):
        eng.streaming_ids(rng.integers(0, spec.vocab, 5).astype(np.int64), long_voice, chunk_frames=2)
    o = oracle_mod.Oracle(path)
    cases = [(rng.integers(0, spec.vocab, 8).astype(np.int64), voice),                                   # the token limit itself
             (rng.integers(0, spec.vocab, 1).astype(np.int64), voice),                                   # one token
             (rng.integers(0, spec.vocab, 4).astype(np.int64), np.zeros((0, spec.d_model), np.float32))]  # no voice rows
    res = eng.generate_ids([c[0] for c in cases], [c[1] for c in cases])
    for (ids, vo), r in zip(cases, res):
        lat, eos, frames = oracle_replay(o, ids, vo, r["latents"], 4, 4)
        assert r["frames"] == frames
        assert np.abs(r["latents"] - lat).max() < 1e-3 and np.abs(r["eos"] - eos).max() < 1e-3
    eng.close()
