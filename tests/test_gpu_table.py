"""The slot table (continuous batching) behind the single-utterance entry points, through the C ABI:

  * several hundred ragged utterances queued on one handle: every utterance's frame count, latents and EOS logits
    against the oracle's teacher-forced replay, and the table's clock against the ideal sum(frames) / 64;
  * eight threads streaming at once: first-chunk latency against one caller alone, and every stream's audio against
    the same utterance generated alone;
  * the same utterance served in a crowded table and alone gives the same frames (slots are independent).

The reference serves one utterance per call (src/pocket_tts.cpp:548-657, :686-979); the table's per-utterance results
are checked against exactly that behaviour, restated by the oracle."""
import os
import threading

import numpy as np
import pytest

from conftest import oracle_decode, oracle_replay, snr_db

pytestmark = pytest.mark.gpu

# two of the tests below state what ONE table does (its largest occupancy; first-chunk latency with the persistent flow head,
# which only the first engine on a GPU holds): they are skipped when the environment asks for several engines per GPU
one_engine = pytest.mark.skipif(os.environ.get("POCKET_TTS_ENGINES_PER_GPU", "1") not in ("", "1"),
                                reason="states the behaviour of a single table; POCKET_TTS_ENGINES_PER_GPU asks for several")


def _note(line):
    import os
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/table_stats.txt", "a") as f:
            f.write(line + "\n")


def _ragged(spec, rng, n, tmax=17, vmax=11):
    ids = [rng.integers(0, spec.vocab, int(rng.integers(1, tmax + 1))).astype(np.int64) for _ in range(n)]
    voices = [(rng.standard_normal((int(rng.integers(1, vmax + 1)), spec.d_model)) * 0.5).astype(np.float32) for _ in range(n)]
    return ids, voices


def test_queue_of_ragged_utterances_refills_slots(packs, oracle_mod):
    """1024 utterances (sixteen tables' worth) on a pack whose EOS logit crosses the threshold anywhere between the first
    steps and never (same weights as tiny_eos, lower bias), max_frames 150.
    Frame counts must equal the oracle's for every utterance; the number of frame steps the table ran must stay within
    1.15 x sum(frames) / 64, which lock-step batches of 64 (each as long as its longest member) cannot meet."""
    import ptts_lib as ptts
    path, spec, _ = packs["tiny_spread"]
    F = 150
    N = 1024
    eng = ptts.Engine(path, precision="fp32", temperature=0.0, lsd_steps=4, max_frames=F)
    o = oracle_mod.Oracle(path)
    rng = np.random.default_rng(2024)
    ids, voices = _ragged(spec, rng, N)
    res = eng.generate_ids(ids, voices, queued=True)
    st = eng.scheduler_stats()
    total = 0
    lockstep = 0
    worst = 0.0
    counts = []
    for i in range(N):
        lat, eos, frames = oracle_replay(o, ids[i], voices[i], res[i]["latents"], 4, F)
        assert res[i]["frames"] == frames, (i, res[i]["frames"], frames)
        assert res[i]["audio"].shape[0] == frames * spec.frame_size
        assert len(res[i]["eos"]) == len(eos)
        worst = max(worst, float(np.abs(res[i]["eos"] - eos).max()))
        if frames:
            worst = max(worst, float(np.abs(res[i]["latents"] - lat).max()))
        counts.append(frames)
        total += frames
    for i in range(0, N, 64):
        lockstep += max(counts[i:i + 64]) + 3   # a lock-step batch runs until its last member has stopped
    assert worst < 1e-3
    # audio of a sample of utterances against the oracle decoder (early, late, shortest, longest)
    for i in sorted({0, 1, 63, 64, 200, N - 1, int(np.argmin(counts)), int(np.argmax(counts))}):
        if counts[i]:
            assert snr_db(oracle_decode(o, res[i]["latents"]), res[i]["audio"]) >= 40.0, i
    ideal = total / 64.0
    _note(f"{N} ragged utterances, tiny_spread, max_frames 150: frames min/mean/max {min(counts)}/{total / N:.1f}/{max(counts)}; "
          f"table frame steps {st['frame_steps']}, ideal sum(frames)/64 = {ideal:.1f} (ratio {st['frame_steps'] / ideal:.3f}), lock-step batches of 64 "
          f"would run {lockstep} steps; admissions {st['admissions']}, decoder calls {st['decode_calls']}, activations {st['batches']}")
    assert len(set(counts)) > 8 and min(counts) < F
    assert st["utterances"] == N and st["frames"] == total
    assert st["frame_steps"] <= 1.15 * ideal, (st, ideal)
    eng.close()


def test_crowded_table_equals_utterance_alone(packs, oracle_mod):
    """One utterance generated alone and the same utterance as number 70 of 100 queued ones: same frame count, latents
    equal to rounding (the kernels treat rows independently; only the launch width differs)."""
    import ptts_lib as ptts
    path, spec, _ = packs["tiny_eos"]
    F = 60
    eng = ptts.Engine(path, precision="fp32", temperature=0.0, lsd_steps=4, max_frames=F)
    rng = np.random.default_rng(5)
    ids, voices = _ragged(spec, rng, 100)
    alone = eng.generate_ids([ids[70]], [voices[70]])[0]
    crowd = eng.generate_ids(ids, voices, queued=True)[70]
    assert alone["frames"] == crowd["frames"] > 0
    np.testing.assert_allclose(crowd["latents"], alone["latents"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(crowd["audio"], alone["audio"], rtol=0, atol=1e-4)
    eng.close()


@pytest.fixture(scope="module")
def full_pack(tmp_path_factory):
    import ptts_pack as pp
    spec = pp.Spec.full()
    path = str(tmp_path_factory.mktemp("table") / "full.ptts")
    pp.make_pack(path, spec, seed=1234, eos_bias=-20.0)
    return path, spec


@one_engine
def test_eight_concurrent_streams_first_chunk_latency(full_pack):
    """Full-size model, bf16, chunk of 5 frames.  One caller alone, then eight threads at once on the same handle: each
    stream's first chunk must arrive within twice the single caller's latency (the streams ride one table instead of
    queueing behind each other), with the reference's chunk schedule and its own utterance's audio."""
    import ptts_lib as ptts
    path, spec = full_pack
    F = 25
    eng = ptts.Engine(path, precision="bf16", temperature=0.0, lsd_steps=10, max_frames=F)
    rng = np.random.default_rng(17)
    ids = [rng.integers(0, spec.vocab, 24).astype(np.int64) for _ in range(8)]
    voices = [(rng.standard_normal((40, spec.d_model)) * 0.5).astype(np.float32) for _ in range(8)]
    # warm-up: builds the table, captures the frame graphs of the widths used below
    eng.streaming_ids(ids[0], voices[0], chunk_frames=5)
    eng.generate_ids(ids, voices, queued=True, want_latents=False)
    single = []
    alone = []
    for i in range(8):
        total, chunks, t = eng.streaming_ids(ids[i], voices[i], chunk_frames=5)
        single.append(t)
        alone.append(np.concatenate([c for c, _ in chunks]))
        assert [len(c) for c, _ in chunks] == [5 * spec.frame_size] * 5
    t_single = float(np.median(single))
    out = [None] * 8
    errs = []
    gate = threading.Barrier(8)

    def run(i):
        try:
            gate.wait()
            out[i] = eng.streaming_ids(ids[i], voices[i], chunk_frames=5)
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=run, args=(i,)) for i in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    firsts = [o[2] for o in out]
    _note(f"full size bf16 streaming, chunk 5: first chunk alone {t_single * 1e3:.2f} ms (median of 8); eight concurrent callers "
          f"{', '.join(f'{x * 1e3:.2f}' for x in sorted(firsts))} ms")
    # Routing: every caller got its own utterance.  A random-weight network amplifies the rounding differences between launch
    # widths from frame to frame (28 dB apart after 25 bf16 frames, measured), so sample parity is asserted where both sides see
    # the same history (the queue test above); here the first chunk must match the caller's own utterance and nobody else's.
    n0 = 5 * spec.frame_size
    for i in range(8):
        total, chunks, _ = out[i]
        assert [len(c) for c, _ in chunks] == [5 * spec.frame_size] * 5
        got = np.concatenate([c for c, _ in chunks])[:n0]
        corr = [float(np.corrcoef(got, alone[j][:n0])[0, 1]) for j in range(8)]
        assert corr[i] > 0.99 and max(c for j, c in enumerate(corr) if j != i) < 0.9, (i, corr)
    assert max(firsts) <= 2.0 * t_single, (firsts, t_single)
    eng.close()


def test_int8_512_ragged_utterances_through_the_table(full_pack, oracle_mod):
    """int8 engine, full-size model, 512 utterances with ragged prompts queued on one handle (the per-GPU share of the
    utterance-sharded configuration when one GPU serves it all): table bookkeeping, sizes, and the first and last
    frames of a sample of utterances (first wave, a refilled slot, the last wave) replayed by the int8 oracle and the
    float32 oracle on the same history.  The bounds are the regression guards of the int8 engine's own parity test
    (the 2e-2 specification against float32 is asserted, as an expected failure, in test_gpu_configs.py)."""
    import ptts_lib as ptts
    import ptts_pack as pp
    from test_gpu_configs import _replay_selected
    path, spec = full_pack
    F, N = 30, 512
    eng = ptts.Engine(path, precision="int8", temperature=0.0, lsd_steps=10, max_frames=F)
    rng = np.random.default_rng(3)
    ids = [rng.integers(0, spec.vocab, int(rng.integers(4, 41))).astype(np.int64) for _ in range(N)]
    voices = [(rng.standard_normal((int(rng.integers(8, 63)), spec.d_model)) * 0.5).astype(np.float32) for _ in range(N)]
    res = eng.generate_ids(ids, voices, queued=True)
    st = eng.scheduler_stats()
    for i in range(N):
        assert res[i]["frames"] == F and res[i]["audio"].shape[0] == F * spec.frame_size
        assert np.isfinite(res[i]["audio"]).all() and np.isfinite(res[i]["latents"]).all()
    assert st["utterances"] == N and st["largest"] == 64 and st["frames"] == N * F
    assert st["frame_steps"] <= 1.15 * N * F / 64.0, st
    d = str(path) + ".int8w"
    pp.write_pack(d, spec, pp.int8_model(pp.read_pack(path)[1]))
    o8 = oracle_mod.Oracle(d, int8=True)
    of = oracle_mod.Oracle(path)
    check = {0, 1, 2, F - 2, F - 1}
    worst = worst_f = 0.0
    for i in (5, 300, N - 1):
        r8 = _replay_selected(o8, ids[i], voices[i], res[i]["latents"], 10, check)
        rf = _replay_selected(of, ids[i], voices[i], res[i]["latents"], 10, check)
        for k in check:
            worst = max(worst, float(np.abs(res[i]["latents"][k] - r8[k][0]).max()))
            worst_f = max(worst_f, float(np.abs(res[i]["latents"][k] - rf[k][0]).max()))
    _note(f"int8 full size, 512 ragged utterances through the table: frame steps {st['frame_steps']} (ideal {N * F / 64:.0f}), admissions "
          f"{st['admissions']}, decoder calls {st['decode_calls']}; sampled per-frame latent max-abs vs int8 oracle {worst:.3e}, vs f32 oracle {worst_f:.3e}")
    assert worst <= 1.5 * worst_f and worst <= 3e-1
    eng.close()


def test_two_engines_behind_one_handle(packs, oracle_mod, monkeypatch):
    """POCKET_TTS_DEVICES lists the GPUs of one handle; listing GPU 0 twice builds two engines on the one GPU of the test box
    (the second runs its flow head as a launch chain: one persistent head per GPU).  A queued batch is dealt out over both,
    single calls go to the engine with fewer utterances pending, and every utterance equals the oracle's replay."""
    import ptts_lib as ptts
    monkeypatch.setenv("POCKET_TTS_DEVICES", "0,0")
    path, spec, _ = packs["tiny_eos"]
    F = 40
    eng = ptts.Engine(path, precision="fp32", temperature=0.0, lsd_steps=4, max_frames=F)
    o = oracle_mod.Oracle(path)
    rng = np.random.default_rng(77)
    ids, voices = _ragged(spec, rng, 41)
    res = eng.generate_ids(ids, voices, queued=True)
    served = eng.engine_served()
    assert len(served) == 2 and sorted(served) == [20, 21], served
    for i in range(41):
        lat, eos, frames = oracle_replay(o, ids[i], voices[i], res[i]["latents"], 4, F)
        assert res[i]["frames"] == frames, i
        if frames:
            assert np.abs(res[i]["latents"] - lat).max() < 1e-3
    # concurrent single calls spread over both engines
    out = [None] * 8

    def one(i):
        out[i] = eng.generate_ids([ids[i]], [voices[i]])[0]
    th = [threading.Thread(target=one, args=(i,)) for i in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(8):   # each against the oracle on its own history (free-running runs of one utterance may part ways)
        lat, eos, frames = oracle_replay(o, ids[i], voices[i], out[i]["latents"], 4, F)
        assert out[i]["frames"] == frames, i
        if frames:
            assert np.abs(out[i]["latents"] - lat).max() < 1e-3
    after = eng.engine_served()
    assert sum(after) == 49 and min(after) > min(served), after
    eng.close()


@one_engine
def test_more_callers_than_slots_mixed_traffic(packs):
    """96 threads on one handle at once, more than the table has slots: streams with chunk sizes 1 to 20, single offline calls
    and a few queued batches.  Every call returns (the test's time limit is the deadlock detector), chunks arrive in the
    reference's schedule (full chunks, then at most one flagged tail), a stream's samples equal frames x samples per frame,
    and afterwards the table is idle and serves a call as before.  The pack never raises EOS, so every utterance has
    max_frames frames whatever company it is prefilled in (on the packs that stop early, rounding differences between
    prefill passes of different sizes move a free-running utterance's stop frame; frame counts on those are checked with
    the oracle on the same history in the tests above)."""
    import ptts_lib as ptts
    path, spec, _ = packs["tiny"]
    F = 47
    eng = ptts.Engine(path, precision="fp32", temperature=0.0, lsd_steps=4, max_frames=F)
    rng = np.random.default_rng(404)
    ids, voices = _ragged(spec, rng, 96)
    ref = eng.generate_ids(ids, voices, queued=True, want_latents=False)
    frames = [r["frames"] for r in ref]
    errs, done = [], [0]
    lock = threading.Lock()
    gate = threading.Barrier(96)

    def stream(i, chunk):
        total, chunks, _ = eng.streaming_ids(ids[i], voices[i], chunk_frames=chunk)
        assert total == frames[i] * spec.frame_size, (i, total, frames[i])
        sizes = [len(c) // spec.frame_size for c, _ in chunks]
        flags = [f for _, f in chunks]
        assert sum(sizes) == frames[i]
        assert all(s == chunk for s in sizes[:-1]) and all(f == 0 for f in flags[:-1]), (i, chunk, sizes, flags)
        if sizes:
            assert (flags[-1] == 1) == (sizes[-1] < chunk), (i, chunk, sizes, flags)   # only a tail remainder is flagged final

    def offline(i):
        r = eng.generate_ids([ids[i]], [voices[i]], want_latents=False)[0]
        assert r["frames"] == frames[i] and len(r["audio"]) == frames[i] * spec.frame_size

    def batch(i):
        rs = eng.generate_ids(ids[i:i + 5], voices[i:i + 5], queued=True, want_latents=False)
        assert [r["frames"] for r in rs] == frames[i:i + 5]

    def run(i):
        try:
            gate.wait()
            if i % 3 == 0:
                stream(i, [1, 2, 5, 7, 15, 20][(i // 3) % 6])
            elif i % 16 == 1:
                batch(min(i, 90))
            else:
                offline(i)
            with lock:
                done[0] += 1
        except Exception as e:   # noqa: BLE001
            errs.append((i, repr(e)))

    th = [threading.Thread(target=run, args=(i,)) for i in range(96)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs[:3]
    assert done[0] == 96
    st = eng.scheduler_stats()
    assert st["largest"] == 64                      # the table filled up and the rest waited for slots
    again = eng.generate_ids([ids[0]], [voices[0]], want_latents=False)[0]
    assert again["frames"] == frames[0]
    eng.close()


def test_latents_of_chunks_that_run_over_the_end_of_the_ring(packs, oracle_mod):
    """Four tables' worth of utterances that all run to max_frames = 25: every wave joins the moment the table is empty,
    at steps 0, 25, 50 and 75 of the clock, and the per-slot latent ring has 60 rows, so the third wave's first decoder
    chunk covers ring rows 50..59 and 0..4.  Latents, EOS logits and audio of every utterance against the oracle."""
    import ptts_lib as ptts
    path, spec, _ = packs["tiny"]
    F, N = 25, 256
    eng = ptts.Engine(path, precision="fp32", temperature=0.0, lsd_steps=4, max_frames=F)
    o = oracle_mod.Oracle(path)
    rng = np.random.default_rng(77)
    ids, voices = _ragged(spec, rng, N)
    res = eng.generate_ids(ids, voices, queued=True)
    st = eng.scheduler_stats()
    worst = 0.0
    for i in range(N):
        lat, eos, frames = oracle_replay(o, ids[i], voices[i], res[i]["latents"], 4, F)
        assert res[i]["frames"] == frames == F, (i, res[i]["frames"], frames)
        assert res[i]["latents"].shape[0] == F
        worst = max(worst, float(np.abs(res[i]["eos"] - eos).max()), float(np.abs(res[i]["latents"] - lat).max()))
    assert worst < 1e-3
    for i in (0, 63, 64, 128, 129, 191, 192, N - 1):
        assert snr_db(oracle_decode(o, res[i]["latents"]), res[i]["audio"]) >= 40.0, i
    assert st["frame_steps"] == 4 * F, st   # no step of the clock was spent waiting for a chunk boundary
    eng.close()


_GIVE_UP_SCRIPT = r"""
import os, sys, time
import numpy as np
root = sys.argv[1]
for p in ("pocket-tts.cpp_amd/pyhost", "oracle", "tests"):
    sys.path.insert(0, os.path.join(root, p))
import ptts_lib as ptts, ptts_pack as pp, oracle_py
from conftest import oracle_replay
path = sys.argv[2]
spec = pp.Spec.full()
os.environ["FLOW_RING_TEST_DROP_WG"] = "1"
F = 6
eng = ptts.Engine(path, precision="bf16", temperature=0.0, lsd_steps=10, max_frames=F)
rng = np.random.default_rng(21)
ids = [rng.integers(0, spec.vocab, int(rng.integers(1, 13))).astype(np.int64) for _ in range(3)]
voices = [(rng.standard_normal((int(rng.integers(1, 21)), spec.d_model)) * 0.5).astype(np.float32) for _ in range(3)]
t0 = time.perf_counter()
try:
    eng.generate_ids(ids, voices, queued=True)
    print("NO ERROR RAISED"); sys.exit(1)
except ptts.PttsError as e:
    took = time.perf_counter() - t0
    assert "hand-off between workgroups timed out" in str(e), str(e)
assert took < 30.0, took
del os.environ["FLOW_RING_TEST_DROP_WG"]
res = eng.generate_ids(ids, voices, queued=True)     # launch chain from here on
d = os.path.join(os.path.dirname(path), "full_bf16w_giveup.ptts")
pp.write_pack(d, spec, pp.bf16_model(pp.read_pack(path)[1]))
o = oracle_py.Oracle(d)
worst = 0.0
for i in range(3):
    assert res[i]["frames"] == F and np.isfinite(res[i]["audio"]).all()
    lat, eos, frames = oracle_replay(o, ids[i], voices[i], res[i]["latents"], 10, F)
    worst = max(worst, float(np.abs(lat - res[i]["latents"]).max()))
assert worst <= 2e-2, worst
eng.close()
print(f"GIVE-UP OK took={took:.2f} worst={worst:.3e}")
"""


def test_flow_head_hand_off_that_gives_up_fails_the_table_and_not_the_queue(full_pack):
    """The persistent flow head launched one workgroup short (FLOW_RING_TEST_DROP_WG): the consumers of the missing producer
    spin to their limit and raise the error word, the launch drains, and the host sees the word with the next frame's EOS
    logits.  The call must come back with an error well inside the wait limit (no hang, no audio made from stale latents);
    the engine then runs the head as a launch chain, so the next call on the same handle succeeds and matches the oracle.
    Runs in a process of its own: only the first engine a process creates on a GPU holds the persistent head."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path, spec = full_pack
    r = subprocess.run([sys.executable, "-c", _GIVE_UP_SCRIPT, root, path], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "GIVE-UP OK" in r.stdout, r.stdout + r.stderr
    _note("flow-head give-up (own process): " + r.stdout.strip().splitlines()[-1])


def test_streaming_hazards_from_a_cpp_caller(full_pack, tmp_path):
    """tests/cpp/stream_hazards.cpp against the library: a chunk callback that throws (the exception reaches the caller, the
    handle serves the next stream), and cancelStreaming() with four streams running followed at once by a fifth (the four
    end early, the fifth completes)."""
    import shutil
    import subprocess
    import ptts_lib as ptts
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "_ref", "stream_hazards")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/stream_hazards not built (make -C oracle own_tests)")
    path, spec = full_pack
    models = tmp_path / "models"
    models.mkdir()
    os.symlink(path, models / "pocket_tts.ptts")
    tok = os.path.join(root, "tests", "golden", "sp_unigram.model")
    t = np.arange(2 * 24000) / 24000.0
    ptts.audio_save_wav(str(tmp_path / "voice.wav"), (0.3 * np.sin(2 * np.pi * 200.0 * t)).astype(np.float32), 24000)
    pkg = os.path.join(root, "pocket-tts.cpp_amd")
    env = dict(os.environ, LD_LIBRARY_PATH=pkg + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, str(models), tok, str(tmp_path / "voice.wav")], env=env, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    _note("stream_hazards: " + " | ".join(r.stdout.strip().splitlines()[:-1]))
