"""CPU-only tests of the checker itself and of the multi-rank host logic.

The C oracle is compared with (1) the committed vectors in tests/golden/tiny_network.npz, (2) an
independent float64 torch transcription of the same five graphs (tests/torch_ref.py) that evaluates whole
sequences at once, so agreement also proves the oracle's streaming state (KV cache, conv history, ring
buffer) is equivalent to the stateless computation.  The network arithmetic itself has no reference output
to pin against (the ONNX graphs and their runtime are not in the reference tree): parity unpinned.
"""
import os
import subprocess
import sys
import textwrap
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def tiny(packs, oracle_mod):
    path, spec, w = packs["tiny"]
    return oracle_mod.Oracle(path), spec, w


def test_oracle_reproduces_golden_vectors(tiny):
    o, spec, w = tiny
    g = np.load(os.path.join(GOLD, "tiny_network.npz"))
    assert zlib.crc32(b"".join(w[k].tobytes() for k in sorted(w))) == int(g["pack_crc32"][0])
    np.testing.assert_allclose(o.text_conditioner(g["ids"]), g["text_emb"], rtol=0, atol=0)
    np.testing.assert_allclose(o.flow_euler(g["cond"], g["x0"], 10), g["flow10"], rtol=0, atol=2e-6)
    r = o.generate(g["ids"], g["voice"], lsd_steps=10, max_frames=8)
    assert r["frames"] == len(g["latents"]) == 8
    # different OpenMP reduction order only; the 8-frame AR loop amplifies 1e-7 to ~1e-5
    np.testing.assert_allclose(r["latents"], g["latents"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(r["eos"], g["eos"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(r["audio"][: len(g["audio"])], g["audio"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(o.mimi_encode(g["clip"]), g["enc"], rtol=0, atol=1e-5)


def test_oracle_matches_float64_transcription(tiny):
    import torch
    from torch_ref import TorchRef
    o, spec, w = tiny
    t = TorchRef(spec, w)
    rng = np.random.default_rng(3)
    ids = rng.integers(0, spec.vocab, 7).astype(np.int64)
    voice = (rng.standard_normal((4, spec.d_model)) * 0.5).astype(np.float32)
    text = o.text_conditioner(ids)
    np.testing.assert_array_equal(text, t.text_conditioner(ids).numpy().astype(np.float32))
    # flow_lm_main: prefill in two calls + 5 single-row steps  ==  one whole-sequence pass
    seq = (rng.standard_normal((5, spec.latent_dim)) * 0.7).astype(np.float32)
    seq[0] = np.nan
    st = o.lm_state(64)
    o.lm_main(st, None, voice)
    o.lm_main(st, None, text)
    cond, eos = zip(*[o.lm_main(st, seq[i:i + 1], None) for i in range(len(seq))])
    tc, te = t.lm_full(voice, text, seq)
    assert np.abs(np.array(cond).reshape(5, -1) - tc.numpy()).max() < 2e-5
    assert np.abs(np.array(eos).ravel() - te.numpy()).max() < 2e-5
    # flow head
    c = rng.standard_normal(spec.d_model).astype(np.float32)
    x0 = rng.standard_normal(spec.latent_dim).astype(np.float32)
    assert np.abs(o.flow(c, 0.3, 0.4, x0) - t.flow(c, 0.3, 0.4, x0).numpy()).max() < 2e-5
    assert np.abs(o.flow_euler(c, x0, 10) - t.flow_euler(c, x0, 10).numpy()).max() < 2e-5
    # decoder: chunks of 3 + 1 + 4 frames with carried state  ==  one stateless pass
    lat = (rng.standard_normal((8, spec.latent_dim)) * 0.8).astype(np.float32)
    ds = o.dec_state()
    audio = np.concatenate([o.mimi_decode(ds, lat[a:b]) for a, b in ((0, 3), (3, 4), (4, 8))])
    ta = t.mimi_decode_full(lat).numpy()
    assert audio.shape == ta.shape == (8 * spec.frame_size,)
    assert np.abs(audio - ta).max() < 5e-5 * max(1.0, np.abs(ta).max())
    # encoder
    clip = (rng.standard_normal(5 * spec.frame_size + 7) * 0.3).astype(np.float32)
    enc = o.mimi_encode(clip)
    tenc = t.mimi_encode(clip).numpy()
    assert enc.shape == tenc.shape
    assert np.abs(enc - tenc).max() < 5e-5 * max(1.0, np.abs(tenc).max())


def test_oracle_stop_rule_and_chunking(packs, oracle_mod):
    """EOS threshold -4.0, three frames after the first EOS, the EOS step's own frame is kept
    (src/pocket_tts.cpp:596-603); decoding in 15-frame chunks equals decoding in 4-frame chunks."""
    path, spec, _ = packs["tiny_eos"]
    o = oracle_mod.Oracle(path)
    rng = np.random.default_rng(0)
    hit = False
    for trial in range(6):
        ids = rng.integers(0, spec.vocab, 6).astype(np.int64)
        voice = (rng.standard_normal((3, spec.d_model)) * 0.5).astype(np.float32)
        r = o.generate(ids, voice, lsd_steps=4, max_frames=40)
        eos = r["eos"]
        first = np.flatnonzero(eos > -4.0)
        if len(first) and first[0] + 3 < 40:
            hit = True
            assert r["frames"] == first[0] + 3 and len(eos) == first[0] + 4
        else:
            assert r["frames"] == 40
        assert len(r["audio"]) == r["frames"] * spec.frame_size
        r2 = o.generate(ids, voice, lsd_steps=4, max_frames=40, chunk_frames=4)
        assert r2["frames"] == r["frames"]
        assert np.abs(r2["audio"] - r["audio"]).max() < 1e-5
    assert hit, "no trial crossed the EOS threshold: the tiny_eos pack no longer exercises the stop rule"


def test_oracle_int8_mode_is_quantisation_noise_away(packs, oracle_mod):
    """int8 mode (LM-layer matrices int8, activations quantised per 64-wide block): close to, but not equal to, the
    float model; and switching it on for one oracle object leaves another object on the same pack untouched."""
    path, spec, _ = packs["tiny"]
    of, o8 = oracle_mod.Oracle(path), oracle_mod.Oracle(path, int8=True)
    rng = np.random.default_rng(9)
    emb = (rng.standard_normal((6, spec.d_model)) * 0.5).astype(np.float32)
    seq = np.full((1, spec.latent_dim), np.nan, np.float32)

    def cond_of(o):
        st = o.lm_state(32)
        o.lm_main(st, None, emb)
        return o.lm_main(st, seq, None)[0]

    cf, c8 = cond_of(of), cond_of(o8)
    d = np.abs(cf - c8).max()
    assert 1e-4 < d < 0.05 * np.abs(cf).max() + 0.05
    o8.close()
    np.testing.assert_array_equal(cond_of(of), cf)   # the registry entry of the closed object is gone
    of2 = oracle_mod.Oracle(path)
    np.testing.assert_array_equal(cond_of(of2), cf)


def test_oracle_dynamic_uint8_product_follows_the_onnx_operators(oracle_mod):
    """The checker's second int8 activation scheme against the operator definitions written out in numpy: identical
    integers, so the results agree to the last float32 rounding of the one scaling multiply.  Rows cover: mixed sign,
    all positive (zero point 0), all negative (zero point 255), all zero (scale 0), one huge outlier."""
    import q8_schemes
    rng = np.random.default_rng(3)
    for (M, K, N) in [(7, 128, 48), (3, 1024, 64), (16, 256, 32)]:
        A = (rng.standard_normal((M, K)) * rng.uniform(0.1, 3.0, (M, 1))).astype(np.float32)
        A[0] = np.abs(A[0])
        A[1] = -np.abs(A[1])
        A[2] = 0.0
        if M > 3:
            A[3, 5] = 250.0
        W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        want, q, zp, s = q8_schemes.dynamic_uint8_product(A, W)
        assert zp[0] == 0 and zp[1] == 255 and s[2] == 0 and q.min() >= 0 and q.max() <= 255
        got = oracle_mod.q8_linear(A, W, scheme=1)
        assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max()), (M, K, N)
        assert np.all(got[2] == 0)
        # and it is a different scheme from the per-block one: same weights, other activation grid
        blk = oracle_mod.q8_linear(A, W, scheme=0)
        exact = A.astype(np.float64) @ W.astype(np.float64).T
        assert np.abs(blk - got).max() > 0
        assert np.abs(got - exact).max() < 0.05 * np.abs(exact).max() and np.abs(blk - exact).max() < 0.05 * np.abs(exact).max()


def test_oracle_int8_scheme_comes_from_the_pack_header(packs, oracle_mod, tmp_path):
    """A pack whose header says q8_act_scheme = 1 makes the int8 checker use the dynamic uint8 activations (LM layers
    only; the flow blocks stay bf16 there), a header of 0 the per-block scheme; two such objects live side by side."""
    path, spec, w = packs["tiny"]
    import dataclasses
    import ptts_pack as pp
    spec1 = dataclasses.replace(spec, q8_act_scheme=1)
    p1 = str(tmp_path / "dyn.ptts")
    pp.write_pack(p1, spec1, w)
    assert pp.read_pack(p1)[0].q8_act_scheme == 1 and pp.read_pack(path)[0].q8_act_scheme == 0
    of, o0, o1 = oracle_mod.Oracle(path), oracle_mod.Oracle(path, int8=True), oracle_mod.Oracle(p1, int8=True)
    rng = np.random.default_rng(19)
    emb = (rng.standard_normal((6, spec.d_model)) * 0.5).astype(np.float32)
    seq = np.full((1, spec.latent_dim), np.nan, np.float32)

    def cond_of(o):
        st = o.lm_state(32)
        o.lm_main(st, None, emb)
        return o.lm_main(st, seq, None)[0]

    cf, c0, c1 = cond_of(of), cond_of(o0), cond_of(o1)
    assert np.abs(c0 - c1).max() > 1e-5
    for c in (c0, c1):
        assert 1e-4 < np.abs(cf - c).max() < 0.05 * np.abs(cf).max() + 0.05
    np.testing.assert_array_equal(cond_of(o0), c0)   # creating the other object did not change this one's scheme


# ----------------------------------------------------------------------------- multi-rank host logic

def test_partition_is_balanced_and_deterministic():
    import ptts_shard
    costs = [125, 40, 300, 300, 10, 77, 125, 125, 500, 3]
    for world in (1, 2, 3, 8):
        parts = ptts_shard.partition(costs, world)
        assert sorted(sum(parts, [])) == list(range(len(costs)))
        loads = [sum(costs[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(costs)
        assert parts == ptts_shard.partition(list(costs), world)
    assert ptts_shard.partition([], 2) == [[], []]
    b = ptts_shard.batches([0, 1, 2, 3, 4], costs, 2)
    assert b == [[2, 3], [0, 1], [4]]
    with pytest.raises(ValueError):
        ptts_shard.partition(costs, 0)


def test_two_rank_job_reduction_over_gloo(tmp_path):
    """world_size 2 on CPU: both ranks derive the same partition, the reported wall time is the slower
    rank's and the units are summed - the contract bench.py's aggregate line relies on."""
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(f"""
        import json, os, sys, time
        sys.path.insert(0, {os.path.join(ROOT, 'pocket-tts.cpp_amd', 'pyhost')!r})
        import torch, torch.distributed as dist
        import ptts_shard
        rank, local_rank, world = ptts_shard.env_ranks()
        dist.init_process_group("gloo")
        costs = [125, 40, 300, 300, 10, 77, 125]
        mine = ptts_shard.partition(costs, world)[rank]
        ptts_shard.barrier(dist, torch.device("cpu"))
        t0 = time.perf_counter()
        time.sleep(0.05 * (rank + 1))
        ptts_shard.barrier(dist, torch.device("cpu"))
        dt = 0.1 * (rank + 1)
        dt_max, units, ranks = ptts_shard.reduce_job(dt, sum(costs[i] for i in mine), dist, torch.device("cpu"))
        json.dump(dict(rank=rank, mine=mine, dt_max=dt_max, units=units, ranks=ranks), open({str(tmp_path)!r} + f"/out{{rank}}.json", "w"))
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29541", str(script)], check=True, env=env, timeout=240, capture_output=True)
    import json
    r0, r1 = (json.load(open(tmp_path / f"out{r}.json")) for r in (0, 1))
    assert sorted(r0["mine"] + r1["mine"]) == list(range(7)) and not set(r0["mine"]) & set(r1["mine"])
    assert r0["dt_max"] == r1["dt_max"] == pytest.approx(0.2)
    assert r0["units"] == r1["units"] == 977.0
    assert r0["ranks"] == r1["ranks"] == 2


def test_spawn_plan_exports_what_a_launcher_would():
    import ptts_shard
    plan = ptts_shard.spawn_plan(4, ["bench.py", "--gpus", "4"], 29777, base_env={"KEEP": "1", "RANK": "9"})
    assert len(plan) == 4
    for r, (env, cmd) in enumerate(plan):
        assert env["RANK"] == env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29777" and env["KEEP"] == "1"
        assert cmd[0] == sys.executable and cmd[1:] == ["bench.py", "--gpus", "4"]
    with pytest.raises(ValueError):
        ptts_shard.spawn_plan(0, [], 1)


def test_launch_ranks_runs_every_rank_and_relays_rank_zero(tmp_path, capfd):
    """The launcher behind `bench.py --gpus N`: two ranks over gloo here (no GPU is opened), rank 0's line comes back on
    standard output; a failing rank makes the launcher return non-zero and stops the other."""
    import ptts_shard
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(f"""
        import json, os, sys, time
        sys.path.insert(0, {os.path.join(ROOT, 'pocket-tts.cpp_amd', 'pyhost')!r})
        import torch.distributed as dist
        import ptts_shard
        rank, local_rank, world = ptts_shard.env_ranks()
        if len(sys.argv) > 1 and sys.argv[1] == "fail" and rank == 1:
            sys.exit(3)
        if len(sys.argv) > 1 and sys.argv[1] == "fail":
            time.sleep(60)
        dist.init_process_group("gloo")
        dt_max, units, ranks = ptts_shard.reduce_job(0.5 + rank, 10.0, dist)
        if rank == 0:
            print(json.dumps(dict(n_gpus=ranks, units=units, dt_max=dt_max)), flush=True)
        dist.destroy_process_group()
    """))
    assert ptts_shard.launch_ranks(2, [str(script)], timeout_s=200, gpus_visible=2) == 0
    out = capfd.readouterr().out.strip().splitlines()[-1]
    import json
    assert json.loads(out) == dict(n_gpus=2, units=20.0, dt_max=1.5)
    assert ptts_shard.launch_ranks(2, [str(script), "fail"], timeout_s=200, gpus_visible=2) == 1
    assert "rank 1 exited with 3" in capfd.readouterr().err


def test_bench_refuses_more_ranks_than_visible_gpus():
    """`python bench.py --gpus 2` with fewer than two GPUs visible fails loudly instead of timing one GPU."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--size", "tiny"], capture_output=True, text=True, timeout=240,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 2 and "64 GPUs requested" in r.stderr and r.stdout.strip() == ""
