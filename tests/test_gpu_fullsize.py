"""GPU parity at the full model size (d_model 1024, 6 layers, flow 512x6, Mimi 512, 1920 samples/frame),
seeded synthetic weights, through the C ABI.  fp32: per-frame latents max-abs <= 1e-3 and PCM SNR >= 40 dB
against the oracle.  bf16: per-frame latents max-abs <= 2e-2.  Also the committed golden fixture, which needs
no oracle at run time."""
import os

import numpy as np
import pytest

from conftest import oracle_decode, oracle_replay, snr_db

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = 5


@pytest.fixture(scope="module")
def full(tmp_path_factory, oracle_mod):
    import ptts_lib
    import ptts_pack as pp
    ptts_lib.load()
    d = tmp_path_factory.mktemp("full")
    spec = pp.Spec.full()
    path = str(d / "full.ptts")
    w = pp.make_pack(path, spec, seed=1234, eos_bias=-20.0)
    path_q = str(d / "full_bf16w.ptts")
    pp.write_pack(path_q, spec, pp.bf16_model(w))
    path_i8 = str(d / "full_int8w.ptts")
    pp.write_pack(path_i8, spec, pp.int8_model(w))
    rng = np.random.default_rng(77)
    ids = [rng.integers(0, spec.vocab, t).astype(np.int64) for t in (32, 11, 20)]
    voices = [(rng.standard_normal((v, spec.d_model)) * 0.5).astype(np.float32) for v in (62, 30, 45)]
    return dict(lib=ptts_lib, spec=spec, path=path, path_q=path_q, ids=ids, voices=voices,
                o=oracle_mod.Oracle(path), oq=oracle_mod.Oracle(path_q), oi8=oracle_mod.Oracle(path_i8, int8=True))


def _note(line):
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/fullsize_parity.txt", "a") as f:
            f.write(line + "\n")


def test_full_fp32_offline_parity(full):
    eng = full["lib"].Engine(full["path"], precision="fp32", temperature=0.0, lsd_steps=10, max_frames=F)
    res = eng.generate_ids(full["ids"], full["voices"])
    worst, worst_snr = 0.0, 1e9
    for i in range(3):
        lat, eos, frames = oracle_replay(full["o"], full["ids"][i], full["voices"][i], res[i]["latents"], 10, F)
        assert res[i]["frames"] == frames == F
        assert res[i]["audio"].shape[0] == F * 1920
        worst = max(worst, float(np.abs(res[i]["latents"] - lat).max()), float(np.abs(res[i]["eos"] - eos).max()))
        worst_snr = min(worst_snr, snr_db(oracle_decode(full["o"], res[i]["latents"]), res[i]["audio"]))
    _note(f"fp32 full size: per-frame latent/eos max-abs {worst:.3e}, PCM SNR {worst_snr:.1f} dB")
    assert worst <= 1e-3 and worst_snr >= 40.0
    eng.close()


def test_full_bf16_offline_parity(full):
    """Per-frame (same history on both sides) latent error of the bf16 engine's ARITHMETIC: against the oracle
    evaluating the same bf16-rounded weights in float32, bound 2e-2.  The distance to the float32 model (weight
    rounding included) is what the engine is specified to (2e-2); this random-weight model does not meet it
    (4.2e-2 measured), which tests/test_gpu_configs.py reports as an expected failure with the bound unwidened.
    The 5e-2 below is a regression guard at the measured level, not the specification."""
    eng = full["lib"].Engine(full["path"], precision="bf16", temperature=0.0, lsd_steps=10, max_frames=F)
    res = eng.generate_ids(full["ids"], full["voices"])
    worst_q = worst_f = 0.0
    snr_q = 1e9
    for i in range(3):
        latq, _, frames = oracle_replay(full["oq"], full["ids"][i], full["voices"][i], res[i]["latents"], 10, F)
        latf, _, _ = oracle_replay(full["o"], full["ids"][i], full["voices"][i], res[i]["latents"], 10, F)
        assert res[i]["frames"] == frames == F
        worst_q = max(worst_q, float(np.abs(res[i]["latents"] - latq).max()))
        worst_f = max(worst_f, float(np.abs(res[i]["latents"] - latf).max()))
        snr_q = min(snr_q, snr_db(oracle_decode(full["oq"], res[i]["latents"]), res[i]["audio"]))
    _note(f"bf16 full size: per-frame latent max-abs vs oracle on bf16-rounded weights {worst_q:.3e}, "
          f"vs oracle on f32 weights {worst_f:.3e}, PCM SNR vs oracle(bf16 weights) {snr_q:.1f} dB")
    assert worst_q <= 2e-2, worst_q
    assert worst_f <= 5e-2, worst_f
    assert snr_q >= 30.0
    eng.close()


def test_full_batch64_bench_shape(full):
    """The bench shape (64 utterances, 32 ids, 62 voice rows, bf16) for 6 frames: (1) two runs are bit-identical,
    (2) utterances 0, 17 and 63 replayed by the oracle on the same bf16-rounded weights stay inside the bf16 bound,
    (3) an utterance generated alone (other kernel variants: one row per row group) gives the same first frame to
    the bf16 bound, (4) ragged lengths: a batch whose utterances differ in text and voice length matches the same
    utterances generated in a uniform batch of their own."""
    spec = full["spec"]
    Fb = 6
    eng = full["lib"].Engine(full["path"], precision="bf16", temperature=0.0, lsd_steps=10, max_frames=Fb)
    ids, voices = [], []
    for i in range(64):
        r = np.random.default_rng(500 + i)
        ids.append(r.integers(0, spec.vocab, 32).astype(np.int64))
        voices.append((r.standard_normal((62, spec.d_model)) * 0.5).astype(np.float32))
    a = eng.generate_ids(ids, voices, want_audio=False)
    b = eng.generate_ids(ids, voices, want_audio=False)
    for i in range(64):
        assert a[i]["frames"] == Fb
        np.testing.assert_array_equal(a[i]["latents"], b[i]["latents"])
    worst = 0.0
    for i in (0, 17, 63):
        lat, _, frames = oracle_replay(full["oq"], ids[i], voices[i], a[i]["latents"], 10, Fb)
        assert frames == Fb
        worst = max(worst, float(np.abs(a[i]["latents"] - lat).max()))
    alone = eng.generate_ids(ids[17:18], voices[17:18], want_audio=False)[0]
    first = float(np.abs(alone["latents"][0] - a[17]["latents"][0]).max())
    # ragged batch: utterance k keeps 8 + k % 25 ids and 20 + (7 k) % 43 voice rows
    rid = [ids[k][: 8 + k % 25] for k in range(64)]
    rvo = [voices[k][: 20 + (7 * k) % 43] for k in range(64)]
    rag = eng.generate_ids(rid, rvo, want_audio=False)
    worst_r = 0.0
    for k in (3, 40, 62):
        lat, _, frames = oracle_replay(full["oq"], rid[k], rvo[k], rag[k]["latents"], 10, Fb)
        assert frames == Fb
        worst_r = max(worst_r, float(np.abs(rag[k]["latents"] - lat).max()))
    _note(f"bf16 full size, 64 utterances: per-frame latent max-abs vs oracle(bf16 weights) {worst:.3e}, ragged batch {worst_r:.3e}, "
          f"first frame alone vs in the batch {first:.3e}")
    assert worst <= 2e-2 and worst_r <= 2e-2 and first <= 2e-2
    eng.close()


@pytest.mark.parametrize("q8_flow", [False, True])
def test_full_int8_graphs_and_offline_parity(full, oracle_mod, tmp_path, q8_flow):
    """int8 engine (int8 LM-layer matrices on v_mfma_i32, block-quantised activations; q8_flow: the flow blocks in int8 as
    well, POCKET_TTS_Q8_FLOW=1, the persistent head's int8 form) against the oracle running the same quantisation scheme on
    the same weights: graph by graph, then per frame."""
    import ptts_pack as pp
    spec = full["spec"]
    if q8_flow:
        _, w = pp.read_pack(full["path"])
        p_chk = str(tmp_path / "int8_flow.ptts")
        pp.write_pack(p_chk, spec, pp.int8_model(w, flow_blocks=True))
        o8 = oracle_mod.Oracle(p_chk, int8=True, q8_flow=True)
        os.environ["POCKET_TTS_Q8_FLOW"] = "1"
    else:
        o8 = full["oi8"]
    try:
        eng = full["lib"].Engine(full["path"], precision="int8", temperature=0.0, lsd_steps=10, max_frames=F)
    finally:
        os.environ.pop("POCKET_TTS_Q8_FLOW", None)
    rng = np.random.default_rng(5)
    # flow head alone
    cond = rng.standard_normal((3, spec.d_model)).astype(np.float32)
    x0 = (rng.standard_normal((3, spec.latent_dim)) * 0.8).astype(np.float32)
    got = eng.flow_euler(cond, x0)
    want = np.stack([o8.flow_euler(cond[i], x0[i], 10) for i in range(3)])
    e_flow = float(np.abs(got - want).max())
    # LM: prefill + 3 decode steps on two slots
    st = eng.state(2)
    ost = [o8.lm_state(128), o8.lm_state(128)]
    e_lm = 0.0
    for slot in range(2):
        emb = (rng.standard_normal((20 + 7 * slot, spec.d_model)) * 0.5).astype(np.float32)
        eng.lm_prefill(st, slot, emb)
        o8.lm_main(ost[slot], None, emb)
    seq = np.full((2, spec.latent_dim), np.nan, np.float32)
    for _ in range(3):
        c, e = eng.lm_step(st, seq)
        for slot in range(2):
            oc, oe = o8.lm_main(ost[slot], seq[slot:slot + 1], None)
            e_lm = max(e_lm, float(np.abs(c[slot] - oc).max()), abs(float(e[slot]) - float(oe)))
        seq = (rng.standard_normal((2, spec.latent_dim)) * 0.7).astype(np.float32)
    # whole path, teacher-forced
    res = eng.generate_ids(full["ids"], full["voices"])
    worst = worst_f = 0.0
    for i in range(3):
        lat8, _, frames = oracle_replay(o8, full["ids"][i], full["voices"][i], res[i]["latents"], 10, F)
        latf, _, _ = oracle_replay(full["o"], full["ids"][i], full["voices"][i], res[i]["latents"], 10, F)
        assert res[i]["frames"] == frames == F
        worst = max(worst, float(np.abs(res[i]["latents"] - lat8).max()))
        worst_f = max(worst_f, float(np.abs(res[i]["latents"] - latf).max()))
    _note(f"int8 full size{' (int8 flow blocks)' if q8_flow else ''}: flow head max-abs {e_flow:.3e}, LM cond/eos max-abs {e_lm:.3e}, per-frame latent max-abs vs int8 oracle "
          f"{worst:.3e}, vs f32 oracle {worst_f:.3e}")
    # Dynamic activation quantisation is discontinuous: the int8 ORACLE moves by ~4e-2 (LM conditioning, magnitude
    # ~3.5) and ~1e-2 (flow head) when its inputs are perturbed by 1e-6, because a handful of roundings flip.  Two
    # correct implementations therefore agree only to that noise floor; the kernel arithmetic itself is checked
    # exactly in test_int8_product_is_exact.  The bounds below are regression guards at a few times that floor
    # (and "never further from the int8 oracle than int8 is from float"), NOT the 2e-2 the int8 engine is specified
    # to against the float32 model: that bound is kept, unwidened and as an expected failure (1.5e-1 measured here;
    # the residual-scaled pack and the share of each matrix family are in profiles/r03_int8_attribution.txt), in
    # tests/test_gpu_configs.py.
    assert e_flow <= 5e-2 and e_lm <= 1.5e-1
    assert worst <= 1.5 * worst_f and worst <= 3e-1
    eng.close()


def test_int8_prequantised_step_matches_staged(full):
    """The int8 decode step with activations quantised by their producers (add & norm, attention, FFN-1 epilogue)
    against the same step with the products quantising while they stage f32 rows (POCKET_TTS_Q8_STAGED=1): the
    scheme and the order of the float operations are the same, so the conditioning and EOS outputs agree to the last
    bits, on 64 rows with ragged contexts over three steps."""
    spec = full["spec"]
    rng = np.random.default_rng(31)
    embs = [(rng.standard_normal((5 + (11 * i) % 40, spec.d_model)) * 0.5).astype(np.float32) for i in range(64)]
    seqs = [np.full((64, spec.latent_dim), np.nan, np.float32)] + [(rng.standard_normal((64, spec.latent_dim)) * 0.7).astype(np.float32) for _ in range(2)]
    outs = []
    for staged in (False, True):
        if staged:
            os.environ["POCKET_TTS_Q8_STAGED"] = "1"
        try:
            eng = full["lib"].Engine(full["path"], precision="int8", temperature=0.0, lsd_steps=10, max_frames=8)
            st = eng.state(64)
            for slot in range(64):
                eng.lm_prefill(st, slot, embs[slot])
            got = [eng.lm_step(st, sq) for sq in seqs]
            outs.append(got)
            st.close(); eng.close()
        finally:
            os.environ.pop("POCKET_TTS_Q8_STAGED", None)
    worst, differing = 0.0, []
    for (c0, e0), (c1, e1) in zip(*outs):
        worst = max(worst, float(np.abs(c0 - c1).max()), float(np.abs(e0 - e1).max()))
        differing.append(float((np.abs(c0 - c1) > 1e-6).mean()))
    _note(f"int8 full size: decode step with producer-side quantisation vs staged quantisation, max-abs {worst:.3e}, "
          f"share of conditioning elements that differ per step {differing}")
    assert worst <= 1e-5, worst


def test_int8_product_is_exact(full):
    """The int8 kernel against a numpy restatement of the scheme on identical inputs: every rounding decision is
    then the same, the block sums are integers, and only the float32 scaling differs in its last bits."""
    eng = full["lib"].Engine(full["path"], precision="int8", temperature=0.0, lsd_steps=1, max_frames=1)
    rng = np.random.default_rng(11)
    for (M, K, N) in [(64, 1024, 3072), (5, 256, 64), (70, 4096, 1024)]:
        A = (rng.standard_normal((M, K)) * rng.uniform(0.1, 3.0, (M, 1))).astype(np.float32)
        A[0, :64] = 0.0                                            # an all-zero block
        W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        mw = np.abs(W).max(axis=1)
        invw = np.where(mw > 0, np.float32(127.0) / mw, 0).astype(np.float32)
        qw = np.clip(np.rint(W * invw[:, None]), -127, 127).astype(np.int32)
        sw = (mw / np.float32(127.0)).astype(np.float32)
        Ab = A.reshape(M, K // 64, 64)
        ma = np.abs(Ab).max(axis=2)
        inva = np.where(ma > 0, np.float32(127.0) / np.where(ma > 0, ma, 1), 0).astype(np.float32)
        qa = np.clip(np.rint(Ab * inva[:, :, None]), -127, 127).astype(np.int32)
        da = (ma / np.float32(127.0)).astype(np.float32)
        acc = np.zeros((M, N), np.float32)
        for kb in range(K // 64):
            isum = qa[:, kb, :] @ qw[:, kb * 64:(kb + 1) * 64].T
            acc = (acc + isum.astype(np.float32) * da[:, kb:kb + 1]).astype(np.float32)
        want = acc * sw[None, :]
        got = eng.q8_linear(A, W)
        assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max()), (M, K, N)
    eng.close()


def test_int8_dynamic_uint8_product_is_exact(full, oracle_mod):
    """The second int8 activation scheme (rows as uint8 with one scale and zero point over the whole row, as ONNX
    DynamicQuantizeLinear + MatMulInteger do it) against the operator definitions in numpy and against the checker:
    the integer sums are identical, only the one float32 scaling multiply can differ in its last bit.  Rows cover all
    positive, all negative, all zero and an outlier; shapes cover ragged row counts and the LM's three K / N pairs."""
    import q8_schemes
    eng = full["lib"].Engine(full["path"], precision="int8", temperature=0.0, lsd_steps=1, max_frames=1)
    rng = np.random.default_rng(12)
    for (M, K, N) in [(64, 1024, 3072), (5, 256, 64), (70, 4096, 1024), (1, 1024, 4096)]:
        A = (rng.standard_normal((M, K)) * rng.uniform(0.1, 3.0, (M, 1))).astype(np.float32)
        A[0] = np.abs(A[0])
        if M > 4:
            A[1] = -np.abs(A[1])
            A[2] = 0.0
            A[3, 5] = 250.0
        W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        want, _, zp, _ = q8_schemes.dynamic_uint8_product(A, W)
        assert zp[0] == 0
        got = eng.q8_linear(A, W, dynamic=True)
        tol = 2e-6 * max(1.0, np.abs(want).max())
        assert np.abs(got - want).max() <= tol, (M, K, N)
        if M * K * N <= 70 * 4096 * 1024:
            assert np.abs(got - oracle_mod.q8_linear(A, W, scheme=1)).max() <= tol, (M, K, N)
    eng.close()


def test_full_int8_dynamic_scheme_parity(full, oracle_mod, tmp_path):
    """A pack whose header selects the dynamic uint8 activation scheme (what tools/onnx_to_ptts.py writes for a graph
    that holds DynamicQuantizeLinear / MatMulInteger nodes): the int8 engine follows it (int8 LM layers with per-row
    uint8 activations, bf16 flow blocks) and is compared with the checker in the same scheme, graph by graph and per frame,
    and with the float32 checker.  POCKET_TTS_Q8_ACT=block on the same pack gives the per-block engine back."""
    import dataclasses
    import ptts_pack as pp
    spec = dataclasses.replace(full["spec"], q8_act_scheme=1)
    _, w = pp.read_pack(full["path"])
    p_dyn, p_chk = str(tmp_path / "dyn.ptts"), str(tmp_path / "dyn_int8w.ptts")
    pp.write_pack(p_dyn, spec, w)
    pp.write_pack(p_chk, spec, pp.int8_model(w, scheme=1))
    o8 = oracle_mod.Oracle(p_chk, int8=True)
    eng = full["lib"].Engine(p_dyn, precision="int8", temperature=0.0, lsd_steps=10, max_frames=F)
    rng = np.random.default_rng(5)
    # LM: prefill + 3 decode steps on two slots (the decode products run split along K)
    st = eng.state(2)
    ost = [o8.lm_state(128), o8.lm_state(128)]
    e_lm = 0.0
    for slot in range(2):
        emb = (rng.standard_normal((20 + 7 * slot, spec.d_model)) * 0.5).astype(np.float32)
        eng.lm_prefill(st, slot, emb)
        o8.lm_main(ost[slot], None, emb)
    seq = np.full((2, spec.latent_dim), np.nan, np.float32)
    for _ in range(3):
        c, e = eng.lm_step(st, seq)
        for slot in range(2):
            oc, oe = o8.lm_main(ost[slot], seq[slot:slot + 1], None)
            e_lm = max(e_lm, float(np.abs(c[slot] - oc).max()), abs(float(e[slot]) - float(oe)))
        seq = (rng.standard_normal((2, spec.latent_dim)) * 0.7).astype(np.float32)
    st.close()
    # flow head: bf16 blocks in this scheme, so it is the bf16 engine's head
    cond = rng.standard_normal((3, spec.d_model)).astype(np.float32)
    x0 = (rng.standard_normal((3, spec.latent_dim)) * 0.8).astype(np.float32)
    e_flow = float(np.abs(eng.flow_euler(cond, x0) - np.stack([o8.flow_euler(cond[i], x0[i], 10) for i in range(3)])).max())
    res = eng.generate_ids(full["ids"], full["voices"])
    worst = worst_f = 0.0
    for i in range(3):
        lat8, _, frames = oracle_replay(o8, full["ids"][i], full["voices"][i], res[i]["latents"], 10, F)
        latf, _, _ = oracle_replay(full["o"], full["ids"][i], full["voices"][i], res[i]["latents"], 10, F)
        assert res[i]["frames"] == frames == F
        worst = max(worst, float(np.abs(res[i]["latents"] - lat8).max()))
        worst_f = max(worst_f, float(np.abs(res[i]["latents"] - latf).max()))
    eng.close()
    os.environ["POCKET_TTS_Q8_ACT"] = "block"
    try:
        outs = []
        for pth in (p_dyn, full["path"]):      # one engine at a time: only one holds the one-launch flow head
            e2 = full["lib"].Engine(pth, precision="int8", temperature=0.0, lsd_steps=10, max_frames=F)
            outs.append(e2.flow_euler(cond, x0))
            e2.close()
        a, b = outs
    finally:
        os.environ.pop("POCKET_TTS_Q8_ACT", None)
    np.testing.assert_array_equal(a, b)
    _note(f"int8 full size, dynamic uint8 activations: flow head (bf16 blocks) max-abs {e_flow:.3e}, LM cond/eos max-abs {e_lm:.3e}, "
          f"per-frame latent max-abs vs int8 oracle {worst:.3e}, vs f32 oracle {worst_f:.3e}")
    # As with the per-block scheme, a rounding that flips moves the quantised model by about 1e-2, so two correct
    # implementations agree to that noise floor and not to float accuracy; the product itself is checked exactly above.
    assert e_flow <= 2e-2 and e_lm <= 1.5e-1
    assert worst <= 1.5 * worst_f and worst <= 3e-1


def test_int8_dynamic_step_with_producer_ranges_matches_scanning_products(full, tmp_path):
    """Dynamic uint8 scheme, decode step: the producers (add & norm, attention, FFN-1 epilogue) leave each row's (min, max)
    and the products quantise from those, against the same step with every product scanning its rows over the full K
    (POCKET_TTS_Q8_STAGED=1).  Minimum and maximum are exact, so the two agree bit for bit: 64 rows, ragged contexts,
    three steps."""
    import dataclasses
    import ptts_pack as pp
    spec = dataclasses.replace(full["spec"], q8_act_scheme=1)
    _, w = pp.read_pack(full["path"])
    p_dyn = str(tmp_path / "dyn.ptts")
    pp.write_pack(p_dyn, spec, w)
    rng = np.random.default_rng(32)
    embs = [(rng.standard_normal((5 + (11 * i) % 40, spec.d_model)) * 0.5).astype(np.float32) for i in range(64)]
    seqs = [np.full((64, spec.latent_dim), np.nan, np.float32)] + [(rng.standard_normal((64, spec.latent_dim)) * 0.7).astype(np.float32) for _ in range(2)]
    outs = []
    for scan in (False, True):
        if scan:
            os.environ["POCKET_TTS_Q8_STAGED"] = "1"
        try:
            eng = full["lib"].Engine(p_dyn, precision="int8", temperature=0.0, lsd_steps=10, max_frames=8)
            st = eng.state(64)
            for slot in range(64):
                eng.lm_prefill(st, slot, embs[slot])
            outs.append([eng.lm_step(st, sq) for sq in seqs])
            st.close(); eng.close()
        finally:
            os.environ.pop("POCKET_TTS_Q8_STAGED", None)
    for (c0, e0), (c1, e1) in zip(*outs):
        np.testing.assert_array_equal(c0, c1)
        np.testing.assert_array_equal(e0, e1)
    assert np.isfinite(outs[0][-1][0]).all() and np.abs(outs[0][-1][0]).max() > 0.1


def test_flow_head_persistent_form_matches_launch_chain(full):
    """bf16 flow head: the one-launch persistent form (every matrix in registers, workgroups handing row groups to each
    other through L2) against the chain of per-block launches on the same inputs, and both against the oracle on the
    bf16-rounded weights.  Row counts cover one row, ragged last row groups and more row groups than replicas."""
    eng = full["lib"].Engine(full["path"], precision="bf16", temperature=0.0, lsd_steps=10, max_frames=F)
    rng = np.random.default_rng(17)
    worst_pair = worst_oracle = 0.0
    for rows in (1, 3, 5, 64, 67):
        cond = rng.standard_normal((rows, full["spec"].d_model)).astype(np.float32)
        x0 = (rng.standard_normal((rows, full["spec"].latent_dim)) * 0.8).astype(np.float32)
        ring = eng.flow_euler(cond, x0)
        again = eng.flow_euler(cond, x0)
        np.testing.assert_array_equal(ring, again)              # fixed summation order: reproducible bit for bit
        os.environ["FLOW_HEAD_NO_RING"] = "1"
        try:
            chain = eng.flow_euler(cond, x0)
        finally:
            os.environ.pop("FLOW_HEAD_NO_RING", None)
        worst_pair = max(worst_pair, float(np.abs(ring - chain).max()))
        for i in (0, rows - 1):
            want = full["oq"].flow_euler(cond[i], x0[i], 10)
            worst_oracle = max(worst_oracle, float(np.abs(ring[i] - want).max()))
    _note(f"bf16 full size flow head: persistent form vs launch chain max-abs {worst_pair:.3e}, vs oracle(bf16 weights) {worst_oracle:.3e}")
    assert worst_pair <= 1e-4 and worst_oracle <= 2e-2
    eng.close()


def test_adaln_product_rows_stationary_form_is_bit_equal(full):
    """The adaLN product of a frame (10 Euler steps x rows against 10240 columns, K = 512) runs on a kernel of its own
    when the row count fills whole 64-row slabs; it accumulates in the order of the kernel it replaces, so the flow head's
    output must not change by a bit when POCKET_TTS_WIDE_ROWS=0 selects the older kernel.  64 and 128 utterances (640 /
    1280 rows) take the form with 32 rows per wave, 96 (960 rows = 15 slabs of 64) the form with 16, 70 (700 rows) the
    older kernel either way."""
    eng = full["lib"].Engine(full["path"], precision="bf16", temperature=0.0, lsd_steps=10, max_frames=F)
    rng = np.random.default_rng(23)
    for rows in (64, 128, 96, 70):
        cond = rng.standard_normal((rows, full["spec"].d_model)).astype(np.float32)
        x0 = (rng.standard_normal((rows, full["spec"].latent_dim)) * 0.8).astype(np.float32)
        wide = eng.flow_euler(cond, x0)
        os.environ["POCKET_TTS_WIDE_ROWS"] = "0"
        try:
            plain = eng.flow_euler(cond, x0)
        finally:
            os.environ.pop("POCKET_TTS_WIDE_ROWS", None)
        np.testing.assert_array_equal(wide, plain)
        os.environ["POCKET_TTS_ADA_F32_ROWS"] = "1"      # the same kernel fed f32 rows instead of ready bf16 fragments
        try:
            f32_rows = eng.flow_euler(cond, x0)
        finally:
            os.environ.pop("POCKET_TTS_ADA_F32_ROWS", None)
        np.testing.assert_array_equal(wide, f32_rows)
        want = full["oq"].flow_euler(cond[rows - 1], x0[rows - 1], 10)
        assert float(np.abs(wide[rows - 1] - want).max()) <= 2e-2
    eng.close()


def test_full_fp32_single_utterance_parity(full):
    """One utterance alone takes the single-row forms of the kernels (float flow head without matrix instructions,
    8 slices; prefill on the skinny kernels): flow head and whole path against the oracle, fp32 bound."""
    eng = full["lib"].Engine(full["path"], precision="fp32", temperature=0.0, lsd_steps=10, max_frames=F)
    rng = np.random.default_rng(21)
    cond = rng.standard_normal((1, full["spec"].d_model)).astype(np.float32)
    x0 = (rng.standard_normal((1, full["spec"].latent_dim)) * 0.8).astype(np.float32)
    got = eng.flow_euler(cond, x0)[0]
    want = full["o"].flow_euler(cond[0], x0[0], 10)
    flow_err = float(np.abs(got - want).max())
    res = eng.generate_ids(full["ids"][:1], full["voices"][:1])[0]
    lat, eos, frames = oracle_replay(full["o"], full["ids"][0], full["voices"][0], res["latents"], 10, F)
    assert res["frames"] == frames == F
    worst = max(float(np.abs(res["latents"] - lat).max()), float(np.abs(res["eos"] - eos).max()))
    snr = snr_db(oracle_decode(full["o"], res["latents"]), res["audio"])
    _note(f"fp32 full size, one utterance: flow head max-abs {flow_err:.3e}, per-frame latent/eos max-abs {worst:.3e}, PCM SNR {snr:.1f} dB")
    assert flow_err <= 1e-3 and worst <= 1e-3 and snr >= 40.0
    eng.close()


def test_full_voice_encoder_parity(full):
    """Mimi encoder at full size on a 0.5 s and a 1 s clip (the second spans more than one output row group and uses the
    K-slice down-sampling product): conditioning rows against the oracle, fp32 bound."""
    eng = full["lib"].Engine(full["path"], precision="fp32", temperature=0.0, lsd_steps=10, max_frames=F)
    rng = np.random.default_rng(9)
    worst = 0.0
    for n in (12000, 24000 + 700):
        t = np.arange(n) / 24000.0
        clip = (0.3 * np.sin(2 * np.pi * 180.0 * t) + 0.05 * rng.standard_normal(n)).astype(np.float32)
        got = eng.mimi_encode(clip)
        want = full["o"].mimi_encode(clip)
        assert got.shape == want.shape
        scale = float(np.abs(want).max())
        worst = max(worst, float(np.abs(got - want).max()) / max(scale, 1.0))
    _note(f"fp32 full size voice encoder: max-abs error relative to the largest output {worst:.3e}")
    assert worst <= 1e-3
    eng.close()


def test_full_streaming_first_chunk(full):
    eng = full["lib"].Engine(full["path"], precision="fp32", temperature=0.0, lsd_steps=10, max_frames=10)
    total, chunks, t_first = eng.streaming_ids(full["ids"][0], full["voices"][0], chunk_frames=5)
    assert total == 10 * 1920 and [len(c) for c, _ in chunks] == [5 * 1920, 5 * 1920]
    assert [f for _, f in chunks] == [0, 0]     # a stream that ends on a chunk boundary never reports is_final
    off = eng.generate_ids(full["ids"][:1], full["voices"][:1])[0]
    np.testing.assert_allclose(np.concatenate([c for c, _ in chunks]), off["audio"], rtol=0, atol=2e-5)
    _note(f"fp32 full size streaming: first 5-frame chunk after {t_first * 1e3:.1f} ms")
    eng.close()


def test_tiny_golden_fixture(packs):
    """Committed vectors (tests/golden/tiny_network.npz, made by tests/golden/make_golden.py): no oracle call here."""
    import zlib
    import ptts_lib
    path, spec, w = packs["tiny"]
    g = np.load(os.path.join(ROOT, "tests", "golden", "tiny_network.npz"))
    assert zlib.crc32(b"".join(w[k].tobytes() for k in sorted(w))) == int(g["pack_crc32"][0])
    eng = ptts_lib.Engine(path, precision="fp32", temperature=0.0, lsd_steps=10, max_frames=8)
    np.testing.assert_array_equal(eng.text_conditioner(g["ids"]), g["text_emb"])
    assert np.abs(eng.flow_euler(g["cond"][None], g["x0"][None])[0] - g["flow10"]).max() < 1e-4
    r = eng.generate_ids([g["ids"]], [g["voice"]])[0]
    assert r["frames"] == 8
    assert np.abs(r["latents"][:3] - g["latents"][:3]).max() < 1e-3      # free-running: early frames only
    assert np.abs(r["eos"][:3] - g["eos"][:3]).max() < 1e-3
    assert snr_db(g["audio"][: spec.frame_size], r["audio"][: spec.frame_size]) >= 40.0
    enc = eng.mimi_encode(g["clip"])
    assert enc.shape == g["enc"].shape and np.abs(enc - g["enc"]).max() < 1e-3
    eng.close()
