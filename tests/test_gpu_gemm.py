"""The bf16-weight GEMM dispatch at shapes and options the model paths do not all visit: tile tails in M, N and K,
every epilogue option of the lean kernel, its XCD tile order with ragged tile counts, bf16 rows in and out, the two-term
A form, and shapes that fall back to the general kernel.  Reference: numpy on the same bf16-rounded operands, float64 accumulation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bf16(a):
    import ptts_pack as pp
    return pp.bf16_round(np.asarray(a, np.float32))


def _act(v, act):
    from scipy.special import erf
    if act == 1:
        return 0.5 * v * (1.0 + erf(v / np.sqrt(2.0)))
    if act == 2:
        return v / (1.0 + np.exp(-v))
    if act == 3:
        return np.where(v > 0, v, np.expm1(np.minimum(v, 0)))
    return v


@pytest.fixture(scope="module")
def eng(packs):
    import ptts_lib
    path, _, _ = packs["tiny"]
    e = ptts_lib.Engine(path, precision="bf16", temperature=0.0, lsd_steps=2, max_frames=4)
    yield e
    e.close()


CASES = [
    # M, N, K, bias, colscale, R, act, split_a, a_bf16, c_bf16
    (1280, 256, 256, True, False, False, 3, False, False, False),     # lean, ELU
    (1290, 132, 200, True, True, True, 0, False, False, False),       # lean: M, N, K tails, scale and residual
    (515, 36, 40, False, False, False, 1, False, False, False),       # lean: N < tile, K < step, GELU
    (1024, 640, 512, True, False, False, 0, False, True, True),       # lean: bf16 rows in and out
    (700, 64, 128, True, False, False, 3, False, False, True),        # lean: bf16 rows out only
    (96, 64, 128, True, False, False, 0, False, True, True),          # small M with bf16 rows: still the lean kernel
    (1500, 384, 1024, False, True, True, 0, True, False, False),      # lean, two-term A
    (15400, 1160, 1024, True, False, True, 0, False, False, False),   # lean, XCD tile order: 121 m tiles (the last XCDs get one less), n tiles in groups of 8 + 2
    (2048, 300, 4096, False, False, False, 0, True, False, False),    # lean, XCD tile order, two-term A: n tiles in groups of 2 + 1
    (200, 96, 64, True, False, True, 2, False, False, False),         # general kernel (small M)
    (64, 1024, 32, False, False, False, 0, True, False, False),       # general kernel, two-term A
]


@pytest.mark.parametrize("M,N,K,bias,cs,res,act,split,abf,cbf", CASES)
def test_bf16_linear_matches_numpy(eng, M, N, K, bias, cs, res, act, split, abf, cbf):
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = (rng.standard_normal((M, K)) * 0.7).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = (rng.standard_normal(N) * 0.3).astype(np.float32) if bias else None
    s = (rng.standard_normal(N) * 0.5 + 1.0).astype(np.float32) if cs else None
    R = rng.standard_normal((M, N)).astype(np.float32) if res else None
    got = eng.bf16_linear(A, W, b, s, R, act=act, split_a=split, a_bf16=abf, c_bf16=cbf)
    Aq = A if split else _bf16(A)                      # two-term A keeps ~16 mantissa bits
    want = Aq.astype(np.float64) @ _bf16(W).astype(np.float64).T
    if b is not None:
        want = want + b
    if s is not None:
        want = want * s
    if R is not None:
        want = want + R
    want = _act(want, act)
    tol = 2e-4 if split else 2e-5
    if cbf:
        np.testing.assert_allclose(got, want, rtol=2 ** -8, atol=1e-6)      # one bf16 rounding of the stored value
    else:
        np.testing.assert_allclose(got, want, rtol=0, atol=tol * max(1.0, np.abs(want).max()))
