"""The allocation-free flow entry point (ptts_flow_euler_noalloc: device pointers in and out, scratch in a static
device array) against the existing ptts_flow_euler (host buffers, scratch from the engine's pooled batch), through
the C ABI.  Both run the same kernels in the same order, so the bar is bit equality.  Two calls in a row, the second
with fewer rows (it finds the first call's contents in the array), then the limit check."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(eng, cond, x0):
    from hip_mem import DevArray
    dc, dx = DevArray.from_numpy(cond), DevArray.from_numpy(x0)
    out = DevArray((cond.shape[0], eng.latent), fill_byte=0xFF)       # all-ones bytes = NaN: unwritten elements show
    eng.flow_euler_noalloc(dc.ptr, dx.ptr, cond.shape[0], out.ptr)
    got = out.numpy()
    for d in (dc, dx, out):
        d.free()
    return got


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_noalloc_flow_matches_flow_euler(packs, precision):
    import ptts_lib
    path, spec, _ = packs["small"]
    eng = ptts_lib.Engine(path, precision=precision, temperature=0.0, lsd_steps=6, max_frames=4)
    rng = np.random.default_rng(3)
    for rows in (9, 4, 1):                      # each call smaller than the one before
        cond = rng.standard_normal((rows, spec.d_model)).astype(np.float32)
        x0 = (rng.standard_normal((rows, spec.latent_dim)) * 0.8).astype(np.float32)
        want = eng.flow_euler(cond, x0)
        got = _run(eng, cond, x0)
        assert np.isfinite(got).all()
        np.testing.assert_array_equal(got, want)
    eng.close()


def test_noalloc_flow_full_size_and_limit(tmp_path):
    import ptts_lib
    import ptts_pack as pp
    spec = pp.Spec.full()
    path = str(tmp_path / "full.ptts")
    pp.make_pack(path, spec, seed=1234, eos_bias=-20.0)
    eng = ptts_lib.Engine(path, precision="bf16", temperature=0.0, lsd_steps=10, max_frames=4)
    rng = np.random.default_rng(5)
    for rows in (48, 3):
        cond = rng.standard_normal((rows, spec.d_model)).astype(np.float32)
        x0 = (rng.standard_normal((rows, spec.latent_dim)) * 0.8).astype(np.float32)
        np.testing.assert_array_equal(_run(eng, cond, x0), eng.flow_euler(cond, x0))
    cap = eng.L.ptts_flow_euler_noalloc_capacity()
    assert cap == 32 << 20
    from hip_mem import DevArray
    big = 80                                     # 80 rows x 560 KB of scratch > 32 MiB
    dc, dx, out = DevArray((big, spec.d_model), 0), DevArray((big, spec.latent_dim), 0), DevArray((big, spec.latent_dim), 0)
    with pytest.raises(ptts_lib.PttsError, match="static array holds"):
        eng.flow_euler_noalloc(dc.ptr, dx.ptr, big, out.ptr)
    eng.close()
