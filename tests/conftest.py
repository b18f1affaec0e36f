import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pocket-tts.cpp_amd", "pyhost"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def packs(tmp_path_factory):
    """Seeded synthetic weight packs, written once per session: name -> (path, spec, weights)."""
    import ptts_pack as pp
    d = tmp_path_factory.mktemp("packs")
    out = {}

    def make(name, spec, seed, eos_bias, layer_scale=0.05):
        path = str(d / f"{name}.ptts")
        w = pp.make_pack(path, spec, seed=seed, eos_bias=eos_bias, layer_scale=layer_scale)
        out[name] = (path, spec, w)

    make("tiny", pp.Spec.tiny(), 7, -20.0, 0.2)
    make("tiny_eos", pp.Spec.tiny(), 7, -4.6, 0.2)   # EOS logit hovers around the -4.0 threshold
    make("tiny_spread", pp.Spec.tiny(), 7, -6.6, 0.2)   # first crossing of the threshold spread over ~3..150+ steps
    make("small", pp.Spec.small(), 11, -20.0, 0.1)
    return out


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle_py
    oracle_py.build()
    return oracle_py


def snr_db(ref: np.ndarray, test: np.ndarray) -> float:
    ref = np.asarray(ref, np.float64)
    err = np.asarray(test, np.float64) - ref
    return 10.0 * np.log10((ref ** 2).sum() / max((err ** 2).sum(), 1e-300))


def oracle_replay(o, ids, voice, gpu_latents, lsd_steps, max_frames, frames_after_eos=3, noise=None):
    """Teacher-forced check of a free-running GPU generation.

    A random-weight network is a chaotic map of its own output: rounding differences of 1e-6 in frame 0
    roughly double every frame, so two correct free-running implementations drift apart.  Per-frame
    parity is therefore checked with both sides seeing the same history: the oracle recomputes frame k
    from the GPU's frames 0..k-1 and applies the reference stop rule (src/pocket_tts.cpp:596-603) to its
    own EOS logits.  Returns (oracle latents [F, L], oracle eos [steps], oracle frame count).
    """
    L = o.latent
    st = o.lm_state(len(voice) + len(ids) + max_frames + 8)
    o.lm_main(st, None, voice)
    o.lm_main(st, None, o.text_conditioner(ids))
    cur = np.full((1, L), np.nan, np.float32)
    lat, eos = [], []
    eos_step, frames = -1, 0
    for step in range(max_frames):
        cond, e = o.lm_main(st, cur, None)
        eos.append(e)
        if e > -4.0 and eos_step < 0:
            eos_step = step
        if eos_step >= 0 and step >= eos_step + frames_after_eos:
            break
        x0 = np.zeros(L, np.float32) if noise is None else noise[step]
        lat.append(o.flow_euler(cond, x0, lsd_steps))
        frames += 1
        if step >= len(gpu_latents):
            break
        cur = gpu_latents[step:step + 1]
    return np.array(lat, np.float32).reshape(-1, L), np.array(eos, np.float32), frames


def oracle_decode(o, latents):
    """Oracle audio for a latent sequence, in the reference's 15-frame chunks (src/pocket_tts.cpp:461)."""
    ds = o.dec_state()
    out = [o.mimi_decode(ds, latents[i:i + 15]) for i in range(0, len(latents), 15)]
    return np.concatenate(out) if out else np.zeros(0, np.float32)
