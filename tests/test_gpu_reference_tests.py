"""Conformance of the drop-in surface, run on a GPU box:

  * the reference's own test programs, test/test_api.c and test/test_streaming.cpp, compiled from where they lie in the
    reference tree against THIS repository's include/ and libpocket_tts.so (oracle/Makefile, target ref_tests; the
    binaries live under oracle/_ref/ and travel with the tree), run unmodified in the directory layout they expect
    (models/onnx, models/tokenizer.model, models/reference_sample.wav, the program one level below the root);
  * the CLI with the reference's arguments (src/main.cpp:9-67) at temperature 0, twice: the two WAV files are equal
    byte for byte;
  * tools/ref_dump --latents and tools/parity_report.py: per-frame latents and EOS logits of two runs of this build.

The model is the full-size synthetic pack (test_streaming.cpp fixes 1024-wide voice rows), so what is checked is the
behaviour of the API (defaults, paths, sample-count law, chunk schedule, determinism), not the audio of a trained model."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pocket-tts.cpp_amd")
REF_BIN = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, os.path.join(PKG, "tools"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def layout(tmp_path_factory):
    """<root>/models/{onnx/pocket_tts.ptts, tokenizer.model, reference_sample.wav}, <root>/bin/ for the programs."""
    import ptts_lib
    import ptts_pack as pp
    root = tmp_path_factory.mktemp("refroot")
    (root / "models" / "onnx").mkdir(parents=True)
    (root / "bin").mkdir()
    spec = pp.Spec.full()
    pp.make_pack(str(root / "models" / "onnx" / "pocket_tts.ptts"), spec, seed=1234, eos_bias=-20.0)
    shutil.copy(os.path.join(ROOT, "tests", "golden", "sp_unigram.model"), root / "models" / "tokenizer.model")
    t = np.arange(3 * 24000) / 24000.0
    clip = (0.3 * np.sin(2 * np.pi * 180.0 * t) + 0.1 * np.sin(2 * np.pi * 611.0 * t)).astype(np.float32)
    ptts_lib.audio_save_wav(str(root / "models" / "reference_sample.wav"), clip, 24000)
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    return root, spec, env


def _ref_program(name, root):
    src = os.path.join(REF_BIN, name)
    if not os.path.exists(src):
        pytest.skip(f"{src} not built (needs the reference tree: make -C oracle ref_tests)")
    dst = root / "bin" / name
    shutil.copy(src, dst)
    return str(dst)


def test_reference_test_api_program(layout):
    """test/test_api.c: pocket_tts_create(NULL) (every default: models/onnx, models/tokenizer.model, int8, temperature 0.7,
    500 frames), pocket_tts_encode_voice on a WAV path, pocket_tts_generate, the three release calls."""
    root, spec, env = layout
    exe = _ref_program("ref_test_api", root)
    r = subprocess.run([exe], cwd=str(root / "bin"), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Instance created" in r.stdout and "Voice encoded" in r.stdout and "Done!" in r.stdout
    # the synthetic pack never raises EOS: max_frames x samples per frame (src/pocket_tts.cpp:548-657)
    assert f"Generated {500 * spec.frame_size} samples at 24000Hz" in r.stdout, r.stdout


def test_reference_test_streaming_program(layout):
    """test/test_streaming.cpp: the C++ class with precision int8, encodeVoice(path), generateStreaming with 5-frame chunks
    and a callback, AudioUtils::saveWav of what the callback collected."""
    root, spec, env = layout
    exe = _ref_program("ref_test_streaming", root)
    r = subprocess.run([exe], cwd=str(root), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("Chunk ")]
    assert len(lines) == 100 and all(f": {5 * spec.frame_size} samples" in ln for ln in lines), lines[:3]
    assert not any("[FINAL]" in ln for ln in lines)   # 500 frames divide into whole chunks: no tail flush (src/pocket_tts.cpp:897)
    assert f"Total samples: {500 * spec.frame_size}" in r.stdout and f"Collected samples: {500 * spec.frame_size}" in r.stdout
    import ptts_lib
    audio = ptts_lib.audio_load_wav(str(root / "streaming_output.wav"))
    assert len(audio) == 500 * spec.frame_size and np.isfinite(audio).all()


def test_cli_temperature_zero_is_deterministic(layout):
    root, spec, env = layout
    exe = os.path.join(PKG, "pocket_tts_cli")
    assert os.path.exists(exe), "build it with `make -C pocket-tts.cpp_amd`"
    outs = []
    for k in range(2):
        out = str(root / f"cli_{k}.wav")
        r = subprocess.run([exe, "--models-dir", "models/onnx", "--tokenizer", "models/tokenizer.model", "--precision", "fp32",
                            "--temperature", "0", "--max-frames", "40", "Hello, this is a test.", "models/reference_sample.wav", out],
                           cwd=str(root), env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(open(out, "rb").read())
    assert len(outs[0]) == 44 + 40 * spec.frame_size * 4   # 32-bit float samples behind a 44-byte header (src/audio_utils.cpp:262-303)
    assert outs[0] == outs[1]
    # no option is also the reference's behaviour: usage and a non-zero status
    r = subprocess.run([exe], cwd=str(root), env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "Usage:" in r.stdout + r.stderr


def test_ref_dump_latents_and_per_frame_report(layout, tmp_path):
    import parity_report
    root, spec, env = layout
    exe = os.path.join(PKG, "ref_dump")
    assert os.path.exists(exe), "build it with `make -C pocket-tts.cpp_amd ref_dump`"
    base = [exe, "--models", "models/onnx", "--tokenizer", "models/tokenizer.model", "--voice", "models/reference_sample.wav",
            "--text", "Hello, this is a test.", "--max-frames", "12", "--chunk", "5", "--latents"]
    for name, prec in (("a", "fp32"), ("b", "fp32"), ("c", "bf16")):
        r = subprocess.run(base + ["--precision", prec, "--out", str(tmp_path / name)], cwd=str(root), env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
    lat = np.fromfile(str(tmp_path / "a" / "latents.f32"), dtype="<f4")
    eos = np.fromfile(str(tmp_path / "a" / "eos.f32"), dtype="<f4")
    assert lat.size == 12 * spec.latent_dim and eos.size == 12
    ok, lines = parity_report.report(str(tmp_path / "a"), str(tmp_path / "b"), max_latent=1e-3)
    assert ok, lines
    assert sum(ln.startswith("  frame") for ln in lines) == 12 and any("latents: max-abs over frames 0.000e+00" in ln for ln in lines)
    # bf16 against fp32, free-running on random weights: the report shows the drift frame by frame and fails the fp32 bound
    ok_c, lines_c = parity_report.report(str(tmp_path / "a"), str(tmp_path / "c"), min_snr=0.0, max_latent=1e-6)
    assert not ok_c and any("PARITY FAILED" in ln for ln in lines_c)
