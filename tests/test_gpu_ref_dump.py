"""tools/ref_dump (the dump tool that also builds against the reference tree) and tools/parity_report.py, run on
this build: two fp32 dumps of the same input are identical, and the report tool tells a bf16 dump from them."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pocket-tts.cpp_amd")
sys.path.insert(0, os.path.join(PKG, "tools"))

pytestmark = pytest.mark.gpu


def test_ref_dump_and_parity_report(packs, tmp_path):
    import parity_report
    import ptts_lib
    exe = os.path.join(PKG, "ref_dump")
    assert os.path.exists(exe), "build it with `make -C pocket-tts.cpp_amd ref_dump`"
    path, spec, _ = packs["small"]
    models = tmp_path / "models" / "onnx"
    models.mkdir(parents=True)
    shutil.copy(path, models / "pocket_tts.ptts")
    tok = os.path.join(ROOT, "tests", "golden", "sp_unigram.model")
    wav = str(tmp_path / "voice.wav")
    clip = (np.sin(np.arange(30000) / 11.0) * 0.3 + np.sin(np.arange(30000) / 3.1) * 0.1).astype(np.float32)
    ptts_lib.audio_save_wav(wav, clip, 24000)
    base = [exe, "--models", str(models), "--tokenizer", tok, "--voice", wav, "--text", "hello from the dump tool", "--max-frames", "7",
            "--lsd-steps", "4", "--chunk", "3", "--d-model", str(spec.d_model)]
    for name, prec in (("a", "fp32"), ("b", "fp32"), ("c", "bf16")):
        r = subprocess.run(base + ["--precision", prec, "--out", str(tmp_path / name)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
    ok, lines = parity_report.report(str(tmp_path / "a"), str(tmp_path / "b"))
    assert ok, lines
    assert any("SNR inf" in ln for ln in lines)
    ok_c, lines_c = parity_report.report(str(tmp_path / "a"), str(tmp_path / "c"), min_snr=200.0)
    assert not ok_c and any("PARITY FAILED" in ln for ln in lines_c)
