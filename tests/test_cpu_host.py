"""CPU-only tests (no GPU): the C ABI loads and exports every declared symbol, struct layouts match the
reference's FFI mirrors, and the host-side pre/post steps (tokenizer, audio utils) reproduce the golden
vectors that come from the sentencepiece package and from the reference's own audio_utils.cpp."""
import ctypes as C
import json
import os
import re
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def ptts():
    import ptts_lib
    if not os.path.exists(ptts_lib.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-j8", "-C", ptts_lib.PKG])
    ptts_lib.load()
    return ptts_lib


def _declared_symbols():
    names = set()
    for hdr in ("pocket_tts_c.h", "pocket_tts_amd.h"):
        text = open(os.path.join(ROOT, "include", "pocket_tts", hdr)).read()
        names |= set(re.findall(r"POCKET_TTS_API\s+[\w\s\*]+?\b(pocket_tts_\w+|ptts_\w+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol(ptts):
    lib = C.CDLL(ptts.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 11 + 20
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert set(ptts.DROP_IN_SYMBOLS) <= declared
    assert set(ptts.EXT_SYMBOLS) <= declared


def test_struct_layouts_match_reference_ffi(ptts):
    # sizes measured on the reference's own ctypes / P/Invoke mirrors (test/test_api.py:68-73, test/PocketTTS.cs:11-53)
    assert C.sizeof(ptts.PocketTTSConfig) == 40
    assert C.sizeof(ptts.AudioResult) == 16
    assert C.sizeof(ptts.StreamingConfig) == 16
    assert ptts.PocketTTSConfig.temperature.offset == 24 and ptts.PocketTTSConfig.max_frames.offset == 32
    assert ptts.AudioResult.sample_count.offset == 8 and ptts.AudioResult.sample_rate.offset == 12
    assert ptts.StreamingConfig.user_data.offset == 8


def test_version_and_null_argument_errors(ptts):
    L = ptts.load()
    assert L.pocket_tts_version() == b"1.0.0"
    assert not L.pocket_tts_encode_voice(None, b"x.wav")
    assert L.pocket_tts_get_last_error() == b"Invalid handle or audio path"
    res = ptts.AudioResult()
    assert L.pocket_tts_generate(None, b"hi", None, C.byref(res)) == -1
    assert L.pocket_tts_get_last_error() == b"Invalid parameters"
    assert L.pocket_tts_generate_streaming(None, b"hi", None, ptts.CHUNK_CB(lambda *a: None), None) == -1
    L.pocket_tts_free_audio(C.byref(res))
    L.pocket_tts_destroy(None)
    L.pocket_tts_free_voice(None)
    L.pocket_tts_cancel_streaming(None)


def test_create_fails_loudly_without_weights(ptts, tmp_path):
    L = ptts.load()
    cfg = ptts.PocketTTSConfig(models_dir=str(tmp_path).encode(), tokenizer_path=None, precision=None, temperature=0, lsd_steps=0, max_frames=0)
    assert not L.pocket_tts_create(C.byref(cfg))
    msg = L.pocket_tts_get_last_error()
    assert msg.startswith(b"Failed to create PocketTTS: ")   # no GPU or no weight pack: either way no silent fallback


def test_voice_handle_roundtrip(ptts):
    L = ptts.load()
    emb = np.arange(3 * 8, dtype=np.float32).reshape(3, 8)
    v = L.pocket_tts_voice_from_embeddings(emb, 3, 8)
    assert L.pocket_tts_voice_frames(v) == 3
    back = np.ctypeslib.as_array(L.pocket_tts_voice_data(v), shape=(24,))
    np.testing.assert_array_equal(back, emb.ravel())
    L.pocket_tts_free_voice(v)


# ----------------------------------------------------------------------------- tokenizer

def test_text_cleanup_rules(ptts):
    """trim, full stop after an alphanumeric ending, capital first letter (src/tokenizer.cpp:31-48)."""
    tk = ptts.Tokenizer(os.path.join(GOLD, "sp_unigram.model"))
    same = lambda a, b: tk.encode(a) == tk.encode(b, raw=True)
    assert same("hello world", "Hello world.")
    assert same("  hello world  \n", "Hello world.")
    assert same("Hello world!", "Hello world!")
    assert same("what's up?", "What's up?")
    assert same("route 66", "Route 66.")
    assert same("1 is a number", "1 is a number.")
    with pytest.raises(ptts.PttsError, match="Text cannot be empty"):
        tk.encode("   \t\n")


@pytest.mark.parametrize("model", ["sp_unigram", "sp_bpe_bytefallback"])
def test_tokenizer_matches_sentencepiece_golden(ptts, model):
    gold = json.load(open(os.path.join(GOLD, "tokenizer_ids.json")))
    tk = ptts.Tokenizer(os.path.join(GOLD, model + ".model"))
    assert tk.vocab() == gold["models"][model]["vocab"]
    for text, ids in zip(gold["texts"], gold["models"][model]["ids"]):
        assert tk.encode(text, raw=True) == ids, text


@pytest.mark.parametrize("model", ["sp_unigram", "sp_bpe_bytefallback"])
def test_tokenizer_matches_sentencepiece_live(ptts, model):
    spm = pytest.importorskip("sentencepiece")
    sp = spm.SentencePieceProcessor(model_file=os.path.join(GOLD, model + ".model"))
    tk = ptts.Tokenizer(os.path.join(GOLD, model + ".model"))
    rng = np.random.default_rng(5)
    words = ["the", "License", "copyright", "def", "return", "import", "zürich", "42", "—", "naïve", "x_y", "(a+b)", "…", "Ω", "\t"]
    for _ in range(200):
        text = " ".join(rng.choice(words, size=rng.integers(1, 12)))
        assert tk.encode(text, raw=True) == sp.encode(text), text


def test_tokenizer_missing_model(ptts):
    with pytest.raises(ptts.PttsError, match="Failed to load tokenizer"):
        ptts.Tokenizer("/nonexistent/tokenizer.model")


# ----------------------------------------------------------------------------- audio utils (bit-exact vs reference)

@pytest.fixture(scope="module")
def audio_gold():
    return np.load(os.path.join(GOLD, "audio_ref.npz"))


@pytest.mark.parametrize("a,b", [(44100, 24000), (16000, 24000), (48000, 24000), (22050, 24000), (8000, 24000), (24000, 16000)])
def test_resample_bit_exact(ptts, audio_gold, a, b):
    np.testing.assert_array_equal(ptts.audio_resample(audio_gold["sig"], a, b), audio_gold[f"resample_{a}_{b}"])


def test_resample_identity_and_empty(ptts, audio_gold):
    np.testing.assert_array_equal(ptts.audio_resample(audio_gold["sig"], 24000, 24000), audio_gold["sig"])
    assert len(ptts.audio_resample(np.zeros(0, np.float32), 16000, 24000)) == 0


def test_normalize_bit_exact(ptts, audio_gold):
    sig = audio_gold["sig"]
    np.testing.assert_array_equal(ptts.audio_normalize((sig * 3.0).astype(np.float32)), audio_gold["normalize_loud"])
    np.testing.assert_array_equal(ptts.audio_normalize(sig), audio_gold["normalize_quiet"])   # never amplifies
    assert abs(float(np.abs(audio_gold["normalize_loud"]).max()) - 0.85) < 1e-6


@pytest.mark.parametrize("case", ["pcm16_mono_44k", "pcm16_stereo_24k", "f32_stereo_48k", "f32_mono_loud_24k", "pcm24_mono_22k"])
def test_wav_decode_bit_exact(ptts, audio_gold, tmp_path, case):
    path = tmp_path / (case + ".wav")
    path.write_bytes(audio_gold["wav_" + case].tobytes())
    np.testing.assert_array_equal(ptts.audio_load_wav(str(path), 24000), audio_gold["dec_" + case])


def test_wav_errors_and_roundtrip(ptts, tmp_path):
    with pytest.raises(ptts.PttsError, match="Failed to open audio file"):
        ptts.audio_load_wav(str(tmp_path / "missing.wav"))
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"RIFX" + b"\0" * 40)
    with pytest.raises(ptts.PttsError, match="Not a valid WAV file"):
        ptts.audio_load_wav(str(bad))
    x = (np.sin(np.arange(500) / 7.0) * 0.5).astype(np.float32)
    out = tmp_path / "rt.wav"
    ptts.audio_save_wav(str(out), x, 24000)
    raw = out.read_bytes()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"WAVE" and len(raw) == 44 + 4 * len(x)
    np.testing.assert_array_equal(ptts.audio_load_wav(str(out), 24000), x)
    ref = __import__("oracle_py").ref_audio()
    if ref is not None:   # the reference writer produces the same bytes
        out2 = tmp_path / "rt_ref.wav"
        assert ref.audio_ref_save_wav(str(out2).encode(), x, len(x), 24000) == 0
        assert out2.read_bytes() == raw


# ----------------------------------------------------------------------------- pack format

def test_pack_roundtrip_and_seed_stability(tmp_path):
    import ptts_pack as pp
    spec = pp.Spec.tiny()
    path = str(tmp_path / "t.ptts")
    w = pp.make_pack(path, spec, seed=7, eos_bias=-20.0, layer_scale=0.2)
    spec2, w2 = pp.read_pack(path)
    import dataclasses
    for k, v in dataclasses.asdict(spec).items():   # float fields are stored as f32
        assert getattr(spec2, k) == (v if isinstance(v, int) else pytest.approx(v, rel=1e-6)), k
    assert set(w2) == set(w)
    for k in w:
        np.testing.assert_array_equal(w[k], w2[k])
    assert spec.frame_size == 48 and pp.Spec.full().frame_size == 1920
    gold = np.load(os.path.join(GOLD, "tiny_network.npz"))
    crc = zlib.crc32(b"".join(w[k].tobytes() for k in sorted(w)))
    assert crc == int(gold["pack_crc32"][0]), "numpy's PCG64 stream changed: regenerate tests/golden"


def test_parity_report_tool(tmp_path):
    """tools/parity_report.py on hand-made dumps: identical -> ok with infinite SNR; a noisy copy fails the 40 dB bound
    and a different chunk layout is reported."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "pocket-tts.cpp_amd", "tools"))
    import parity_report
    rng = np.random.default_rng(5)
    pcm = (rng.standard_normal(3840) * 0.2).astype("<f4")
    voice = rng.standard_normal(2048).astype("<f4")

    def dump(name, p, v, chunks):
        d = tmp_path / name
        d.mkdir()
        p.tofile(d / "pcm.f32"); v.tofile(d / "voice.f32"); p.tofile(d / "stream_pcm.f32")
        json.dump(dict(text="t", temperature=0.0, lsd_steps=10, max_frames=500, chunk_frames=5, frames=len(p) // 1920, chunks=chunks),
                  open(d / "meta.json", "w"))
        return str(d)

    a = dump("a", pcm, voice, [[3840, 1]])
    b = dump("b", pcm.copy(), voice.copy(), [[3840, 1]])
    c = dump("c", (pcm + rng.standard_normal(3840).astype("<f4") * 0.02).astype("<f4"), voice, [[1920, 0], [1920, 1]])
    ok, lines = parity_report.report(a, b)
    assert ok and "SNR inf" in lines[1]
    ok, lines = parity_report.report(a, c)
    assert not ok and lines[-1] == "PARITY FAILED"
    snr = float(lines[1].split("SNR ")[1].split(" dB")[0])
    assert 19.0 < snr < 21.0                                                     # 0.2 / 0.02 -> 20 dB
    ok, _ = parity_report.report(a, c, min_snr=15.0)
    assert not ok                                                                # chunk layouts still differ
