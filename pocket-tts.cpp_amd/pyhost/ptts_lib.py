"""ctypes bindings of libpocket_tts.so (drop-in C ABI + additive entry points).

This is the product-side host wrapper used by bench.py and the GPU tests.  It never touches oracle/.
Loading fails loudly if the HIP library has not been built; nothing here falls back to a CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("POCKET_TTS_LIB") or os.path.join(PKG, "libpocket_tts.so")   # POCKET_TTS_LIB: another build of the library

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


class PocketTTSConfig(C.Structure):
    _fields_ = [("models_dir", C.c_char_p), ("tokenizer_path", C.c_char_p), ("precision", C.c_char_p),
                ("temperature", C.c_float), ("lsd_steps", C.c_int), ("max_frames", C.c_int)]


class AudioResult(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_float)), ("sample_count", C.c_int), ("sample_rate", C.c_int)]


class StreamingConfig(C.Structure):
    _fields_ = [("chunk_size_frames", C.c_int), ("user_data", C.c_void_p)]


class PocketTTSConfigEx(C.Structure):
    _fields_ = [("models_dir", C.c_char_p), ("tokenizer_path", C.c_char_p), ("precision", C.c_char_p),
                ("pack_path", C.c_char_p), ("temperature", C.c_float), ("lsd_steps", C.c_int),
                ("max_frames", C.c_int), ("frames_after_eos", C.c_int), ("device", C.c_int),
                ("max_text_tokens", C.c_int), ("has_seed", C.c_int), ("seed", C.c_ulonglong)]


CHUNK_CB = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_void_p)

DROP_IN_SYMBOLS = [
    "pocket_tts_create", "pocket_tts_destroy", "pocket_tts_encode_voice", "pocket_tts_encode_voice_from_samples",
    "pocket_tts_free_voice", "pocket_tts_generate", "pocket_tts_free_audio", "pocket_tts_generate_streaming",
    "pocket_tts_cancel_streaming", "pocket_tts_get_last_error", "pocket_tts_version",
]
EXT_SYMBOLS = [
    "pocket_tts_create_ex", "pocket_tts_dims", "pocket_tts_voice_from_embeddings", "pocket_tts_voice_frames",
    "pocket_tts_voice_data", "pocket_tts_encode_voices", "pocket_tts_generate_batch", "pocket_tts_generate_ids", "pocket_tts_generate_ids_queued", "pocket_tts_streaming_ids",
    "pocket_tts_stage_ids", "pocket_tts_run_staged", "pocket_tts_fetch_staged",
    "pocket_tts_tokenize", "ptts_text_conditioner", "ptts_state_new", "ptts_state_free", "ptts_state_reset",
    "ptts_lm_prefill", "ptts_lm_step", "ptts_flow_euler", "ptts_flow_euler_noalloc", "ptts_flow_euler_noalloc_capacity", "ptts_mimi_decode", "ptts_mimi_encode", "ptts_mimi_encode_batch",
    "ptts_phase_timing", "ptts_phase_times", "ptts_weight_bytes", "ptts_scheduler_stats", "ptts_scheduler_stats_ex", "ptts_engine_count", "ptts_q8_linear", "ptts_q8_linear_q", "ptts_q8_linear_dyn", "ptts_bf16_linear",
    "ptts_tokenizer_new", "ptts_tokenizer_free", "ptts_tokenizer_vocab", "ptts_tokenizer_encode",
    "ptts_audio_load_wav", "ptts_audio_save_wav", "ptts_audio_resample", "ptts_audio_normalize",
    "ptts_onnx_open", "ptts_onnx_close", "ptts_onnx_tensor_count", "ptts_onnx_tensor_info", "ptts_onnx_find",
    "ptts_onnx_read_f32", "ptts_onnx_dequantize", "ptts_onnx_node_count", "ptts_onnx_node_info",
]

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C {PKG}` (hipcc, gfx950)")
    L = C.CDLL(LIB_PATH)
    vp, cp, ci = C.c_void_p, C.c_char_p, C.c_int
    L.pocket_tts_create.restype = vp; L.pocket_tts_create.argtypes = [C.POINTER(PocketTTSConfig)]
    L.pocket_tts_create_ex.restype = vp; L.pocket_tts_create_ex.argtypes = [C.POINTER(PocketTTSConfigEx)]
    L.pocket_tts_destroy.restype = None; L.pocket_tts_destroy.argtypes = [vp]
    L.pocket_tts_encode_voice.restype = vp; L.pocket_tts_encode_voice.argtypes = [vp, cp]
    L.pocket_tts_encode_voice_from_samples.restype = vp
    L.pocket_tts_encode_voice_from_samples.argtypes = [vp, f32p, ci, ci]
    L.pocket_tts_free_voice.restype = None; L.pocket_tts_free_voice.argtypes = [vp]
    L.pocket_tts_generate.restype = ci; L.pocket_tts_generate.argtypes = [vp, cp, vp, C.POINTER(AudioResult)]
    L.pocket_tts_free_audio.restype = None; L.pocket_tts_free_audio.argtypes = [C.POINTER(AudioResult)]
    L.pocket_tts_generate_streaming.restype = ci
    L.pocket_tts_generate_streaming.argtypes = [vp, cp, vp, CHUNK_CB, C.POINTER(StreamingConfig)]
    L.pocket_tts_cancel_streaming.restype = None; L.pocket_tts_cancel_streaming.argtypes = [vp]
    L.pocket_tts_get_last_error.restype = cp; L.pocket_tts_get_last_error.argtypes = []
    L.pocket_tts_version.restype = cp; L.pocket_tts_version.argtypes = []
    L.pocket_tts_dims.restype = ci; L.pocket_tts_dims.argtypes = [vp, C.POINTER(ci)]
    L.pocket_tts_voice_from_embeddings.restype = vp; L.pocket_tts_voice_from_embeddings.argtypes = [f32p, ci, ci]
    L.pocket_tts_voice_frames.restype = ci; L.pocket_tts_voice_frames.argtypes = [vp]
    L.pocket_tts_encode_voices.restype = ci; L.pocket_tts_encode_voices.argtypes = [vp, ci, C.POINTER(cp), C.POINTER(vp)]
    L.pocket_tts_voice_data.restype = C.POINTER(C.c_float); L.pocket_tts_voice_data.argtypes = [vp]
    L.pocket_tts_tokenize.restype = ci; L.pocket_tts_tokenize.argtypes = [vp, cp, i64p, ci]
    L.pocket_tts_generate_batch.restype = ci
    L.pocket_tts_generate_batch.argtypes = [vp, ci, C.POINTER(cp), C.POINTER(vp), C.POINTER(AudioResult)]
    L.pocket_tts_generate_ids.restype = ci
    L.pocket_tts_generate_ids.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(ci), C.POINTER(vp), C.POINTER(ci),
                                          C.POINTER(vp), C.POINTER(AudioResult), C.POINTER(vp), C.POINTER(vp),
                                          C.POINTER(ci), C.POINTER(ci)]
    L.pocket_tts_generate_ids_queued.restype = ci
    L.pocket_tts_generate_ids_queued.argtypes = L.pocket_tts_generate_ids.argtypes
    L.pocket_tts_stage_ids.restype = vp
    L.pocket_tts_stage_ids.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(ci), C.POINTER(vp), C.POINTER(ci), C.POINTER(vp)]
    L.pocket_tts_run_staged.restype = ci; L.pocket_tts_run_staged.argtypes = [vp, vp, C.POINTER(ci)]
    L.pocket_tts_fetch_staged.restype = ci
    L.pocket_tts_fetch_staged.argtypes = [vp, vp, C.POINTER(ci), C.POINTER(AudioResult), C.POINTER(vp)]
    L.pocket_tts_streaming_ids.restype = ci
    L.pocket_tts_streaming_ids.argtypes = [vp, i64p, ci, f32p, ci, vp, ci, CHUNK_CB, vp]
    L.ptts_text_conditioner.restype = ci; L.ptts_text_conditioner.argtypes = [vp, i64p, ci, f32p]
    L.ptts_state_new.restype = vp; L.ptts_state_new.argtypes = [vp, ci]
    L.ptts_state_free.restype = None; L.ptts_state_free.argtypes = [vp, vp]
    L.ptts_state_reset.restype = ci; L.ptts_state_reset.argtypes = [vp, vp]
    L.ptts_lm_prefill.restype = ci; L.ptts_lm_prefill.argtypes = [vp, vp, ci, f32p, ci]
    L.ptts_lm_step.restype = ci; L.ptts_lm_step.argtypes = [vp, vp, f32p, f32p, f32p]
    L.ptts_flow_euler.restype = ci; L.ptts_flow_euler.argtypes = [vp, f32p, f32p, ci, f32p]
    L.ptts_flow_euler_noalloc.restype = ci; L.ptts_flow_euler_noalloc.argtypes = [vp, vp, vp, ci, vp]
    L.ptts_flow_euler_noalloc_capacity.restype = C.c_ulonglong; L.ptts_flow_euler_noalloc_capacity.argtypes = []
    L.ptts_mimi_decode.restype = ci; L.ptts_mimi_decode.argtypes = [vp, vp, f32p, ci, f32p]
    L.ptts_mimi_encode.restype = ci; L.ptts_mimi_encode.argtypes = [vp, f32p, ci, f32p, ci]
    L.ptts_mimi_encode_batch.restype = ci; L.ptts_mimi_encode_batch.argtypes = [vp, C.POINTER(vp), ci, ci, C.POINTER(vp), ci]
    L.ptts_phase_timing.restype = ci; L.ptts_phase_timing.argtypes = [vp, ci]
    L.ptts_phase_times.restype = ci; L.ptts_phase_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.ptts_scheduler_stats.restype = ci; L.ptts_scheduler_stats.argtypes = [vp, C.POINTER(C.c_long)]
    L.ptts_scheduler_stats_ex.restype = ci; L.ptts_scheduler_stats_ex.argtypes = [vp, C.POINTER(C.c_long)]
    L.ptts_q8_linear.restype = ci; L.ptts_q8_linear.argtypes = [vp, f32p, ci, ci, f32p, ci, f32p]
    L.ptts_engine_count.restype = ci; L.ptts_engine_count.argtypes = [vp, C.POINTER(C.c_long), ci]
    L.ptts_q8_linear_dyn.restype = ci; L.ptts_q8_linear_dyn.argtypes = [vp, f32p, ci, ci, f32p, ci, f32p]
    L.ptts_q8_linear_q.restype = ci; L.ptts_q8_linear_q.argtypes = [vp, f32p, ci, ci, vp, f32p, ci, f32p]
    L.ptts_bf16_linear.restype = ci
    L.ptts_bf16_linear.argtypes = [vp, f32p, ci, ci, f32p, ci, vp, vp, vp, ci, ci, f32p]
    L.ptts_weight_bytes.restype = ci; L.ptts_weight_bytes.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.ptts_tokenizer_new.restype = vp; L.ptts_tokenizer_new.argtypes = [cp]
    L.ptts_tokenizer_free.restype = None; L.ptts_tokenizer_free.argtypes = [vp]
    L.ptts_tokenizer_vocab.restype = ci; L.ptts_tokenizer_vocab.argtypes = [vp]
    L.ptts_tokenizer_encode.restype = ci; L.ptts_tokenizer_encode.argtypes = [vp, cp, ci, i64p, ci]
    L.ptts_audio_load_wav.restype = ci; L.ptts_audio_load_wav.argtypes = [cp, ci, f32p, ci]
    L.ptts_audio_save_wav.restype = ci; L.ptts_audio_save_wav.argtypes = [cp, f32p, ci, ci]
    L.ptts_audio_resample.restype = ci; L.ptts_audio_resample.argtypes = [f32p, ci, ci, ci, f32p, ci]
    L.ptts_audio_normalize.restype = ci; L.ptts_audio_normalize.argtypes = [f32p, ci, f32p]
    _lib = L
    return L


class Tokenizer:
    """Host-side text pre-step (no GPU needed)."""

    def __init__(self, model_path: str):
        self.L = load()
        self.p = self.L.ptts_tokenizer_new(model_path.encode())
        if not self.p:
            raise PttsError(f"tokenizer: {last_error()}")

    def vocab(self) -> int:
        return self.L.ptts_tokenizer_vocab(self.p)

    def encode(self, text: str, raw: bool = False) -> List[int]:
        buf = np.zeros(4096, np.int64)
        n = self.L.ptts_tokenizer_encode(self.p, text.encode("utf-8"), 1 if raw else 0, buf, len(buf))
        if n < 0:
            raise PttsError(last_error())
        return buf[:n].tolist()

    def close(self):
        if self.p:
            self.L.ptts_tokenizer_free(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def audio_resample(x, in_rate: int, out_rate: int) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    cap = int(np.ceil(len(x) * out_rate / in_rate)) + 8
    out = np.empty(cap, np.float32)
    n = load().ptts_audio_resample(x, len(x), in_rate, out_rate, out, cap)
    if n < 0:
        raise PttsError(last_error())
    return out[:n].copy()


def audio_normalize(x) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(max(len(x), 1), np.float32)
    n = load().ptts_audio_normalize(x, len(x), out)
    if n < 0:
        raise PttsError(last_error())
    return out[:n].copy()


def audio_load_wav(path: str, target_rate: int = 24000, cap: int = 1 << 24) -> np.ndarray:
    out = np.empty(cap, np.float32)
    n = load().ptts_audio_load_wav(path.encode(), target_rate, out, cap)
    if n < 0:
        raise PttsError(last_error())
    return out[:n].copy()


def audio_save_wav(path: str, x, rate: int = 24000) -> None:
    x = np.ascontiguousarray(x, np.float32)
    if load().ptts_audio_save_wav(path.encode(), x, len(x), rate) < 0:
        raise PttsError(last_error())


def last_error() -> str:
    return load().pocket_tts_get_last_error().decode(errors="replace")


class PttsError(RuntimeError):
    pass


def _check(rc: int, what: str) -> int:
    if rc < 0:
        raise PttsError(f"{what}: {last_error()}")
    return rc


class State:
    """Recurrent state of `batch` lock-step utterances."""

    def __init__(self, eng: "Engine", batch: int):
        self.eng, self.batch = eng, batch
        self.p = eng.L.ptts_state_new(eng.h, batch)
        if not self.p:
            raise PttsError(f"state_new: {last_error()}")

    def reset(self):
        _check(self.eng.L.ptts_state_reset(self.eng.h, self.p), "state_reset")

    def close(self):
        if self.p and self.eng.h:
            self.eng.L.ptts_state_free(self.eng.h, self.p)
        self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One engine handle on one GPU (pocket_tts_create_ex)."""

    def __init__(self, pack_path: str, precision: str = "fp32", temperature: float = 0.0, lsd_steps: int = 10,
                 max_frames: int = 500, frames_after_eos: int = 3, device: int = 0, tokenizer: Optional[str] = None,
                 max_text_tokens: int = 512, seed: Optional[int] = None):
        self.L = load()
        cfg = PocketTTSConfigEx()
        cfg.models_dir = None
        cfg.tokenizer_path = (tokenizer or "none").encode()
        cfg.precision = precision.encode()
        cfg.pack_path = pack_path.encode()
        cfg.temperature = temperature
        cfg.lsd_steps = lsd_steps
        cfg.max_frames = max_frames
        cfg.frames_after_eos = frames_after_eos
        cfg.device = device
        cfg.max_text_tokens = max_text_tokens
        cfg.has_seed = 0 if seed is None else 1
        cfg.seed = 0 if seed is None else seed
        self.h = self.L.pocket_tts_create_ex(C.byref(cfg))
        if not self.h:
            raise PttsError(f"create: {last_error()}")
        d = (C.c_int * 8)()
        _check(self.L.pocket_tts_dims(self.h, d), "dims")
        (self.latent, self.d_model, self.frame_size, self.vocab, self.mimi_dim, self.sample_rate,
         self.lsd_steps, self.max_frames) = list(d)

    def close(self):
        if self.h:
            self.L.pocket_tts_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- graph-level calls ----
    def text_conditioner(self, ids) -> np.ndarray:
        ids = np.ascontiguousarray(ids, np.int64)
        out = np.empty((len(ids), self.d_model), np.float32)
        _check(self.L.ptts_text_conditioner(self.h, ids, len(ids), out), "text_conditioner")
        return out

    def state(self, batch: int) -> State:
        return State(self, batch)

    def lm_prefill(self, st: State, slot: int, emb) -> None:
        emb = np.ascontiguousarray(emb, np.float32)
        _check(self.L.ptts_lm_prefill(self.h, st.p, slot, emb, len(emb)), "lm_prefill")

    def lm_step(self, st: State, seq):
        seq = np.ascontiguousarray(seq, np.float32).reshape(st.batch, self.latent)
        cond = np.empty((st.batch, self.d_model), np.float32)
        eos = np.empty(st.batch, np.float32)
        _check(self.L.ptts_lm_step(self.h, st.p, seq, cond, eos), "lm_step")
        return cond, eos

    def flow_euler(self, cond, x0) -> np.ndarray:
        cond = np.ascontiguousarray(cond, np.float32)
        x0 = np.ascontiguousarray(x0, np.float32)
        rows = cond.shape[0]
        out = np.empty((rows, self.latent), np.float32)
        _check(self.L.ptts_flow_euler(self.h, cond, x0, rows, out), "flow_euler")
        return out

    def flow_euler_noalloc(self, cond_ptr: int, x0_ptr: int, rows: int, out_ptr: int) -> None:
        """Device pointers in and out; the library allocates nothing (static scratch array)."""
        _check(self.L.ptts_flow_euler_noalloc(self.h, cond_ptr, x0_ptr, rows, out_ptr), "flow_euler_noalloc")

    def mimi_decode(self, st: State, latents) -> np.ndarray:
        latents = np.ascontiguousarray(latents, np.float32)
        assert latents.shape[0] == st.batch and latents.shape[2] == self.latent
        n = latents.shape[1]
        audio = np.empty((st.batch, n * self.frame_size), np.float32)
        _check(self.L.ptts_mimi_decode(self.h, st.p, latents, n, audio), "mimi_decode")
        return audio

    def mimi_encode(self, audio) -> np.ndarray:
        audio = np.ascontiguousarray(audio, np.float32)
        cap = len(audio) // max(self.frame_size, 1) + 4
        out = np.empty((cap, self.d_model), np.float32)
        n = _check(self.L.ptts_mimi_encode(self.h, audio, len(audio), out, cap), "mimi_encode")
        return out[:n].copy()

    def mimi_encode_batch(self, clips) -> List[np.ndarray]:
        """Equal-length clips through the encoder together (stacked along the row dimension)."""
        keep = [np.ascontiguousarray(c, np.float32) for c in clips]
        n, samples = len(keep), len(keep[0])
        assert all(len(c) == samples for c in keep)
        cap = samples // max(self.frame_size, 1) + 4
        outs = [np.empty((cap, self.d_model), np.float32) for _ in range(n)]
        cp = (C.c_void_p * n)(*[c.ctypes.data for c in keep])
        op = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        rows = _check(self.L.ptts_mimi_encode_batch(self.h, cp, n, samples, op, cap), "mimi_encode_batch")
        return [o[:rows].copy() for o in outs]

    def encode_voice_paths(self, paths: Sequence[str]) -> List[np.ndarray]:
        """pocket_tts_encode_voices: one embedding [frames][d_model] per WAV file, files prepared on host threads while the
        GPU encodes."""
        n = len(paths)
        pp_ = (C.c_char_p * n)(*[p.encode() for p in paths])
        vo = (C.c_void_p * n)()
        _check(self.L.pocket_tts_encode_voices(self.h, n, pp_, vo), "encode_voices")
        out = []
        for i in range(n):
            nf = self.L.pocket_tts_voice_frames(vo[i])
            out.append(np.ctypeslib.as_array(self.L.pocket_tts_voice_data(vo[i]), shape=(nf * self.d_model,)).reshape(nf, self.d_model).copy())
            self.L.pocket_tts_free_voice(vo[i])
        return out

    # ---- whole path ----
    def generate_ids(self, ids_list: Sequence[np.ndarray], voices: Sequence[np.ndarray],
                     noise: Optional[Sequence[Optional[np.ndarray]]] = None, want_latents: bool = True,
                     want_audio: bool = True, queued: bool = False) -> List[dict]:
        """queued: through the slot table (continuous batching, any n) instead of one lock-step batch."""
        n = len(ids_list)
        ids_keep = [np.ascontiguousarray(a, np.int64) for a in ids_list]
        v_keep = [np.ascontiguousarray(v, np.float32) for v in voices]
        n_keep = [None if (noise is None or noise[i] is None) else np.ascontiguousarray(noise[i], np.float32) for i in range(n)]
        for a in n_keep:
            if a is not None:
                assert a.shape == (self.max_frames, self.latent)
        vp = C.c_void_p
        ids_p = (vp * n)(*[a.ctypes.data for a in ids_keep])
        nid = (C.c_int * n)(*[len(a) for a in ids_keep])
        v_p = (vp * n)(*[a.ctypes.data for a in v_keep])
        vf = (C.c_int * n)(*[a.shape[0] for a in v_keep])
        no_p = (vp * n)(*[None if a is None else a.ctypes.data for a in n_keep])
        res = (AudioResult * n)() if want_audio else None
        lat = [np.zeros((self.max_frames, self.latent), np.float32) for _ in range(n)] if want_latents else None
        eos = [np.zeros(self.max_frames, np.float32) for _ in range(n)] if want_latents else None
        lat_p = (vp * n)(*[a.ctypes.data for a in lat]) if want_latents else None
        eos_p = (vp * n)(*[a.ctypes.data for a in eos]) if want_latents else None
        frames = (C.c_int * n)()
        neos = (C.c_int * n)()
        fn = self.L.pocket_tts_generate_ids_queued if queued else self.L.pocket_tts_generate_ids
        _check(fn(self.h, n, ids_p, nid, v_p, vf, no_p, res, lat_p, eos_p, frames, neos), "generate_ids")
        out = []
        for i in range(n):
            d = dict(frames=int(frames[i]))
            if want_audio:
                d["audio"] = np.ctypeslib.as_array(res[i].data, shape=(max(res[i].sample_count, 1),))[:res[i].sample_count].copy()
                d["sample_rate"] = res[i].sample_rate
                self.L.pocket_tts_free_audio(C.byref(res[i]))
            if want_latents:
                d["latents"] = lat[i][:frames[i]].copy()
                d["eos"] = eos[i][:neos[i]].copy()
            out.append(d)
        return out

    def stage_ids(self, ids_list, voices, noise=None):
        """Host -> HBM staging of a batch; returns an opaque handle plus the arrays kept alive."""
        n = len(ids_list)
        ids_keep = [np.ascontiguousarray(a, np.int64) for a in ids_list]
        v_keep = [np.ascontiguousarray(v, np.float32) for v in voices]
        n_keep = [None if (noise is None or noise[i] is None) else np.ascontiguousarray(noise[i], np.float32) for i in range(n)]
        vp = C.c_void_p
        ids_p = (vp * n)(*[a.ctypes.data for a in ids_keep])
        nid = (C.c_int * n)(*[len(a) for a in ids_keep])
        v_p = (vp * n)(*[a.ctypes.data for a in v_keep])
        vf = (C.c_int * n)(*[a.shape[0] for a in v_keep])
        no_p = (vp * n)(*[None if a is None else a.ctypes.data for a in n_keep])
        st = self.L.pocket_tts_stage_ids(self.h, n, ids_p, nid, v_p, vf, no_p)
        if not st:
            raise PttsError(f"stage_ids: {last_error()}")
        return st, n

    def run_staged(self, staged) -> List[int]:
        st, n = staged
        frames = (C.c_int * n)()
        _check(self.L.pocket_tts_run_staged(self.h, st, frames), "run_staged")
        return list(frames)

    def fetch_staged(self, staged, frames, want_latents=False) -> List[dict]:
        st, n = staged
        fr = (C.c_int * n)(*frames)
        res = (AudioResult * n)()
        lat = [np.zeros((self.max_frames, self.latent), np.float32) for _ in range(n)] if want_latents else None
        lat_p = (C.c_void_p * n)(*[a.ctypes.data for a in lat]) if want_latents else None
        _check(self.L.pocket_tts_fetch_staged(self.h, st, fr, res, lat_p), "fetch_staged")
        out = []
        for i in range(n):
            d = dict(frames=frames[i],
                     audio=np.ctypeslib.as_array(res[i].data, shape=(max(res[i].sample_count, 1),))[:res[i].sample_count].copy())
            self.L.pocket_tts_free_audio(C.byref(res[i]))
            if want_latents:
                d["latents"] = lat[i][:frames[i]].copy()
            out.append(d)
        return out

    def streaming_ids(self, ids, voice, chunk_frames: int = 5, noise=None):
        """Returns (total_samples, [(chunk, is_final), ...], first_chunk_seconds)."""
        import time
        ids = np.ascontiguousarray(ids, np.int64)
        voice = np.ascontiguousarray(voice, np.float32)
        noise_k = None if noise is None else np.ascontiguousarray(noise, np.float32)
        chunks = []
        t_first = [None]
        t0 = time.perf_counter()

        def _cb(ptr, n, fin, _u):
            if t_first[0] is None:
                t_first[0] = time.perf_counter() - t0
            chunks.append((np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].copy(), int(fin)))

        cb = CHUNK_CB(_cb)
        total = _check(self.L.pocket_tts_streaming_ids(self.h, ids, len(ids), voice, voice.shape[0],
                                                       None if noise_k is None else noise_k.ctypes.data, chunk_frames, cb, None),
                       "streaming_ids")
        return total, chunks, t_first[0]

    def phase_timing(self, on: bool):
        _check(self.L.ptts_phase_timing(self.h, 1 if on else 0), "phase_timing")

    def phase_times(self):
        t = (C.c_double * 4)()
        c = (C.c_long * 4)()
        _check(self.L.ptts_phase_times(self.h, t, c), "phase_times")
        return dict(lm_ms=t[0], flow_ms=t[1], dec_ms=t[2], head_ms=t[3], lm_calls=c[0], flow_calls=c[1], dec_calls=c[2], head_calls=c[3])

    def q8_linear(self, A, W, dynamic=False) -> np.ndarray:
        """A . W^T on the int8 kernel; dynamic=True: rows of A as uint8 with a zero point over the whole row."""
        A = np.ascontiguousarray(A, np.float32)
        W = np.ascontiguousarray(W, np.float32)
        out = np.empty((A.shape[0], W.shape[0]), np.float32)
        fn = self.L.ptts_q8_linear_dyn if dynamic else self.L.ptts_q8_linear
        _check(fn(self.h, A, A.shape[0], A.shape[1], W, W.shape[0], out), "q8_linear")
        return out

    def q8_linear_q(self, A, q, scale) -> np.ndarray:
        """A . (q * scale)^T on the int8 kernel with the matrix given as int8 values [N][K] and per-row scales."""
        A = np.ascontiguousarray(A, np.float32)
        q = np.ascontiguousarray(q, np.int8)
        scale = np.ascontiguousarray(scale, np.float32)
        out = np.empty((A.shape[0], q.shape[0]), np.float32)
        _check(self.L.ptts_q8_linear_q(self.h, A, A.shape[0], A.shape[1], q.ctypes.data_as(C.c_void_p), scale, q.shape[0], out), "q8_linear_q")
        return out

    def bf16_linear(self, A, W, bias=None, colscale=None, R=None, act=0, split_a=False, a_bf16=False, c_bf16=False) -> np.ndarray:
        """act(A . W^T + bias) * colscale + R through the engine's bf16-weight GEMM dispatch (W rounded to bf16)."""
        A = np.ascontiguousarray(A, np.float32)
        W = np.ascontiguousarray(W, np.float32)
        keep = [None if v is None else np.ascontiguousarray(v, np.float32) for v in (bias, colscale, R)]
        ptrs = [None if v is None else v.ctypes.data_as(C.c_void_p) for v in keep]
        out = np.empty((A.shape[0], W.shape[0]), np.float32)
        flags = (1 if split_a else 0) | (2 if a_bf16 else 0) | (4 if c_bf16 else 0)
        _check(self.L.ptts_bf16_linear(self.h, A, A.shape[0], A.shape[1], W, W.shape[0], ptrs[0], ptrs[1], ptrs[2], act, flags, out), "bf16_linear")
        return out

    def scheduler_stats(self):
        st = (C.c_long * 8)()
        _check(self.L.ptts_scheduler_stats_ex(self.h, st), "scheduler_stats")
        return dict(batches=st[0], utterances=st[1], largest=st[2], frame_steps=st[3], slot_steps=st[4], frames=st[5],
                    decode_calls=st[6], admissions=st[7])

    def engine_served(self) -> List[int]:
        """Utterances served by each engine behind the handle (one engine per device of POCKET_TTS_DEVICES)."""
        buf = (C.c_long * 64)()
        n = _check(self.L.ptts_engine_count(self.h, buf, 64), "engine_count")
        return [int(buf[i]) for i in range(n)]

    def weight_bytes(self):
        b = (C.c_ulonglong * 4)()
        _check(self.L.ptts_weight_bytes(self.h, b), "weight_bytes")
        return dict(lm=b[0], flow=b[1], dec=b[2], enc=b[3])
