"""ctypes view of oracle/libptts_oracle.so and oracle/_ref/libaudio_ref.so.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
CHUNK_CB = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_void_p)


def build() -> None:
    """Compile the checker (and the reference audio utils when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "libptts_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.ptts_oracle_load.restype = C.c_void_p
        L.ptts_oracle_load.argtypes = [C.c_char_p]
        L.ptts_oracle_free.argtypes = [C.c_void_p]
        L.ptts_oracle_enable_q8.argtypes = [C.c_void_p]
        L.ptts_oracle_enable_q8_ex.argtypes = [C.c_void_p, C.c_int]
        L.ptts_oracle_q8_linear.argtypes = [f32p, C.c_int, C.c_int, f32p, C.c_int, C.c_int, f32p]
        L.ptts_oracle_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.ptts_oracle_text_conditioner.argtypes = [C.c_void_p, i64p, C.c_int, f32p]
        L.ptts_oracle_lm_state_new.restype = C.c_void_p
        L.ptts_oracle_lm_state_new.argtypes = [C.c_void_p, C.c_int]
        L.ptts_oracle_lm_state_free.argtypes = [C.c_void_p]
        L.ptts_oracle_lm_main.restype = C.c_int
        L.ptts_oracle_lm_main.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, f32p, f32p]
        L.ptts_oracle_flow.argtypes = [C.c_void_p, f32p, C.c_float, C.c_float, f32p, f32p]
        L.ptts_oracle_flow_euler.argtypes = [C.c_void_p, f32p, f32p, C.c_int]
        L.ptts_oracle_dec_state_new.restype = C.c_void_p
        L.ptts_oracle_dec_state_new.argtypes = [C.c_void_p]
        L.ptts_oracle_dec_state_free.argtypes = [C.c_void_p]
        L.ptts_oracle_mimi_decode.argtypes = [C.c_void_p, C.c_void_p, f32p, C.c_int, f32p]
        L.ptts_oracle_mimi_encode_frames.restype = C.c_int
        L.ptts_oracle_mimi_encode_frames.argtypes = [C.c_void_p, C.c_int]
        L.ptts_oracle_mimi_encode.restype = C.c_int
        L.ptts_oracle_mimi_encode.argtypes = [C.c_void_p, f32p, C.c_int, f32p]
        L.ptts_oracle_generate.restype = C.c_int
        L.ptts_oracle_generate.argtypes = [C.c_void_p, i64p, C.c_int, f32p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.POINTER(C.c_int)]
        _LIB = L
    return _LIB


def q8_linear(A, W, scheme: int = 0) -> np.ndarray:
    """A . W^T with W int8 per output row and A quantised per the scheme (0: per row and block of 64, symmetric;
    1: ONNX DynamicQuantizeLinear per row, uint8 with a zero point), as ptts_oracle.c's linear_q8 does inside the model."""
    A = np.ascontiguousarray(A, np.float32)
    W = np.ascontiguousarray(W, np.float32)
    out = np.empty((A.shape[0], W.shape[0]), np.float32)
    lib().ptts_oracle_q8_linear(A, A.shape[0], A.shape[1], W, W.shape[0], scheme, out)
    return out


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One loaded weight pack plus the five graph functions and the orchestration."""

    def __init__(self, pack_path: str, int8: bool = False, q8_flow: bool = False):
        self.L = lib()
        self.h = self.L.ptts_oracle_load(pack_path.encode())
        if not self.h:
            raise RuntimeError(f"oracle: cannot load pack {pack_path}")
        if int8:   # LM-layer matrices (q8_flow: and the flow blocks', as an engine with POCKET_TTS_Q8_FLOW=1 holds them) in the
            # engine's int8 scheme (see ptts_oracle.c, linear_q8); the activation scheme is the pack's (Spec.q8_act_scheme)
            self.L.ptts_oracle_enable_q8_ex(self.h, 1 if q8_flow else 0)
        d = (C.c_int * 6)()
        self.L.ptts_oracle_dims(self.h, d)
        self.latent, self.d_model, self.frame_size, self.vocab, self.mimi_dim, self.sample_rate = list(d)

    def close(self):
        if self.h:
            self.L.ptts_oracle_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def text_conditioner(self, ids) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        out = np.empty((len(ids), self.d_model), np.float32)
        self.L.ptts_oracle_text_conditioner(self.h, ids, len(ids), out)
        return out

    def lm_state(self, max_ctx: int = 1024):
        return _State(self.L.ptts_oracle_lm_state_new(self.h, max_ctx), self.L.ptts_oracle_lm_state_free)

    def lm_main(self, state, seq, text):
        """seq [S, latent] or None, text [T, d_model] or None -> (cond [d_model], eos) (None, None if S == 0)."""
        S = 0 if seq is None else len(seq)
        T = 0 if text is None else len(text)
        seq = None if seq is None else np.ascontiguousarray(seq, np.float32)
        text = None if text is None else np.ascontiguousarray(text, np.float32)
        cond = np.zeros(self.d_model, np.float32)
        eos = np.zeros(1, np.float32)
        rc = self.L.ptts_oracle_lm_main(self.h, state.p, _ptr(seq), S, _ptr(text), T, cond, eos)
        if rc != 0:
            raise RuntimeError("oracle lm_main: context overflow")
        return (cond, float(eos[0])) if S else (None, None)

    def flow(self, c, s, t, x) -> np.ndarray:
        v = np.empty(self.latent, np.float32)
        self.L.ptts_oracle_flow(self.h, np.ascontiguousarray(c, np.float32), s, t,
                                np.ascontiguousarray(x, np.float32), v)
        return v

    def flow_euler(self, c, x0, lsd_steps: int) -> np.ndarray:
        x = np.array(x0, np.float32, copy=True)
        self.L.ptts_oracle_flow_euler(self.h, np.ascontiguousarray(c, np.float32), x, lsd_steps)
        return x

    def dec_state(self):
        return _State(self.L.ptts_oracle_dec_state_new(self.h), self.L.ptts_oracle_dec_state_free)

    def mimi_decode(self, state, latent) -> np.ndarray:
        latent = np.ascontiguousarray(latent, np.float32)
        n = len(latent)
        audio = np.empty(n * self.frame_size, np.float32)
        self.L.ptts_oracle_mimi_decode(self.h, state.p, latent, n, audio)
        return audio

    def mimi_encode(self, audio) -> np.ndarray:
        audio = np.ascontiguousarray(audio, np.float32)
        nv = self.L.ptts_oracle_mimi_encode_frames(self.h, len(audio))
        out = np.empty((nv, self.d_model), np.float32)
        got = self.L.ptts_oracle_mimi_encode(self.h, audio, len(audio), out)
        assert got == nv
        return out

    def generate(self, ids, voice, noise=None, lsd_steps=10, max_frames=500, frames_after_eos=3,
                 chunk_frames=0, want_audio=True):
        """Returns dict(latents [F, latent], eos [steps], audio [F*frame], chunks [(n, is_final)])."""
        ids = np.ascontiguousarray(ids, np.int64)
        voice = np.ascontiguousarray(voice, np.float32)
        lat = np.zeros((max_frames, self.latent), np.float32)
        eos = np.zeros(max_frames, np.float32)
        audio = np.zeros(max_frames * self.frame_size, np.float32) if want_audio else None
        steps = C.c_int(0)
        chunks = []

        def _cb(ptr, n, fin, _u):
            chunks.append((np.ctypeslib.as_array(ptr, shape=(n,)).copy(), int(fin)))

        cb = CHUNK_CB(_cb)
        if noise is not None:
            noise = np.ascontiguousarray(noise, np.float32)
            assert noise.shape == (max_frames, self.latent)
        F = self.L.ptts_oracle_generate(self.h, ids, len(ids), voice, len(voice), _ptr(noise), lsd_steps, max_frames,
                                        frames_after_eos, chunk_frames, C.cast(cb, C.c_void_p), None,
                                        _ptr(lat), _ptr(eos), _ptr(audio), C.byref(steps))
        return dict(latents=lat[:F].copy(), eos=eos[:steps.value].copy(),
                    audio=None if audio is None else audio[:F * self.frame_size].copy(), chunks=chunks, frames=F)


class _State:
    def __init__(self, p, free):
        self.p, self._free = p, free

    def __del__(self):
        try:
            if self.p:
                self._free(self.p)
                self.p = None
        except Exception:
            pass


# ----------------------------------------------------------------------------- reference AudioUtils

def ref_audio():
    """oracle/_ref/libaudio_ref.so (the reference's audio_utils.cpp compiled as is), or None."""
    global _REF
    if _REF is None:
        path = os.path.join(HERE, "_ref", "libaudio_ref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.audio_ref_resample.restype = C.c_int
        R.audio_ref_resample.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int]
        R.audio_ref_normalize.restype = C.c_int
        R.audio_ref_normalize.argtypes = [f32p, C.c_int, f32p]
        R.audio_ref_load_wav.restype = C.c_int
        R.audio_ref_load_wav.argtypes = [C.c_char_p, C.c_int, f32p, C.c_int]
        R.audio_ref_save_wav.restype = C.c_int
        R.audio_ref_save_wav.argtypes = [C.c_char_p, f32p, C.c_int, C.c_int]
        _REF = R
    return _REF
