"""Weight-pack format (.ptts) for the MI355X Pocket TTS engine: spec, tensor table, writer, reader.

The reference loads five ONNX graphs by file name (src/pocket_tts.cpp:89-113).  Those graphs and
their weights are external downloads that do not exist offline, so the engine ingests one flat
weight pack instead.  Every model dimension is a run-time field of the pack header; nothing about
layer counts or widths is compiled into the kernels.

File layout (little endian):
    0   char[8]  magic "PTTSPK01"
    8   u32      version (1)
    12  u32      n_tensors
    16  i32[64]  integer spec fields (order = SPEC_INT_FIELDS, unused = 0)
    272 f32[16]  float spec fields   (order = SPEC_FLOAT_FIELDS, unused = 0)
    336 entries  n_tensors x { char name[64]; u32 dtype; u32 ndim; u64 shape[4]; u64 offset; u64 nbytes }
    ...  tensor data, each tensor starting at a 256-byte aligned absolute file offset
dtype: 0 = f32 (the only type the canonical pack uses; bf16/int8 copies are derived at load time).
"""
from __future__ import annotations

import dataclasses
import struct
from typing import Optional, Dict, List, Tuple

import numpy as np

MAGIC = b"PTTSPK01"
VERSION = 1
ENTRY_FMT = "<64sII4QQQ"
ENTRY_SIZE = struct.calcsize(ENTRY_FMT)  # 120
HEADER_SIZE = 16 + 64 * 4 + 16 * 4  # 336

SPEC_INT_FIELDS = [
    "latent_dim", "d_model", "lm_layers", "lm_heads", "lm_ffn", "lm_max_period", "vocab",
    "flow_dim", "flow_blocks", "flow_freq_dim",
    "mimi_dim", "mimi_layers", "mimi_heads", "mimi_ffn", "mimi_context", "mimi_max_period",
    "n_filters", "ratio0", "ratio1", "ratio2", "kernel", "last_kernel", "res_kernel",
    "upsample", "sample_rate", "q8_act_scheme",
]
SPEC_FLOAT_FIELDS = ["lm_ln_eps", "flow_ln_eps", "mimi_ln_eps", "flow_max_period"]


@dataclasses.dataclass
class Spec:
    """Model dimensions.  Defaults are the full-size hypothesis; `tiny()` is the unit-test size."""
    latent_dim: int = 32          # reference fixes 32 (src/pocket_tts.cpp:426,579)
    d_model: int = 1024           # reference fixes 1024 (src/pocket_tts.cpp:560, src/pocket_tts_c.cpp:74)
    lm_layers: int = 6
    lm_heads: int = 16
    lm_ffn: int = 4096
    lm_max_period: int = 10000
    vocab: int = 4000             # table has vocab+1 rows (last = padding id)
    flow_dim: int = 512
    flow_blocks: int = 6
    flow_freq_dim: int = 256
    mimi_dim: int = 512
    mimi_layers: int = 2
    mimi_heads: int = 8
    mimi_ffn: int = 2048
    mimi_context: int = 250
    mimi_max_period: int = 10000
    n_filters: int = 64
    ratio0: int = 6               # decoder up-sampling order; the encoder uses the reverse
    ratio1: int = 5
    ratio2: int = 4
    kernel: int = 7
    last_kernel: int = 3
    res_kernel: int = 3
    upsample: int = 16
    sample_rate: int = 24000
    # activation scheme of the int8 engine: 0 = per row and block of 64, symmetric int8 (this build's own); 1 = the dynamic
    # uint8-with-zero-point scheme of ONNX DynamicQuantizeLinear + MatMulInteger (what a dynamically quantised ONNX export of
    # the model applies; tools/onnx_to_ptts.py sets it when the int8 graphs contain those operators)
    q8_act_scheme: int = 0
    lm_ln_eps: float = 1e-5
    flow_ln_eps: float = 1e-6
    mimi_ln_eps: float = 1e-5
    flow_max_period: float = 10000.0

    @property
    def ratios(self) -> Tuple[int, int, int]:
        return (self.ratio0, self.ratio1, self.ratio2)

    @property
    def frame_size(self) -> int:
        """PCM samples per latent frame; 1920 for the full model (pocket_tts.hpp:81)."""
        return self.upsample * self.ratio0 * self.ratio1 * self.ratio2

    @staticmethod
    def full() -> "Spec":
        return Spec()

    @staticmethod
    def tiny() -> "Spec":
        return Spec(d_model=128, lm_layers=2, lm_heads=2, lm_ffn=256, vocab=64,
                    flow_dim=128, flow_blocks=2, flow_freq_dim=64,
                    mimi_dim=128, mimi_layers=1, mimi_heads=2, mimi_ffn=256, mimi_context=20,
                    n_filters=16, ratio0=3, ratio1=2, ratio2=2, upsample=4)

    @staticmethod
    def small() -> "Spec":
        """Mid size: real frame geometry (1920 samples/frame), reduced widths."""
        return Spec(d_model=256, lm_layers=2, lm_heads=4, lm_ffn=512, vocab=128,
                    flow_dim=128, flow_blocks=2, flow_freq_dim=64,
                    mimi_dim=128, mimi_layers=1, mimi_heads=2, mimi_ffn=256, mimi_context=250,
                    n_filters=16)


def tensor_table(s: Spec) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, init-kind) for every tensor, in file order.

    Linear and convolution weights are stored in GEMM layout [N][K]:
      * Linear            : [out][in]
      * Conv1d  k taps    : [Cout][k*Cin], K index = tap*Cin + ci (tap 0 = oldest sample)
      * ConvTranspose1d with kernel 2r, stride r: [r*Cout][2*Cin]; row = phase*Cout + co,
        K index = j*Cin + ci with j=0 the previous input step (kernel tap phase+r) and j=1 the
        current input step (kernel tap phase)
    init-kind: w = N(0, 1/sqrt(K)); b = small bias; g = LayerNorm gain; z = LayerNorm shift;
               ls = layer scale; std/mean = latent de-normalisation; e = embedding rows; eos = EOS bias.
    """
    L, D, F = s.latent_dim, s.d_model, s.lm_ffn
    C, Q = s.flow_dim, s.flow_freq_dim
    M, Fm, U = s.mimi_dim, s.mimi_ffn, s.upsample
    nf = s.n_filters
    t: List[Tuple[str, Tuple[int, ...], str]] = []
    t.append(("text.embed", (s.vocab + 1, D), "e"))
    # flow_lm_main
    t.append(("lm.bos_emb", (L,), "e"))
    t.append(("lm.in_proj.w", (D, L), "w"))
    for i in range(s.lm_layers):
        p = f"lm.l{i}."
        t += [(p + "ln1.g", (D,), "g"), (p + "ln1.b", (D,), "z"),
              (p + "qkv.w", (3 * D, D), "w"), (p + "out.w", (D, D), "w"),
              (p + "ln2.g", (D,), "g"), (p + "ln2.b", (D,), "z"),
              (p + "ff1.w", (F, D), "w"), (p + "ff2.w", (D, F), "w")]
    t += [("lm.out_norm.g", (D,), "g"), ("lm.out_norm.b", (D,), "z"),
          ("lm.eos.w", (1, D), "w"), ("lm.eos.b", (1,), "eos")]
    # flow_lm_flow
    t += [("flow.in_proj.w", (C, L), "w"), ("flow.in_proj.b", (C,), "b")]
    for j in range(2):
        p = f"flow.time{j}."
        t += [(p + "l1.w", (C, Q), "w"), (p + "l1.b", (C,), "b"),
              (p + "l2.w", (C, C), "w"), (p + "l2.b", (C,), "b")]
    t += [("flow.cond.w", (C, D), "w"), ("flow.cond.b", (C,), "b")]
    for i in range(s.flow_blocks):
        p = f"flow.b{i}."
        t += [(p + "ada.w", (3 * C, C), "w"), (p + "ada.b", (3 * C,), "b"),
              (p + "ln.g", (C,), "g"), (p + "ln.b", (C,), "z"),
              (p + "fc1.w", (C, C), "w"), (p + "fc1.b", (C,), "b"),
              (p + "fc2.w", (C, C), "w"), (p + "fc2.b", (C,), "b")]
    t += [("flow.final.ada.w", (2 * C, C), "w"), ("flow.final.ada.b", (2 * C,), "b"),
          ("flow.final.lin.w", (L, C), "w"), ("flow.final.lin.b", (L,), "b")]
    # mimi shared
    t += [("mimi.emb_mean", (L,), "mean"), ("mimi.emb_std", (L,), "std")]

    def transformer(prefix: str):
        out = []
        for i in range(s.mimi_layers):
            p = f"{prefix}.t{i}."
            out += [(p + "ln1.g", (M,), "g"), (p + "ln1.b", (M,), "z"),
                    (p + "qkv.w", (3 * M, M), "w"), (p + "out.w", (M, M), "w"), (p + "ls1", (M,), "ls"),
                    (p + "ln2.g", (M,), "g"), (p + "ln2.b", (M,), "z"),
                    (p + "ff1.w", (Fm, M), "w"), (p + "ff2.w", (M, Fm), "w"), (p + "ls2", (M,), "ls")]
        return out

    # mimi decoder
    t.append(("dec.in_proj.w", (M, L), "w"))
    t.append(("dec.up.w", (M, 2 * U), "w_up"))  # depthwise transposed conv, [channel][tap]
    t += transformer("dec")
    c = 8 * nf
    t += [("dec.conv0.w", (c, s.kernel * M), "w"), ("dec.conv0.b", (c,), "b")]
    for si, r in enumerate(s.ratios):
        p = f"dec.s{si}."
        co = c // 2
        t += [(p + "up.w", (r * co, 2 * c), "w"), (p + "up.b", (co,), "b"),
              (p + "res.c1.w", (co // 2, s.res_kernel * co), "w"), (p + "res.c1.b", (co // 2,), "b"),
              (p + "res.c2.w", (co, co // 2), "w"), (p + "res.c2.b", (co,), "b")]
        c = co
    t += [("dec.final.w", (1, s.last_kernel * c), "w"), ("dec.final.b", (1,), "b")]
    # mimi encoder
    c = nf
    t += [("enc.conv0.w", (c, s.kernel * 1), "w"), ("enc.conv0.b", (c,), "b")]
    for si, r in enumerate(reversed(s.ratios)):
        p = f"enc.s{si}."
        t += [(p + "res.c1.w", (c // 2, s.res_kernel * c), "w"), (p + "res.c1.b", (c // 2,), "b"),
              (p + "res.c2.w", (c, c // 2), "w"), (p + "res.c2.b", (c,), "b"),
              (p + "down.w", (2 * c, 2 * r * c), "w"), (p + "down.b", (2 * c,), "b")]
        c *= 2
    t += [("enc.final.w", (M, s.last_kernel * c), "w"), ("enc.final.b", (M,), "b")]
    t += transformer("enc")
    t += [("enc.down.w", (M, 2 * U * M), "w"), ("enc.spk_proj.w", (D, M), "w")]
    return t


def synth_weights(s: Spec, seed: int = 1234, eos_bias: float = -20.0,
                  layer_scale: float = 0.05, residual_gain: float = 1.0, gate_gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Seeded synthetic weights: linear/conv N(0, 1/sqrt(K)), norms near identity.

    eos_bias = -20 keeps the EOS logit far below the reference's -4.0 threshold
    (src/pocket_tts.cpp:596) so generation length is set by max_frames alone.

    residual_gain scales the matrices that write into the LM's residual stream (attention output and FFN-2
    projections); gate_gain scales the gate third of every flow block's adaLN projection.  With both at 1 every
    residual branch is as large as the stream it is added to, which no trained network is: trained pre-norm
    transformers initialise those projections at 1/sqrt(2 * layers) of the others and adaLN gates at zero, and
    their sensitivity to a rounding error in one layer is correspondingly smaller.  Values below 1 give a model
    whose error propagation is closer to a trained one's; the draws themselves are the same for every gain.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out: Dict[str, np.ndarray] = {}
    for name, shape, kind in tensor_table(s):
        if kind == "w":
            a = rng.standard_normal(shape, dtype=np.float32) / np.float32(np.sqrt(shape[-1]))
        elif kind == "w_up":
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.7)
        elif kind == "b":
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.02)
        elif kind == "g":
            a = np.float32(1.0) + rng.standard_normal(shape, dtype=np.float32) * np.float32(0.05)
        elif kind == "z":
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.05)
        elif kind == "ls":
            a = np.float32(layer_scale) * (np.float32(1.0) + rng.random(shape, dtype=np.float32))
        elif kind == "e":
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.5)
        elif kind == "mean":
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
        elif kind == "std":
            a = np.float32(1.0) + rng.random(shape, dtype=np.float32) * np.float32(0.5)
        elif kind == "eos":
            a = np.full(shape, eos_bias, dtype=np.float32)
        else:
            raise ValueError(kind)
        if kind == "w" and name.startswith("lm.l") and name.endswith((".out.w", ".ff2.w")):
            a = a * np.float32(residual_gain)
        if name.startswith("flow.b") and name.endswith((".ada.w", ".ada.b")):
            a = a.copy()
            a[2 * s.flow_dim:] *= np.float32(gate_gain)      # rows [2C, 3C) produce the gate
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def write_pack(path: str, s: Spec, weights: Dict[str, np.ndarray], q8: Optional[Dict[str, Tuple[np.ndarray, np.ndarray]]] = None) -> None:
    """q8 (optional): name -> (int8 values [N][K], f32 scale per output row [N]) of matrices that exist in int8 form at the
    source (the `_int8` graphs the reference loads, src/pocket_tts.cpp:90-96).  They are stored beside the f32 tensor as
    `<name>.q8` (element type 1 = int8) and `<name>.q8s`; the int8 engine multiplies with exactly these values instead of
    quantising the f32 tensor again."""
    table = [(name, shape, 0) for name, shape, _ in tensor_table(s)]
    arrays = {}
    for name, shape, _ in table:
        a = weights[name]
        if tuple(a.shape) != tuple(shape) or a.dtype != np.float32:
            raise ValueError(f"{name}: expected f32 {shape}, got {a.dtype} {a.shape}")
        arrays[name] = a
    for name, (q, sc) in sorted((q8 or {}).items()):
        if name not in arrays or arrays[name].ndim != 2:
            raise ValueError(f"{name}: int8 form given for a tensor that is not a matrix of the pack")
        q = np.ascontiguousarray(q)
        sc = np.ascontiguousarray(sc, np.float32)
        if q.dtype != np.int8 or q.shape != arrays[name].shape or sc.shape != (q.shape[0],):
            raise ValueError(f"{name}: int8 form must be int8 {arrays[name].shape} with {q.shape[0]} f32 scales")
        table.append((name + ".q8", q.shape, 1))
        arrays[name + ".q8"] = q
        table.append((name + ".q8s", sc.shape, 0))
        arrays[name + ".q8s"] = sc
    n = len(table)
    ints = [int(getattr(s, f)) for f in SPEC_INT_FIELDS] + [0] * (64 - len(SPEC_INT_FIELDS))
    flts = [float(getattr(s, f)) for f in SPEC_FLOAT_FIELDS] + [0.0] * (16 - len(SPEC_FLOAT_FIELDS))
    off = HEADER_SIZE + n * ENTRY_SIZE
    entries = []
    for name, shape, dt in table:
        a = arrays[name]
        off = (off + 255) & ~255
        shp = list(shape) + [1] * (4 - len(shape))
        entries.append((name, dt, len(shape), shp, off, a.nbytes))
        off += a.nbytes
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<II", VERSION, n))
        f.write(struct.pack("<64i", *ints))
        f.write(struct.pack("<16f", *flts))
        for name, dt, ndim, shp, o, nb in entries:
            f.write(struct.pack(ENTRY_FMT, name.encode(), dt, ndim, *shp, o, nb))
        for name, dt, ndim, shp, o, nb in entries:
            f.seek(o)
            f.write(arrays[name].tobytes())


def read_pack(path: str) -> Tuple[Spec, Dict[str, np.ndarray]]:
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:8] != MAGIC:
        raise ValueError("not a PTTS pack")
    ver, n = struct.unpack_from("<II", raw, 8)
    if ver != VERSION:
        raise ValueError(f"unsupported pack version {ver}")
    ints = struct.unpack_from("<64i", raw, 16)
    flts = struct.unpack_from("<16f", raw, 16 + 256)
    kw = {f: ints[i] for i, f in enumerate(SPEC_INT_FIELDS)}
    kw.update({f: flts[i] for i, f in enumerate(SPEC_FLOAT_FIELDS)})
    s = Spec(**kw)
    w: Dict[str, np.ndarray] = {}
    for i in range(n):
        e = struct.unpack_from(ENTRY_FMT, raw, HEADER_SIZE + i * ENTRY_SIZE)
        name = e[0].rstrip(b"\0").decode()
        ndim = e[2]
        shape = tuple(e[3:3 + ndim])
        o, nb = e[7], e[8]
        if e[1] == 1 or name.endswith(".q8s"):
            continue                      # int8 forms: read_pack_q8
        w[name] = np.frombuffer(raw, dtype=np.float32, count=nb // 4, offset=o).reshape(shape).copy()
    return s, w


def read_pack_q8(path: str) -> Dict[str, Tuple[np.ndarray, np.ndarray]]:
    """The int8 forms a pack carries: name -> (int8 [N][K], f32 scale [N])."""
    with open(path, "rb") as f:
        raw = f.read()
    n = struct.unpack_from("<II", raw, 8)[1]
    q, sc = {}, {}
    for i in range(n):
        e = struct.unpack_from(ENTRY_FMT, raw, HEADER_SIZE + i * ENTRY_SIZE)
        name = e[0].rstrip(b"\0").decode()
        shape = tuple(e[3:3 + e[2]])
        o, nb = e[7], e[8]
        if e[1] == 1 and name.endswith(".q8"):
            q[name[:-3]] = np.frombuffer(raw, dtype=np.int8, count=nb, offset=o).reshape(shape).copy()
        elif name.endswith(".q8s"):
            sc[name[:-4]] = np.frombuffer(raw, dtype=np.float32, count=nb // 4, offset=o).copy()
    return {k: (q[k], sc[k]) for k in q if k in sc}


def make_pack(path: str, spec: Spec, seed: int = 1234, eos_bias: float = -20.0,
              layer_scale: float = 0.05, residual_gain: float = 1.0, gate_gain: float = 1.0) -> Dict[str, np.ndarray]:
    w = synth_weights(spec, seed=seed, eos_bias=eos_bias, layer_scale=layer_scale, residual_gain=residual_gain,
                      gate_gain=gate_gain)
    write_pack(path, spec, w)
    return w


F32_ONLY_WEIGHTS = ("dec.up.w", "dec.final.w", "lm.eos.w")
F32_ONLY_PREFIXES = ("enc.", "flow.time")   # voice encoder (fp32 at every precision) and the host-evaluated time embeddings


def bf16_round(a: np.ndarray) -> np.ndarray:
    """float32 -> nearest-even bfloat16 -> float32 (the rounding the engine applies to its bf16 weight copy)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


# Matrix families the bf16 engine keeps as two bf16 terms (~16 mantissa bits, float-grade) by default; the int8 engine
# keeps the same five, or only the first three when its flow blocks are int8 as well (POCKET_TTS_Q8_FLOW=1: that flow head
# has no room for second planes).
# Mirrors two_term_families() in csrc/engine.cpp.
TWO_TERM_DEFAULT = ("lm.in_proj", "flow.cond", "flow.ada", "flow.in_proj", "flow.final.lin")


def in_family(name: str, family: str) -> bool:
    if family in ("lm.in_proj", "flow.cond", "flow.in_proj", "flow.final.lin"):
        return name == family + ".w"
    if family == "flow.ada":
        return name.startswith("flow.") and name.endswith(".ada.w")
    if family in ("lm.qkv", "lm.out", "lm.ff1", "lm.ff2"):
        return name.startswith("lm.l") and name.endswith("." + family[3:] + ".w")
    raise ValueError(family)


def bf16_model(weights: Dict[str, np.ndarray], two_term=TWO_TERM_DEFAULT) -> Dict[str, np.ndarray]:
    """The model the bf16 engine actually holds: matrix weights rounded to bf16, everything else
    (biases, norms, embedding table, depthwise/1-channel convs, EOS head, time-embedding MLP, voice encoder)
    left in float32, and so are the families held as two bf16 terms (their 2^-17 residue is below float32
    accumulation noise).  two_term=() gives every matrix rounded."""
    def rounded(k):
        if any(in_family(k, f) for f in two_term):
            return False
        return k.endswith(".w") and k not in F32_ONLY_WEIGHTS and not k.startswith(F32_ONLY_PREFIXES)
    return {k: (bf16_round(v) if rounded(k) else v) for k, v in weights.items()}


def is_int8_weight(name: str, flow_dim: int = 512, scheme: int = 0, flow_blocks: bool = False) -> bool:
    """Matrices the int8 engine holds as int8 (per-output-channel scale): the LM layers and, with flow_blocks
    (POCKET_TTS_Q8_FLOW=1 on the engine), flow_dim a multiple of 256 and the per-block activation scheme (scheme 0),
    the flow residual blocks."""
    if name.startswith("lm.l") and name.endswith((".qkv.w", ".out.w", ".ff1.w", ".ff2.w")):
        return True
    return flow_blocks and scheme == 0 and flow_dim % 256 == 0 and name.startswith("flow.b") and name.endswith((".fc1.w", ".fc2.w"))


def int8_model(weights: Dict[str, np.ndarray], scheme: int = 0, flow_blocks: bool = False) -> Dict[str, np.ndarray]:
    """Weight set for checking the int8 engine: like bf16_model, except that the matrices the engine quantises
    to int8 stay float32 here (both sides derive the int8 values and scales from the same float32 numbers).
    scheme = Spec.q8_act_scheme of the pack the weights go into; flow_blocks = the engine runs with POCKET_TTS_Q8_FLOW=1."""
    flow_dim = weights["flow.in_proj.w"].shape[0]
    blocks_q8 = flow_blocks and scheme == 0 and flow_dim % 256 == 0
    q = bf16_model(weights, two_term=TWO_TERM_DEFAULT[:3] if blocks_q8 else TWO_TERM_DEFAULT)
    for k, v in weights.items():
        if is_int8_weight(k, flow_dim, scheme, flow_blocks):
            q[k] = v
    return q


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="write a seeded synthetic Pocket TTS weight pack")
    ap.add_argument("out")
    ap.add_argument("--size", choices=["tiny", "small", "full"], default="full")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--eos-bias", type=float, default=-20.0)
    a = ap.parse_args()
    sp = {"tiny": Spec.tiny, "small": Spec.small, "full": Spec.full}[a.size]()
    make_pack(a.out, sp, seed=a.seed, eos_bias=a.eos_bias)
    print(f"wrote {a.out}: frame_size={sp.frame_size}")
