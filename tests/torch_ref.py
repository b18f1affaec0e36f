"""Independent float64 PyTorch transcription of the five graphs, whole-sequence (non-streaming) form.

It exists to cross-check oracle/ptts_oracle.c: the C checker is written as streaming code with GEMM
layout weights, this file is written with torch's own conv1d / conv_transpose1d / layer_norm on
channels-first tensors and full-sequence masks.  Agreement of the two pins the checker's layout
conventions and its streaming-state logic.  Not used by the product.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

DT = torch.float64


class TorchRef:
    def __init__(self, spec, weights):
        self.s = spec
        self.w = {k: torch.from_numpy(v).to(DT) for k, v in weights.items()}

    # ------------------------------------------------------------------ helpers
    def _rope(self, x, pos0, max_period):
        # x [T, H, dh]; interleaved pairs
        T, H, dh = x.shape
        i = torch.arange(dh // 2, dtype=DT)
        freq = torch.exp(-math.log(max_period) * 2.0 * i / dh)
        pos = torch.arange(pos0, pos0 + T, dtype=DT)
        ang = pos[:, None] * freq[None, :]
        c, s = torch.cos(ang)[:, None, :], torch.sin(ang)[:, None, :]
        a, b = x[..., 0::2], x[..., 1::2]
        out = torch.empty_like(x)
        out[..., 0::2] = a * c - b * s
        out[..., 1::2] = a * s + b * c
        return out

    def _transformer(self, prefix, x, layers, heads, eps, max_period, context, has_ls):
        # x [T, d]; full sequence from position 0
        T, d = x.shape
        dh = d // heads
        idx = torch.arange(T)
        delta = idx[:, None] - idx[None, :]
        mask = delta >= 0
        if context > 0:
            mask &= delta < context
        for l in range(layers):
            p = f"{prefix}{l}."
            h = F.layer_norm(x, (d,), self.w[p + "ln1.g"], self.w[p + "ln1.b"], eps)
            qkv = h @ self.w[p + "qkv.w"].T
            q, k, v = qkv.split(d, dim=1)
            q = self._rope(q.reshape(T, heads, dh), 0, max_period)
            k = self._rope(k.reshape(T, heads, dh), 0, max_period)
            v = v.reshape(T, heads, dh)
            sc = torch.einsum("thd,shd->hts", q, k) / math.sqrt(dh)
            sc = sc.masked_fill(~mask[None], float("-inf"))
            a = torch.einsum("hts,shd->thd", torch.softmax(sc, dim=-1), v).reshape(T, d)
            o = a @ self.w[p + "out.w"].T
            x = x + (self.w[p + "ls1"] * o if has_ls else o)
            h = F.layer_norm(x, (d,), self.w[p + "ln2.g"], self.w[p + "ln2.b"], eps)
            f = F.gelu(h @ self.w[p + "ff1.w"].T) @ self.w[p + "ff2.w"].T
            x = x + (self.w[p + "ls2"] * f if has_ls else f)
        return x

    @staticmethod
    def _conv_w(w, cin):
        cout, kk = w.shape
        k = kk // cin
        return w.reshape(cout, k, cin).permute(0, 2, 1).contiguous()  # [Cout, Cin, k]

    @staticmethod
    def _convtr_w(w, cin, r):
        rc, _ = w.shape
        cout = rc // r
        w4 = w.reshape(r, cout, 2, cin)            # [phase, co, j, ci]
        wt = torch.empty(cin, cout, 2 * r, dtype=w.dtype)
        for tap in range(2 * r):
            phase, j = (tap, 1) if tap < r else (tap - r, 0)
            wt[:, :, tap] = w4[phase, :, j, :].T
        return wt

    def _causal_conv(self, x, w, b, cin, stride=1):
        # x [C, T] channels first
        wt = self._conv_w(w, cin)
        k = wt.shape[-1]
        T = x.shape[-1]
        To = -(-T // stride)
        right = To * stride - T
        xp = F.pad(x[None], (k - stride, right))
        return F.conv1d(xp, wt, b, stride=stride)[0]

    # ------------------------------------------------------------------ graphs
    def text_conditioner(self, ids):
        return self.w["text.embed"][torch.as_tensor(ids, dtype=torch.long)]

    def lm_full(self, voice, text, seq):
        """voice [Nv, D], text [T, D], seq [S, L] with NaN rows as BOS -> per-seq-row (cond [S, D], eos [S])."""
        s = self.s
        seq = torch.as_tensor(seq, dtype=DT)
        seq = torch.where(torch.isnan(seq), self.w["lm.bos_emb"][None, :], seq)
        x = torch.cat([torch.as_tensor(voice, dtype=DT), torch.as_tensor(text, dtype=DT),
                       seq @ self.w["lm.in_proj.w"].T], dim=0)
        x = self._transformer("lm.l", x, s.lm_layers, s.lm_heads, s.lm_ln_eps, s.lm_max_period, 0, False)
        out = F.layer_norm(x[-len(seq):], (s.d_model,), self.w["lm.out_norm.g"], self.w["lm.out_norm.b"], s.lm_ln_eps)
        eos = out @ self.w["lm.eos.w"][0] + self.w["lm.eos.b"][0]
        return out, eos

    def _time_embed(self, j, t):
        s = self.s
        half = s.flow_freq_dim // 2
        freq = torch.exp(-math.log(s.flow_max_period) * torch.arange(half, dtype=DT) / half)
        a = t * freq
        fe = torch.cat([torch.cos(a), torch.sin(a)])
        h = F.silu(fe @ self.w[f"flow.time{j}.l1.w"].T + self.w[f"flow.time{j}.l1.b"])
        return h @ self.w[f"flow.time{j}.l2.w"].T + self.w[f"flow.time{j}.l2.b"]

    def flow(self, c, sv, tv, x):
        s = self.s
        C = s.flow_dim
        c = torch.as_tensor(c, dtype=DT)
        x = torch.as_tensor(x, dtype=DT)
        h = x @ self.w["flow.in_proj.w"].T + self.w["flow.in_proj.b"]
        y = (self._time_embed(0, sv) + self._time_embed(1, tv)) / 2 + c @ self.w["flow.cond.w"].T + self.w["flow.cond.b"]
        sy = F.silu(y)
        for i in range(s.flow_blocks):
            p = f"flow.b{i}."
            shift, scale, gate = (sy @ self.w[p + "ada.w"].T + self.w[p + "ada.b"]).split(C)
            n = F.layer_norm(h, (C,), self.w[p + "ln.g"], self.w[p + "ln.b"], s.flow_ln_eps) * (1 + scale) + shift
            a = F.silu(n @ self.w[p + "fc1.w"].T + self.w[p + "fc1.b"])
            h = h + gate * (a @ self.w[p + "fc2.w"].T + self.w[p + "fc2.b"])
        shift, scale = (sy @ self.w["flow.final.ada.w"].T + self.w["flow.final.ada.b"]).split(C)
        n = F.layer_norm(h, (C,), None, None, s.flow_ln_eps) * (1 + scale) + shift
        return n @ self.w["flow.final.lin.w"].T + self.w["flow.final.lin.b"]

    def flow_euler(self, c, x0, steps):
        x = torch.as_tensor(x0, dtype=DT).clone()
        for j in range(steps):
            x = x + self.flow(c, j / steps, (j + 1) / steps, x) / steps
        return x

    def mimi_decode_full(self, latent):
        """latent [n, L] from stream start -> audio [n*frame]."""
        s = self.s
        M, U = s.mimi_dim, s.upsample
        z = torch.as_tensor(latent, dtype=DT) * self.w["mimi.emb_std"] + self.w["mimi.emb_mean"]
        q = (z @ self.w["dec.in_proj.w"].T).T  # [M, n]
        n = q.shape[1]
        y = F.conv_transpose1d(q[None], self.w["dec.up.w"].reshape(M, 1, 2 * U), stride=U, groups=M)[0][:, : n * U]
        y = self._transformer("dec.t", y.T, s.mimi_layers, s.mimi_heads, s.mimi_ln_eps, s.mimi_max_period,
                              s.mimi_context, True).T
        c = 8 * s.n_filters
        x = self._causal_conv(y, self.w["dec.conv0.w"], self.w["dec.conv0.b"], M)
        for si, r in enumerate(s.ratios):
            p = f"dec.s{si}."
            co = c // 2
            T = x.shape[1]
            wt = self._convtr_w(self.w[p + "up.w"], c, r)
            u = F.conv_transpose1d(F.elu(x)[None], wt, self.w[p + "up.b"], stride=r)[0][:, : T * r]
            h = self._causal_conv(F.elu(u), self.w[p + "res.c1.w"], self.w[p + "res.c1.b"], co)
            h = self._causal_conv(F.elu(h), self.w[p + "res.c2.w"], self.w[p + "res.c2.b"], co // 2)
            x = u + h
            c = co
        return self._causal_conv(F.elu(x), self.w["dec.final.w"], self.w["dec.final.b"], c)[0]

    def mimi_encode(self, audio):
        s = self.s
        M = s.mimi_dim
        c = s.n_filters
        x = self._causal_conv(torch.as_tensor(audio, dtype=DT)[None], self.w["enc.conv0.w"], self.w["enc.conv0.b"], 1)
        for si, r in enumerate(reversed(s.ratios)):
            p = f"enc.s{si}."
            h = self._causal_conv(F.elu(x), self.w[p + "res.c1.w"], self.w[p + "res.c1.b"], c)
            h = self._causal_conv(F.elu(h), self.w[p + "res.c2.w"], self.w[p + "res.c2.b"], c // 2)
            x = self._causal_conv(F.elu(x + h), self.w[p + "down.w"], self.w[p + "down.b"], c, stride=r)
            c *= 2
        y = self._causal_conv(F.elu(x), self.w["enc.final.w"], self.w["enc.final.b"], c)
        y = self._transformer("enc.t", y.T, s.mimi_layers, s.mimi_heads, s.mimi_ln_eps, s.mimi_max_period,
                              s.mimi_context, True).T
        d = self._causal_conv(y, self.w["enc.down.w"], None, M, stride=s.upsample)
        return d.T @ self.w["enc.spk_proj.w"].T
