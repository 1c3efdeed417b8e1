"""Restatement of `x_transformers.Encoder` as configured by the reference.

TEST INFRASTRUCTURE (oracle).  PARITY UNPINNED: `x_transformers` (declared
`>=1.27.20`, unpinned, /root/reference/modeling_utils/pyproject.toml:12) is a
third-party dependency that is neither vendored under /root/reference nor
installed in this image, and the reference holds no test or golden vector at
this boundary.  This file restates the library's published algorithm for the
exact keyword set the reference passes and *defines* the encoder arithmetic
for this build.

Call sites anchored on:
  * /root/reference/modeling_utils/modeling_utils/models/transformer.py:43-61
    -> Encoder(dim=3072, heads=8, depth=8, attn_dim_head=384, ff_mult=4,
       attn_flash=False, attn_dropout=0, ff_dropout=0, use_scalenorm=True,
       use_rmsnorm=False, rel_pos_bias=False, alibi_pos_bias=False,
       rotary_pos_emb=True, rotary_xpos=False, residual_attn=False,
       scale_residual=True, layer_dropout=0, cross_attend=False)
  * /root/reference/algonauts2025/model.py:109-111 (construction), :173 (call
    `self.encoder(x)` with x [B, T, dim], no mask).

Restated library behaviour (x_transformers 2.x `AttentionLayers`):
  * pre-norm residual architecture, layer types ('a', 'f') * depth, a final
    norm after the last layer;
  * norm = ScaleNorm: y = x / max(||x||_2, 1e-12) * sqrt(dim) * g, g scalar
    parameter initialised to 1 (`legacy_scalenorm=True` gives the 1.27-era
    form y = x / max(||x||_2, 1e-5) * g with g initialised to dim**-0.5);
  * Residual with `scale_residual`: out = branch(norm(x)) + x * residual_scale,
    residual_scale a [dim] parameter initialised to 1;
  * Attention: bias-free to_q / to_k / to_v / to_out Linear layers, heads x
    dim_head = 8 x 384, partial rotary embedding on the first
    rotary_emb_dim = max(dim_head // 2, 32) = 192 dims of q and k (theta =
    10000), softmax(q k^T * dim_head**-0.5) computed in fp32, no mask;
    `rotary_interleaved=True` is the 2.x pairing (2i, 2i+1); False is the
    1.27-era half-split pairing (i, i + rot/2);
  * FeedForward: Linear(dim, 4 dim) + exact (erf) GELU + Linear(4 dim, dim),
    both with bias.

State-dict keys follow the library so a real TRIBE checkpoint maps 1:1:
  layers.{2i}.0.0.g, layers.{2i}.1.to_{q,k,v,out}.weight,
  layers.{2i}.2.residual_scale, layers.{2i+1}.0.0.g,
  layers.{2i+1}.1.ff.0.0.{weight,bias}, layers.{2i+1}.1.ff.2.{weight,bias},
  layers.{2i+1}.2.residual_scale, final_norm.g, rotary_pos_emb.inv_freq.
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn


class ScaleNorm(nn.Module):
    def __init__(self, dim: int, legacy: bool = False):
        super().__init__()
        self.legacy = legacy
        self.dim = dim
        if legacy:
            self.eps = 1e-5
            self.g = nn.Parameter(torch.ones(1) * dim**-0.5)
        else:
            self.eps = 1e-12
            self.g = nn.Parameter(torch.ones(1))

    def gain(self) -> torch.Tensor:
        return self.g if self.legacy else self.g * self.dim**0.5

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        norm = torch.linalg.vector_norm(x, dim=-1, keepdim=True)
        return x / norm.clamp(min=self.eps) * self.gain()


class RotaryEmbedding(nn.Module):
    def __init__(self, dim: int, base: float = 10000.0, interleaved: bool = True):
        super().__init__()
        self.interleaved = interleaved
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq)

    def forward(self, seq_len: int) -> torch.Tensor:
        t = torch.arange(seq_len, device=self.inv_freq.device).type_as(self.inv_freq)
        freqs = torch.einsum("i,j->ij", t, self.inv_freq)  # [T, rot/2]
        if self.interleaved:
            return torch.stack((freqs, freqs), dim=-1).flatten(-2)  # f0 f0 f1 f1 ...
        return torch.cat((freqs, freqs), dim=-1)  # f0 f1 ... f0 f1 ...


def rotate_half(x: torch.Tensor, interleaved: bool) -> torch.Tensor:
    if interleaved:
        x = x.unflatten(-1, (-1, 2))
        x1, x2 = x.unbind(dim=-1)
        return torch.stack((-x2, x1), dim=-1).flatten(-2)
    half = x.shape[-1] // 2
    x1, x2 = x[..., :half], x[..., half:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rotary_pos_emb(t: torch.Tensor, freqs: torch.Tensor, interleaved: bool) -> torch.Tensor:
    """t: [B, h, T, d]; freqs: [T, rot]; rotates the first `rot` dims only."""
    rot = freqs.shape[-1]
    t_rot, t_pass = t[..., :rot], t[..., rot:]
    t_rot = t_rot * freqs.cos() + rotate_half(t_rot, interleaved) * freqs.sin()
    return torch.cat((t_rot, t_pass), dim=-1)


class Attention(nn.Module):
    def __init__(self, dim: int, heads: int, dim_head: int):
        super().__init__()
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head**-0.5
        inner = heads * dim_head
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_k = nn.Linear(dim, inner, bias=False)
        self.to_v = nn.Linear(dim, inner, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)

    def forward(self, x: torch.Tensor, freqs: torch.Tensor | None, interleaved: bool) -> torch.Tensor:
        B, T, _ = x.shape
        h, d = self.heads, self.dim_head
        q = self.to_q(x).view(B, T, h, d).transpose(1, 2)
        k = self.to_k(x).view(B, T, h, d).transpose(1, 2)
        v = self.to_v(x).view(B, T, h, d).transpose(1, 2)
        if freqs is not None:
            q = apply_rotary_pos_emb(q, freqs, interleaved)
            k = apply_rotary_pos_emb(k, freqs, interleaved)
        sim = torch.einsum("bhid,bhjd->bhij", q, k) * self.scale
        attn = sim.softmax(dim=-1, dtype=torch.float32).type(sim.dtype)
        out = torch.einsum("bhij,bhjd->bhid", attn, v)
        out = out.transpose(1, 2).reshape(B, T, h * d)
        return self.to_out(out)


class FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        inner = int(dim * mult)
        self.ff = nn.Sequential(
            nn.Sequential(nn.Linear(dim, inner), nn.GELU()),
            nn.Dropout(0.0),
            nn.Linear(inner, dim),
        )

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.ff(x)


class Residual(nn.Module):
    def __init__(self, dim: int, scale_residual: bool = True):
        super().__init__()
        self.residual_scale = nn.Parameter(torch.ones(dim)) if scale_residual else None

    def forward(self, x: torch.Tensor, residual: torch.Tensor) -> torch.Tensor:
        if self.residual_scale is not None:
            residual = residual * self.residual_scale
        return x + residual


class Encoder(nn.Module):
    """Drop-in for `x_transformers.Encoder(dim=..., **kwargs)` as the reference calls it."""

    def __init__(
        self,
        dim: int,
        depth: int = 8,
        heads: int = 8,
        attn_dim_head: int | None = None,
        ff_mult: int = 4,
        use_scalenorm: bool = True,
        rotary_pos_emb: bool = True,
        scale_residual: bool = True,
        rotary_interleaved: bool = True,
        legacy_scalenorm: bool = False,
        **unused,
    ):
        super().__init__()
        # kwargs the reference passes that select no-op behaviour here
        for key, want in dict(
            cross_attend=False, attn_flash=False, attn_dropout=0.0, ff_dropout=0.0,
            use_rmsnorm=False, rel_pos_bias=False, alibi_pos_bias=False,
            rotary_xpos=False, residual_attn=False, layer_dropout=0.0,
        ).items():
            if key in unused and unused[key] != want:
                raise NotImplementedError(f"{key}={unused[key]!r} is outside the restated path")
        if not use_scalenorm:
            raise NotImplementedError("only use_scalenorm=True is restated")
        self.dim = dim
        self.depth = depth
        self.heads = heads
        dim_head = attn_dim_head if attn_dim_head is not None else 64
        self.dim_head = dim_head
        self.rotary_interleaved = rotary_interleaved
        self.rotary_emb_dim = max(dim_head // 2, 32)
        self.rotary_pos_emb = (
            RotaryEmbedding(self.rotary_emb_dim, interleaved=rotary_interleaved)
            if rotary_pos_emb else None
        )
        layers = []
        for _ in range(depth):
            for kind in ("a", "f"):
                block = Attention(dim, heads, dim_head) if kind == "a" else FeedForward(dim, ff_mult)
                norms = nn.ModuleList([ScaleNorm(dim, legacy_scalenorm), None, None])
                layers.append(nn.ModuleList([norms, block, Residual(dim, scale_residual)]))
        self.layers = nn.ModuleList(layers)
        self.final_norm = ScaleNorm(dim, legacy_scalenorm)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        freqs = self.rotary_pos_emb(x.shape[1]) if self.rotary_pos_emb is not None else None
        for norms, block, residual in self.layers:
            inner = x
            x = norms[0](x)
            if isinstance(block, Attention):
                x = block(x, freqs, self.rotary_interleaved)
            else:
                x = block(x)
            x = residual(x, inner)
        return self.final_norm(x)


Decoder = Encoder  # the reference imports both names (transformer.py:44); causal is never set


def flops_per_token(dim: int, depth: int, ff_mult: int, seq_len: int) -> float:
    """Forward FLOPs per token."""
    return depth * (2 * 4 * dim * dim + 2 * 2 * dim * dim * ff_mult + 4 * seq_len * dim)


__all__ = ["Encoder", "Decoder", "ScaleNorm", "RotaryEmbedding", "Attention", "FeedForward",
           "Residual", "apply_rotary_pos_emb", "rotate_half", "flops_per_token"]

_ = (math, F)
