"""Transformer encoder config + HIP-backed encoder module.

API mirror of /root/reference/modeling_utils/modeling_utils/models/transformer.py:16-61:
`TransformerEncoderConfig(...).build(dim)` validates `dim % heads == 0`, `dim >= 256` and returns
an `nn.Module` mapping [B, T, dim] -> [B, T, dim].  The reference returns
`x_transformers.Encoder(dim=dim, attn_dim_head=dim // heads, **fields)`; this build returns
`HipEncoder`, which keeps the library's parameter names (so a TRIBE checkpoint's `encoder.*`
keys load unchanged) and runs the arithmetic in gfx950 HIP kernels (tribe_encoder_fwd).

Encoder semantics (third-party, restated -- see oracle/xt_encoder.py for the definition and
its "parity unpinned" status): pre-ScaleNorm, bias-free q/k/v/out projections, partial rotary on
the first max(dim_head // 2, 32) dims, fp32 softmax, scaled residual, GELU(erf) feed-forward,
final ScaleNorm.
"""

from __future__ import annotations

import logging
import typing as tp

import pydantic
import torch
from torch import nn

from tribe_hip import ops

from .._pack import PackCache, f32c

logger = logging.getLogger(__name__)


class ScaleNorm(nn.Module):
    def __init__(self, dim: int, legacy: bool = False):
        super().__init__()
        self.dim, self.legacy = dim, legacy
        self.g = nn.Parameter(torch.ones(1) * (dim**-0.5 if legacy else 1.0))

    @property
    def gain_scale(self) -> float:
        return 1.0 if self.legacy else self.dim**0.5

    @property
    def eps(self) -> float:
        return 1e-5 if self.legacy else 1e-12


class _Attention(nn.Module):  # parameter holder with the library's names
    def __init__(self, dim: int, heads: int, dim_head: int):
        super().__init__()
        inner = heads * dim_head
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_k = nn.Linear(dim, inner, bias=False)
        self.to_v = nn.Linear(dim, inner, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)


class _FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int):
        super().__init__()
        inner = int(dim * mult)
        self.ff = nn.Sequential(nn.Sequential(nn.Linear(dim, inner), nn.GELU()), nn.Dropout(0.0), nn.Linear(inner, dim))


class _Residual(nn.Module):
    def __init__(self, dim: int, scale_residual: bool):
        super().__init__()
        self.residual_scale = nn.Parameter(torch.ones(dim)) if scale_residual else None


class _Rotary(nn.Module):
    def __init__(self, dim: int, base: float = 10000.0):
        super().__init__()
        self.register_buffer("inv_freq", 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim)))


class HipEncoder(nn.Module):
    def __init__(self, dim: int, depth: int, heads: int, dim_head: int, ff_mult: int = 4, rotary_pos_emb: bool = True,
                 scale_residual: bool = True, rotary_interleaved: bool = True, legacy_scalenorm: bool = False):
        super().__init__()
        self.dim, self.depth, self.heads, self.dim_head, self.ff_mult = dim, depth, heads, dim_head, ff_mult
        self.rotary_interleaved = rotary_interleaved
        self.rotary_emb_dim = max(dim_head // 2, 32) if rotary_pos_emb else 0
        if rotary_pos_emb:
            self.rotary_pos_emb = _Rotary(self.rotary_emb_dim)
        layers = []
        for _ in range(depth):
            for kind in ("a", "f"):
                block = _Attention(dim, heads, dim_head) if kind == "a" else _FeedForward(dim, ff_mult)
                layers.append(nn.ModuleList([nn.ModuleList([ScaleNorm(dim, legacy_scalenorm), None, None]), block,
                                             _Residual(dim, scale_residual)]))
        self.layers = nn.ModuleList(layers)
        self.final_norm = ScaleNorm(dim, legacy_scalenorm)
        self._packs = PackCache()

    # -- packed weights ---------------------------------------------------------------------
    def _sources(self) -> list[torch.Tensor | None]:
        return [p for p in self.parameters()]

    def rotary_tables(self, T: int, device: torch.device) -> tuple[torch.Tensor | None, torch.Tensor | None, torch.Tensor | None]:
        """(cos, sin, -sin) f32 [T, rot_dim / 2] for the training path: the same tables EncoderPack.tables builds, WITHOUT packing the weights
        (asking `packed()` for them re-packed every encoder weight after every optimiser step -- 32 casts and 8 f32 q|k|v concatenations
        per step that only the inference entry point needs)."""
        if not self.rotary_emb_dim:
            return None, None, None
        key = (T, device.index or 0)
        cache = self.__dict__.setdefault("_rotary_tabs", {})
        if key not in cache:
            t = torch.arange(T, device=device, dtype=torch.float32)
            freqs = torch.einsum("i,j->ij", t, self.rotary_pos_emb.inv_freq.to(device=device, dtype=torch.float32))
            cos, sin = freqs.cos().contiguous(), freqs.sin().contiguous()
            cache[key] = (cos, sin, (-sin).contiguous())
        return cache[key]

    def packed(self) -> ops.EncoderPack:
        def build() -> ops.EncoderPack:
            pack = ops.EncoderPack(self.dim, self.depth, self.heads, self.dim_head, int(self.dim * self.ff_mult),
                                   self.rotary_emb_dim, self.rotary_interleaved, self.final_norm.gain_scale, self.final_norm.eps)
            keep = pack.keep

            def own(t: torch.Tensor) -> int:
                keep.append(t)
                return t.data_ptr()

            for i in range(self.depth):
                norms_a, attn, ares = self.layers[2 * i]
                norms_f, ff, fres = self.layers[2 * i + 1]
                an, fn = norms_a[0], norms_f[0]
                L = pack.layers[i]
                wqkv = torch.cat([f32c(attn.to_q.weight), f32c(attn.to_k.weight), f32c(attn.to_v.weight)], dim=0)
                L.attn_norm_g = own(f32c(an.g))
                L.w_qkv = own(ops.pack_weight(wqkv))
                L.w_out = own(ops.pack_weight(f32c(attn.to_out.weight)))
                L.attn_res_scale = own(f32c(ares.residual_scale)) if ares.residual_scale is not None else None
                L.ff_norm_g = own(f32c(fn.g))
                L.w_ff1 = own(ops.pack_weight(f32c(ff.ff[0][0].weight)))
                L.b_ff1 = own(f32c(ff.ff[0][0].bias))
                L.w_ff2 = own(ops.pack_weight(f32c(ff.ff[2].weight)))
                L.b_ff2 = own(f32c(ff.ff[2].bias))
                L.ff_res_scale = own(f32c(fres.residual_scale)) if fres.residual_scale is not None else None
            pack.final_norm_g = f32c(self.final_norm.g)
            if self.rotary_emb_dim:
                pack.inv_freq = self.rotary_pos_emb.inv_freq
            return pack

        return self._packs.get("enc", self._sources(), build)

    # -- forward ----------------------------------------------------------------------------
    def forward_tokens(self, x2d: torch.Tensor, B: int, T: int, out_dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
        """x2d f32 [B*T, dim], CONSUMED (the residual stream is updated in place) -> y [B*T, dim]."""
        return ops.encoder_fwd(x2d, self.packed(), B, T, out_dtype)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, T, D = x.shape
        if D != self.dim:
            raise ValueError(f"HipEncoder: expected last dim {self.dim}, got {D}")
        x2d = x.detach().to(torch.float32).reshape(B * T, D).clone()
        return self.forward_tokens(x2d, B, T, torch.float32).view(B, T, D)


class TransformerEncoderConfig(pydantic.BaseModel):
    model_config = pydantic.ConfigDict(extra="forbid")
    name: tp.Literal["TransformerEncoder"] = "TransformerEncoder"
    heads: int = 8
    depth: int = 12
    cross_attend: bool = False
    causal: bool = False
    attn_flash: bool = False
    attn_dropout: float = 0.1
    ff_mult: int = 4
    ff_dropout: float = 0.0
    use_scalenorm: bool = True
    use_rmsnorm: bool = False
    rel_pos_bias: bool = False
    alibi_pos_bias: bool = False
    rotary_pos_emb: bool = True
    rotary_xpos: bool = False
    residual_attn: bool = False
    scale_residual: bool = True
    layer_dropout: float = 0.0

    # Not in the reference config: which generation of x_transformers the restated encoder follows
    # (see oracle/xt_encoder.py).  Defaults = the 2.x behaviour.
    rotary_interleaved: bool = True
    legacy_scalenorm: bool = False

    def build(self, dim: int) -> nn.Module:
        if dim % self.heads != 0:
            raise ValueError(f"dim ({dim}) must be divisible by the number of heads ({self.heads})")
        if dim < 256:
            raise ValueError(f"dim ({dim}) is less than 256, which causes weird bug in x-transformers")
        unsupported = {
            "cross_attend": self.cross_attend, "causal": self.causal, "use_rmsnorm": self.use_rmsnorm,
            "rel_pos_bias": self.rel_pos_bias, "alibi_pos_bias": self.alibi_pos_bias, "rotary_xpos": self.rotary_xpos,
            "residual_attn": self.residual_attn, "not use_scalenorm": not self.use_scalenorm,
        }
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(f"HipEncoder covers the TRIBE configuration only; unsupported options set: {bad}")
        if self.layer_dropout or self.ff_dropout:
            raise NotImplementedError("layer_dropout / ff_dropout are 0 on the TRIBE path (model.py:109-111)")
        # attn_dropout only acts in training; model.py passes 0.0.  attn_flash selects a kernel, not a result.
        return HipEncoder(dim=dim, depth=self.depth, heads=self.heads, dim_head=dim // self.heads, ff_mult=self.ff_mult,
                          rotary_pos_emb=self.rotary_pos_emb, scale_residual=self.scale_residual,
                          rotary_interleaved=self.rotary_interleaved, legacy_scalenorm=self.legacy_scalenorm)
