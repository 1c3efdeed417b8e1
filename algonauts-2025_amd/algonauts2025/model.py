"""FmriEncoder: trimodal fusion encoder + per-subject voxel head, on MI355X HIP kernels.

Drop-in for /root/reference/algonauts2025/model.py: same config fields and defaults
(`FmriEncoderConfig`, model.py:20-33), same constructor / `build` signature (:35-53), same
module API (`forward(batch, pool_outputs)`, `aggregate_features`, `transformer_forward`,
`compute_contrastive_loss`, ...) and the same parameter names, so `state_dict()` keys
(`projectors.<m>.{weight,bias}`, `predictor.{weights,bias}`, `time_pos_embed`,
`subject_embed.weight`, `encoder.*`, `contrastive_heads.<m>.*`) interchange with the reference.

Data flow of `forward` (every step a HIP launch through libtribe_hip.so, nothing in torch):
  features [B,L,D,T] --tribe_pack_features--> bf16 [B*T, L*D]          (model.py:146-155)
    --tribe_projector_fwd x3 (MFMA GEMM; epilogue = bias + column slice of the fused stream
      + time_pos_embed (+ subject_embed))--> f32 x [B*T, 3072]          (model.py:157-172)
    --tribe_encoder_fwd (8 layers)--> bf16 y [B*T, 3072]                  (model.py:173)
    --tribe_voxel_head_fwd (grouped-by-subject MFMA GEMM)--> f32 [B,V,T]  (model.py:117-118)
    --tribe_adaptive_avg_pool_fwd--> f32 [B,V,T']                          (model.py:119-120)
fp32 master parameters live in torch; bf16 packed copies are cached per parameter version.
In `.eval()` / no-grad mode this fused path runs and records no autograd graph; in `.train()` mode with grad enabled
`forward` composes the same arithmetic from the autograd functions of modeling_utils/autograd.py (HIP forward AND backward).
"""

from __future__ import annotations

import typing as tp

import numpy as np
import pydantic
import torch
from torch import nn

from data_utils.dataloader import SegmentData
from modeling_utils._pack import PackCache, f32c
from modeling_utils.models.common import MlpConfig, SubjectLayers
from modeling_utils.models.transformer import TransformerEncoderConfig
from tribe_hip import ops


class FmriEncoderConfig(pydantic.BaseModel):
    model_config = pydantic.ConfigDict(extra="forbid")
    name: tp.Literal["FmriEncoder"] = "FmriEncoder"
    n_subjects: int | None = None
    feature_aggregation: tp.Literal["sum", "cat"] = "cat"
    layer_aggregation: tp.Literal["mean", "cat"] = "cat"
    subject_embedding: bool = False
    modality_dropout: float = 0.0

    contrastive_enabled: bool = False
    contrastive_modalities: list[str] = ["video"]
    contrastive_weight: float = 0.1
    contrastive_temperature: float = 0.07

    # Extensions (absent from the reference, whose model.py hard-codes them at :61, :106, :109-111);
    # the defaults reproduce the reference exactly, smaller values exist for tests.
    hidden: int = 3072
    depth: int = 8
    heads: int = 8
    max_timesteps: int = 1024
    rotary_interleaved: bool = True
    legacy_scalenorm: bool = False
    # training: let the contrastive pass reuse the prediction pass's latents whenever its own dropout draw, the inputs and the
    # parameters coincide (bit-identical loss; False = always re-run the encoder as model.py:228 does)
    share_contrastive_latents: bool = True

    def build(self, feature_dims: dict[str, tuple[int, int] | None], n_outputs: int, n_output_timesteps: int) -> nn.Module:
        return FmriEncoder(feature_dims, n_outputs, n_output_timesteps, config=self)


class FmriEncoder(nn.Module):
    def __init__(self, feature_dims: dict[str, tuple[int, int] | None], n_outputs: int, n_output_timesteps: int,
                 config: FmriEncoderConfig):
        super().__init__()
        self.config = config
        self.feature_dims = feature_dims
        self.n_outputs = n_outputs
        self.n_output_timesteps = n_output_timesteps
        self.projectors = nn.ModuleDict()
        self.contrastive_heads = nn.ModuleDict()
        hidden = self.hidden = config.hidden
        for modality, tup in feature_dims.items():
            if tup is None:
                print(f"Warning: {modality} has no feature dimensions. Skipping projector.")
                continue
            num_layers, feature_dim = tup
            input_dim = feature_dim * num_layers if config.layer_aggregation == "cat" else feature_dim
            output_dim = hidden // len(feature_dims) if config.feature_aggregation == "cat" else hidden
            self.projectors[modality] = MlpConfig(norm_layer="layer", activation_layer="gelu", dropout=0.0).build(input_dim, output_dim)
            if config.contrastive_enabled and modality in config.contrastive_modalities:
                self.contrastive_heads[modality] = MlpConfig(norm_layer="layer", activation_layer="gelu", dropout=0.0).build(input_dim, hidden)
        self.combiner = nn.Identity()
        self.predictor = SubjectLayers(in_channels=hidden, out_channels=n_outputs, n_subjects=config.n_subjects,
                                       average_subjects=False, bias=True)
        self.time_pos_embed = nn.Parameter(torch.randn(1, config.max_timesteps, hidden))
        if config.subject_embedding:
            self.subject_embed = nn.Embedding(config.n_subjects, hidden)
        self.encoder = TransformerEncoderConfig(attn_dropout=0.0, ff_dropout=0.0, layer_dropout=0.0, depth=config.depth,
                                                heads=config.heads, rotary_interleaved=config.rotary_interleaved,
                                                legacy_scalenorm=config.legacy_scalenorm).build(dim=hidden)
        self._packs = PackCache()

    # ------------------------------------------------------------------------------------------
    def _batch_dict(self, batch: SegmentData | dict) -> dict[str, torch.Tensor]:
        return batch.data if hasattr(batch, "data") else batch

    def _pack(self, feat: tp.Any) -> torch.Tensor:
        """features -> bf16 [B*T, K_pad] projector operand (model.py:146-155).  Batches from the GPU segment loader
        (data_utils/gpu_loader.py: `PackedFeature`) already are in that layout and skip the pass."""
        mean = self.config.layer_aggregation == "mean"
        if hasattr(feat, "packed"):
            if not mean or feat.L == 1:
                return feat.packed
            feat = feat.unpack()
        if feat.ndim not in (3, 4):
            raise AssertionError(f"expected [B, L, D, T] or [B, D, T] features, got {tuple(feat.shape)}")
        return ops.pack_features(feat.contiguous(), layer_mean=mean)

    def _packed_linear(self, key: str, lin: nn.Linear) -> tuple[torch.Tensor, torch.Tensor]:
        return self._packs.get(key, [lin.weight, lin.bias], lambda: (ops.pack_weight(f32c(lin.weight)), f32c(lin.bias)))

    def _draw_modality_dropout(self) -> list[str]:
        # model.py:133-141 -- same RNG consumption order as the reference
        drop = []
        for modality in self.feature_dims.keys():
            if torch.rand(1).item() < self.config.modality_dropout and self.training:
                drop.append(modality)
        if len(drop) == len(self.feature_dims):
            drop = list(np.random.choice(drop, len(drop) - 1, replace=False))
        return drop

    def _fused_embed(self, data: dict[str, torch.Tensor], add_embeddings: bool) -> tuple[torch.Tensor, int, int]:
        """aggregate_features (+ transformer_forward's embedding adds when `add_embeddings`): f32 [B*T, hidden]."""
        for modality in data.keys():
            if modality in self.feature_dims:
                break
        ref = data[modality]
        B, T = ref.shape[0], ref.shape[-1]
        device = ref.device
        cat = self.config.feature_aggregation == "cat"
        n_mod = len(self.feature_dims)
        if not cat and any(m not in self.projectors for m in self.feature_dims):
            # reference behaviour: a 3072//n zero block cannot be summed with hidden-wide projections (model.py:143-144,163-164)
            raise RuntimeError(f"The size of tensor a ({self.hidden}) must match the size of tensor b ({self.hidden // n_mod}) "
                               "at non-singleton dimension 2")
        if add_embeddings and T > self.time_pos_embed.shape[1]:
            raise RuntimeError(f"sequence length {T} exceeds the positional table ({self.time_pos_embed.shape[1]})")
        dropped = self._draw_modality_dropout()
        pos = f32c(self.time_pos_embed)[0] if add_embeddings else None
        semb, sid = None, None
        if add_embeddings and hasattr(self, "subject_embed"):
            semb = f32c(self.subject_embed.weight)
            # the projector epilogue gathers subject_embed rows by this index with no bounds check on the device:
            # validate 0 <= id < n_subjects here (nn.Embedding raises IndexError in the reference, model.py:171-172)
            sid = self.predictor.check_subjects(data["subject_id"])
            if self.predictor.average_subjects:
                sid = data["subject_id"].flatten().to(torch.int64).contiguous()
        width = (self.hidden // n_mod) * n_mod if cat else self.hidden
        x = torch.empty(B * T, width, dtype=torch.float32, device=device)
        slot = self.hidden // n_mod
        first = True
        for i, modality in enumerate(self.feature_dims.keys()):
            col0 = i * slot if cat else 0
            n_out = slot if cat else self.hidden
            emb = (pos, semb, sid) if (cat or first) else (None, None, None)  # 'sum': embeddings are added once
            if modality not in self.projectors or modality in dropped:
                if cat or first:
                    ops.projector_zero_fwd(B * T, T, n_out, x, col0, *emb)
                # 'sum' with a dropped non-first modality contributes nothing
            else:
                feat = data[modality]
                if feat.ndim not in (3, 4):
                    raise AssertionError(f"expected [B, L, D, T] or [B, D, T] features, got {tuple(feat.shape)}")
                packed = self._pack(feat)
                w, b = self._packed_linear(f"proj.{modality}", self.projectors[modality])
                ops.projector_fwd(packed, T, w, b, n_out, x, col0, accumulate=(not cat and not first), pos_embed=emb[0],
                                  subj_embed=emb[1], subject_id=emb[2])
            first = False
        return x, B, T

    # -- reference API ---------------------------------------------------------------------------
    def aggregate_features(self, batch: SegmentData | dict) -> torch.Tensor:
        """model.py:125-165 -> f32 [B, T, hidden]."""
        x, B, T = self._fused_embed(self._batch_dict(batch), add_embeddings=False)
        return x.view(B, T, -1)

    def transformer_forward(self, x: torch.Tensor, subject_id: torch.Tensor | None = None) -> torch.Tensor:
        """model.py:167-174 on an explicit [B, T, hidden] tensor (the fused `forward` never takes this detour)."""
        x = x + self.time_pos_embed[:, : x.size(1)].to(x.device)
        if hasattr(self, "subject_embed"):
            x = x + self.subject_embed(subject_id)
        return self.encoder(x)

    def _latents(self, data: dict[str, torch.Tensor], out_dtype: torch.dtype) -> tuple[torch.Tensor, int, int]:
        x, B, T = self._fused_embed(data, add_embeddings=True)
        return self.encoder.forward_tokens(x, B, T, out_dtype), B, T

    # -- training path: same arithmetic, every op an autograd function whose forward AND backward are HIP kernels ----
    def _wants_grad(self) -> bool:
        # Lightning runs training_step in train mode with grad enabled; evaluation (.eval()) keeps the fused fast path
        return self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def _latents_key(self, data: dict[str, torch.Tensor], dropped: list[str]) -> tuple:
        """Everything the latents of a training pass depend on: the inputs (storage, version, shape), the modality-dropout draw and the
        version of every parameter -- two passes with equal keys compute bit-identical latents (the kernels are deterministic)."""
        inputs = tuple((k, v.data_ptr(), v._version, tuple(v.shape)) for k, v in data.items()
                       if (k in self.feature_dims or k == "subject_id") and isinstance(v, torch.Tensor))
        return inputs, tuple(dropped), tuple(p._version for p in self.parameters())

    def _latents_autograd(self, data: dict[str, torch.Tensor], share: str = "") -> tuple[torch.Tensor, int, int]:
        """aggregate_features + transformer_forward up to (not including) the final ScaleNorm: f32 [B*T, hidden].

        `share`: the reference's training step runs the encoder TWICE -- once for the prediction (model.py:113-123) and once more
        inside compute_contrastive_loss -> get_brain_latents (model.py:177-182, 228), each with its own modality-dropout draw.
        Whenever the two draws coincide (always at modality_dropout == 0; 19.5 % of the steps at the default 0.3 with three
        modalities) the second pass recomputes the first bit for bit.  The prediction pass ("produce") therefore leaves its latents
        behind and the contrastive pass ("consume") -- after drawing ITS dropout set from the same RNG stream the reference
        consumes -- takes them when inputs, draw and parameter versions all match, and recomputes otherwise.  The loss is
        bit-identical either way; autograd sums the two branches' gradients at the shared node."""
        dropped = self._draw_modality_dropout()
        key = self._latents_key(data, dropped) if share else None
        if share == "consume":
            kept, self._shared_latents = getattr(self, "_shared_latents", None), None
            if kept is not None and kept[0] == key and self.config.share_contrastive_latents:
                self.shared_latent_hits = getattr(self, "shared_latent_hits", 0) + 1
                return kept[1]
        out = self._latents_autograd_compute(data, dropped)
        self._shared_latents = (key, out) if share == "produce" else None
        return out

    def _latents_autograd_compute(self, data: dict[str, torch.Tensor], dropped: list[str]) -> tuple[torch.Tensor, int, int]:
        from modeling_utils import autograd as ag

        cfg = self.config
        cat = cfg.feature_aggregation == "cat"
        for modality in data.keys():
            if modality in self.feature_dims:
                break
        ref = data[modality]
        B, T = ref.shape[0], ref.shape[-1]
        n_mod = len(self.feature_dims)
        slot = self.hidden // n_mod if cat else self.hidden
        if not cat and any(m not in self.projectors for m in self.feature_dims):
            # reference behaviour: a hidden // n zero block cannot be summed with hidden-wide projections (model.py:143-144,163-164)
            raise RuntimeError(f"The size of tensor a ({self.hidden}) must match the size of tensor b ({self.hidden // n_mod}) "
                               "at non-singleton dimension 2")
        slices = []
        for m in self.feature_dims.keys():
            if m not in self.projectors or m in dropped:  # model.py:143-144,158-159: zero block, no gradient
                slices.append(torch.zeros(B * T, slot, dtype=torch.float32, device=ref.device))
                continue
            packed = self._pack(data[m])
            lin = self.projectors[m]
            slices.append(ag.ProjectorFuse.apply(packed, lin.weight, lin.bias))
        if cat:
            x = torch.cat(slices, dim=1).view(B, T, n_mod * slot)   # model.py:161-162
        else:
            x = slices[0]
            for extra in slices[1:]:                                # model.py:163-164: sum(tensors)
                x = x + extra
            x = x.view(B, T, slot)
        x = x + self.time_pos_embed[:, :T]                      # model.py:169-170 (autograd sums the rows over the batch)
        subject_id = data["subject_id"]
        if hasattr(self, "subject_embed"):
            x = x + self.subject_embed(subject_id)              # model.py:171-172
        x = x.reshape(B * T, -1).contiguous()
        enc = self.encoder
        cos, sin, neg_sin = enc.rotary_tables(T, x.device)     # neg_sin: the backward's rotation by -theta
        gs, eps = enc.final_norm.gain_scale, enc.final_norm.eps
        scale = enc.dim_head**-0.5
        for i in range(enc.depth):
            norms_a, attn, res_a = enc.layers[2 * i]
            norms_f, ff, res_f = enc.layers[2 * i + 1]
            # the block input forks into norm(x) and the scaled residual; ScaleNormFork's backward sums the two gradients in its own kernel
            xn, xr = ag.ScaleNormFork.apply(x, norms_a[0].g, gs, eps, res_a.residual_scale)
            qkv = ag.QKVLinear.apply(xn, attn.to_q.weight, attn.to_k.weight, attn.to_v.weight)
            if enc.rotary_emb_dim:
                ao, _ = ag.RotaryAttention.apply(qkv, cos, sin, neg_sin, B, T, enc.heads, enc.dim_head, scale, enc.rotary_emb_dim,
                                                 enc.rotary_interleaved)
            else:
                ao = ag.Attention.apply(qkv, B, T, enc.heads, enc.dim_head, scale)
            x = ag.Linear.apply(ao, attn.to_out.weight, None, xr, res_a.residual_scale, True, True)
            xn, xr = ag.ScaleNormFork.apply(x, norms_f[0].g, gs, eps, res_f.residual_scale)
            x = ag.FeedForward.apply(xn, ff.ff[0][0].weight, ff.ff[0][0].bias, ff.ff[2].weight, ff.ff[2].bias, xr, res_f.residual_scale, True)
        return x, B, T

    def _forward_autograd(self, data: dict[str, torch.Tensor], pool_outputs: bool) -> torch.Tensor:
        from modeling_utils import autograd as ag

        x, B, T = self._latents_autograd(data, share="produce" if self.config.contrastive_enabled else "")
        enc = self.encoder
        y = ag.ScaleNorm.apply(x, enc.final_norm.g, enc.final_norm.gain_scale, enc.final_norm.eps)
        out = ag.VoxelHead.apply(y.view(B, T, -1), self.predictor.weights, self.predictor.bias,
                                 self.predictor.check_subjects(data["subject_id"]))
        if pool_outputs and T != self.n_output_timesteps:
            out = ag.AdaptivePool.apply(out, self.n_output_timesteps)
        return out

    def forward(self, batch: SegmentData | dict, pool_outputs: bool = True) -> torch.Tensor:
        data = self._batch_dict(batch)
        if self._wants_grad():
            return self._forward_autograd(data, pool_outputs)
        with torch.no_grad():
            return self._forward_inference(data, pool_outputs)

    def _forward_inference(self, data: dict[str, torch.Tensor], pool_outputs: bool = True) -> torch.Tensor:
        y, B, T = self._latents(data, torch.bfloat16)  # [B*T, hidden] bf16, final-normed
        out = self.predictor.forward_tokens(y.view(B, T, -1), data["subject_id"])  # [B, V, T] f32
        if pool_outputs and T != self.n_output_timesteps:  # AdaptiveAvgPool1d(T)(x[..., T]) is the identity
            out = ops.adaptive_avg_pool(out, self.n_output_timesteps)
        return out

    # --- contrastive alignment helpers (model.py:177-241) -----------------------------------------
    @torch.no_grad()
    def get_brain_latents(self, batch: SegmentData | dict) -> torch.Tensor:
        y, B, T = self._latents(self._batch_dict(batch), torch.float32)
        return y.view(B, T, -1)

    @torch.no_grad()
    def get_modality_latents(self, batch: SegmentData | dict, modality: str) -> torch.Tensor:
        assert modality in self.contrastive_heads, f"No contrastive head found for modality '{modality}'"
        data = self._batch_dict(batch)
        feat = data.get(modality, None)
        if feat is None:
            raise KeyError(f"Modality '{modality}' not found in batch.data")
        B, T = feat.shape[0], feat.shape[-1]
        packed = self._pack(feat)
        w, b = self._packed_linear(f"chead.{modality}", self.contrastive_heads[modality])
        out = torch.empty(B * T, self.hidden, dtype=torch.float32, device=feat.device)
        ops.projector_fwd(packed, T, w, b, self.hidden, out, 0, False, None, None, None)
        return out.view(B, T, -1)

    @staticmethod
    def _info_nce(q: torch.Tensor, k: torch.Tensor, tau: float = 0.07) -> torch.Tensor:
        """model.py:208-221: symmetric InfoNCE over flattened [B, T, H] sequences (HIP: row normalisation, logits MFMA
        GEMM, log-sum-exp kernels; differentiable through modeling_utils.autograd.InfoNCE)."""
        from modeling_utils import autograd as ag

        bt, h = q.shape[0] * q.shape[1], q.shape[2]
        return ag.InfoNCE.apply(q.reshape(bt, h).float().contiguous(), k.reshape(bt, h).float().contiguous(), tau)

    def _contrastive_autograd(self, data: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
        """model.py:223-241 on the training path: a SECOND pass through aggregate_features + encoder (fresh
        modality-dropout draws, as the reference's get_brain_latents does, model.py:177-182,228), the contrastive heads
        and the symmetric InfoNCE -- all differentiable HIP functions."""
        from modeling_utils import autograd as ag

        x, B, T = self._latents_autograd(data, share="consume")
        enc = self.encoder
        brain = ag.ScaleNorm.apply(x, enc.final_norm.g, enc.final_norm.gain_scale, enc.final_norm.eps, True)  # f32 [B*T, H]
        losses: dict[str, torch.Tensor] = {}
        for modality in self.config.contrastive_modalities:
            if modality not in self.contrastive_heads or modality not in data:
                continue
            feat = data[modality]
            packed = self._pack(feat)
            head = self.contrastive_heads[modality]
            lat = ag.ProjectorFuse.apply(packed, head.weight, head.bias)            # f32 [B * T_mod, hidden]
            T_mod = feat.shape[-1]
            if T_mod != T:   # model.py:234-238: adaptive average pool of the modality latents to the brain's time axis
                lat = lat.view(B, T_mod, -1).transpose(1, 2).contiguous()           # [B, hidden, T_mod]
                lat = ag.AdaptivePool.apply(lat, T).transpose(1, 2).reshape(B * T, -1).contiguous()
            losses[modality] = ag.InfoNCE.apply(brain, lat, self.config.contrastive_temperature)
        return losses

    def compute_contrastive_loss(self, batch: SegmentData | dict) -> dict[str, torch.Tensor]:
        if not self.config.contrastive_enabled:
            return {}
        data = self._batch_dict(batch)
        if self._wants_grad():
            return self._contrastive_autograd(data)
        brain = self.get_brain_latents(batch)
        losses: dict[str, torch.Tensor] = {}
        for modality in self.config.contrastive_modalities:
            if modality not in self.contrastive_heads or modality not in data:
                continue
            lat = self.get_modality_latents(batch, modality)
            if lat.size(1) != brain.size(1):
                lat = ops.adaptive_avg_pool(lat.transpose(1, 2).contiguous(), brain.size(1)).transpose(1, 2)
            losses[modality] = self._info_nce(brain, lat, tau=self.config.contrastive_temperature)
        return losses
