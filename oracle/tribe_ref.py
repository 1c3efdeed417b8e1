"""fp32 PyTorch-CPU restatement of the TRIBE trimodal-encode hot path.

TEST INFRASTRUCTURE (oracle) -- see oracle/__init__.py.  Every function cites
the reference lines it follows (paths relative to /root/reference).  Pinned by
tests/golden/*.npz, which were produced by executing the reference's own
model.py / common.py / losses.py / text.py (tests/golden/make_golden.py), and by
scipy.stats.pearsonr.  The encoder internals come from oracle/xt_encoder.py
(PARITY UNPINNED, third-party x_transformers absent).
"""

from __future__ import annotations

import dataclasses
import math
import typing as tp

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import xt_encoder

# ----------------------------------------------------------------------------
# SubjectLayers -- modeling_utils/modeling_utils/models/common.py:14-67
# ----------------------------------------------------------------------------


def subject_layers_fwd(
    x: torch.Tensor,  # [B, C, T]
    weights: torch.Tensor,  # [S, C, D]
    bias: torch.Tensor | None,  # [S, D]
    subjects: torch.Tensor,  # [B] or [B, 1] int64
) -> torch.Tensor:
    """common.py:45-67 (average_subjects=False branch, the one model.py:98-104 uses)."""
    B, C, T = x.shape
    N, C2, D = weights.shape
    assert C == C2
    if int(subjects.max()) >= N:  # common.py:53-55
        raise AssertionError("Subject index higher than number of subjects used to initialize the weights.")
    w = weights.index_select(0, subjects.flatten())  # common.py:61
    out = torch.einsum("bct,bcd->bdt", x, w)  # common.py:64
    if bias is not None:
        out = out + bias.index_select(0, subjects.flatten()).view(B, D, 1)  # common.py:63,65-66
    return out


class SubjectLayersRef(nn.Module):
    """common.py:16-43: N(0,1)/sqrt(in_channels) init for weights and bias."""

    def __init__(self, in_channels: int, out_channels: int, n_subjects: int, bias: bool = False):
        super().__init__()
        self.weights = nn.Parameter(torch.randn(n_subjects, in_channels, out_channels) / in_channels**0.5)
        self.bias = nn.Parameter(torch.randn(n_subjects, out_channels) / in_channels**0.5) if bias else None

    def forward(self, x: torch.Tensor, subjects: torch.Tensor) -> torch.Tensor:
        return subject_layers_fwd(x, self.weights, self.bias, subjects)


# ----------------------------------------------------------------------------
# nn.AdaptiveAvgPool1d -- model.py:60,119-120
# ----------------------------------------------------------------------------


def adaptive_pool_windows(t_in: int, t_out: int) -> list[tuple[int, int]]:
    """ATen adaptive pooling windows: [floor(i*T/T'), ceil((i+1)*T/T'))."""
    return [((i * t_in) // t_out, -((-(i + 1) * t_in) // t_out)) for i in range(t_out)]


def adaptive_avg_pool1d(x: torch.Tensor, t_out: int) -> torch.Tensor:
    wins = adaptive_pool_windows(x.shape[-1], t_out)
    return torch.stack([x[..., a:b].mean(dim=-1) for a, b in wins], dim=-1)


# ----------------------------------------------------------------------------
# FmriEncoder -- algonauts2025/model.py:46-174
# ----------------------------------------------------------------------------


@dataclasses.dataclass
class EncoderDims:
    """model.py hard-codes hidden=3072 (:61), depth=8 / heads=8 (:109-111), pos table 1024 (:106)."""

    hidden: int = 3072
    depth: int = 8
    heads: int = 8
    max_len: int = 1024
    ff_mult: int = 4
    rotary_interleaved: bool = True
    legacy_scalenorm: bool = False


def prepare_modality(data: torch.Tensor, layer_aggregation: str) -> torch.Tensor:
    """model.py:146-156: [B,L,D,T] (or [B,D,T]) -> [B,T,L*D] ('cat') / [B,T,D] ('mean'), fp32."""
    data = data.to(torch.float32)
    if data.ndim == 3:
        data = data.unsqueeze(1)
    if layer_aggregation == "mean":
        data = data.mean(dim=1)
    elif layer_aggregation == "cat":
        data = data.flatten(1, 2)  # "b l d t -> b (l d) t"
    data = data.transpose(1, 2)
    assert data.ndim == 3
    return data


class FmriEncoderRef(nn.Module):
    """Restated FmriEncoder; parameter names equal the reference's state_dict keys."""

    def __init__(
        self,
        feature_dims: dict[str, tuple[int, int] | None],
        n_outputs: int,
        n_output_timesteps: int,
        n_subjects: int,
        feature_aggregation: str = "cat",
        layer_aggregation: str = "cat",
        subject_embedding: bool = False,
        contrastive_modalities: tp.Sequence[str] = (),
        contrastive_temperature: float = 0.07,
        dims: EncoderDims | None = None,
    ):
        super().__init__()
        dims = dims or EncoderDims()
        self.dims = dims
        self.feature_dims = feature_dims
        self.n_outputs = n_outputs
        self.n_output_timesteps = n_output_timesteps
        self.feature_aggregation = feature_aggregation
        self.layer_aggregation = layer_aggregation
        self.contrastive_modalities = list(contrastive_modalities)
        self.contrastive_temperature = contrastive_temperature
        hidden = dims.hidden
        self.projectors = nn.ModuleDict()
        self.contrastive_heads = nn.ModuleDict()
        for modality, tup in feature_dims.items():  # model.py:62-91
            if tup is None:
                continue
            num_layers, feature_dim = tup
            input_dim = feature_dim * num_layers if layer_aggregation == "cat" else feature_dim
            output_dim = hidden // len(feature_dims) if feature_aggregation == "cat" else hidden
            # MlpConfig(...).build degenerates to a bare Linear: common.py:124-128
            self.projectors[modality] = nn.Linear(input_dim, output_dim)
            if modality in self.contrastive_modalities:
                self.contrastive_heads[modality] = nn.Linear(input_dim, hidden)
        self.predictor = SubjectLayersRef(hidden, n_outputs, n_subjects, bias=True)  # model.py:98-104
        self.time_pos_embed = nn.Parameter(torch.randn(1, dims.max_len, hidden))  # model.py:106
        if subject_embedding:
            self.subject_embed = nn.Embedding(n_subjects, hidden)  # model.py:107-108
        self.encoder = xt_encoder.Encoder(  # model.py:109-111 -> transformer.py:43-61
            dim=hidden, depth=dims.depth, heads=dims.heads, attn_dim_head=hidden // dims.heads,
            ff_mult=dims.ff_mult, rotary_interleaved=dims.rotary_interleaved,
            legacy_scalenorm=dims.legacy_scalenorm,
        )

    # model.py:125-165 (eval mode; `dropped` reproduces a given modality-dropout draw)
    def aggregate_features(self, data: dict[str, torch.Tensor], dropped: tp.Collection[str] = ()) -> torch.Tensor:
        for modality in data.keys():
            if modality in self.feature_dims:
                break
        x = data[modality]
        B, T = x.shape[0], x.shape[-1]
        tensors = []
        for modality in self.feature_dims.keys():
            if modality not in self.projectors:
                t = torch.zeros(B, T, self.dims.hidden // len(self.feature_dims))  # model.py:143-144
            else:
                t = self.projectors[modality](prepare_modality(data[modality], self.layer_aggregation))
                if modality in dropped:
                    t = torch.zeros_like(t)  # model.py:158-159
            tensors.append(t)
        if self.feature_aggregation == "cat":
            return torch.cat(tensors, dim=-1)
        return sum(tensors)

    # model.py:167-174
    def transformer_forward(self, x: torch.Tensor, subject_id: torch.Tensor | None) -> torch.Tensor:
        x = x + self.time_pos_embed[:, : x.size(1)]
        if hasattr(self, "subject_embed"):
            x = x + self.subject_embed(subject_id)
        return self.encoder(x)

    # model.py:113-123
    def forward(self, data: dict[str, torch.Tensor], pool_outputs: bool = True) -> torch.Tensor:
        x = self.aggregate_features(data)
        subject_id = data.get("subject_id", None)
        x = self.transformer_forward(x, subject_id)
        x = x.transpose(1, 2)
        x = self.predictor(x, subject_id)
        if pool_outputs:
            return adaptive_avg_pool1d(x, self.n_output_timesteps)
        return x

    # model.py:177-241
    def compute_contrastive_loss(self, data: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
        brain = self.transformer_forward(self.aggregate_features(data), data.get("subject_id", None))
        out = {}
        for modality in self.contrastive_modalities:
            if modality not in self.contrastive_heads or modality not in data:
                continue
            lat = self.contrastive_heads[modality](prepare_modality(data[modality], self.layer_aggregation))
            if lat.size(1) != brain.size(1):
                lat = adaptive_avg_pool1d(lat.transpose(1, 2), brain.size(1)).transpose(1, 2)
            out[modality] = info_nce(brain, lat, self.contrastive_temperature)
        return out


def info_nce(q: torch.Tensor, k: torch.Tensor, tau: float = 0.07) -> torch.Tensor:
    """model.py:208-221: symmetric InfoNCE over flattened [B,T,H] sequences."""
    bt, h = q.shape[0] * q.shape[1], q.shape[2]
    q = F.normalize(q.reshape(bt, h), dim=-1)
    k = F.normalize(k.reshape(bt, h), dim=-1)
    logits = (q @ k.t()) / tau
    labels = torch.arange(bt)
    return 0.5 * (F.cross_entropy(logits, labels) + F.cross_entropy(logits.t(), labels))


# ----------------------------------------------------------------------------
# pl_module.BrainModule._run_step -- algonauts2025/pl_module.py:46-107
# ----------------------------------------------------------------------------


def flatten_bt(y: torch.Tensor) -> torch.Tensor:
    """pl_module.py:54-55: rearrange 'b d t -> (b t) d'."""
    return y.permute(0, 2, 1).reshape(-1, y.shape[1])


def run_step(
    y_pred: torch.Tensor, y_true: torch.Tensor, subject_id: torch.Tensor, loss: str = "mse"
) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """pl_module.py:47-56: returns (loss, y_pred_flat, y_true_flat, subject_ids_flat)."""
    subject_ids_flat = subject_id.repeat_interleave(y_pred.shape[2], 0)  # :52
    p, t = flatten_bt(y_pred), flatten_bt(y_true)
    val = mse_loss(p, t) if loss == "mse" else pearson_loss(p, t)
    return val, p, t, subject_ids_flat


def mse_loss(pred: torch.Tensor, true: torch.Tensor) -> torch.Tensor:
    """nn.MSELoss() (defaults.py:125 via losses/base.py:43-59): mean over all elements."""
    return ((pred - true) ** 2).mean()


def pearson_loss(x: torch.Tensor, y: torch.Tensor, reduction: str = "mean", dim: int = 1) -> torch.Tensor:
    """modeling_utils/modeling_utils/losses/losses.py:17-42."""
    x = x.transpose(0, dim).reshape(x.shape[dim], -1)
    y = y.transpose(0, dim).reshape(y.shape[dim], -1)
    x = x - x.mean(dim=1, keepdim=True)
    y = y - y.mean(dim=1, keepdim=True)
    cov = (x * y).sum(dim=1)
    pcc = cov / ((x**2).sum(dim=1).sqrt() * (y**2).sum(dim=1).sqrt() + 1e-8)
    loss = 1 - pcc
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    raise ValueError(f"Invalid reduction: {reduction}")


# ----------------------------------------------------------------------------
# Per-voxel Pearson -- algonauts2025/main.py:459-477 (scipy) and
# modeling_utils/modeling_utils/metrics/base.py:26-29 (torchmetrics, unpinned)
# ----------------------------------------------------------------------------


def pearson_stats(pred: torch.Tensor, true: torch.Tensor) -> torch.Tensor:
    """Sufficient statistics [5, V] in fp64: sum x, sum y, sum x^2, sum y^2, sum xy over rows of [N, V]."""
    x, y = pred.double(), true.double()
    return torch.stack([x.sum(0), y.sum(0), (x * x).sum(0), (y * y).sum(0), (x * y).sum(0)])


def pearson_from_stats(stats: torch.Tensor, n: int) -> torch.Tensor:
    sx, sy, sxx, syy, sxy = stats.double()
    cov = sxy - sx * sy / n
    vx = sxx - sx * sx / n
    vy = syy - sy * sy / n
    return cov / (vx * vy).sqrt()


def scipy_pearson_columns(pred: np.ndarray, true: np.ndarray) -> np.ndarray:
    """main.py:474-477 verbatim in behaviour: scipy pearsonr(trues[:, p], preds[:, p]) per parcel, fp32 out."""
    from scipy.stats import pearsonr

    out = np.zeros((true.shape[1]), dtype=np.float32)
    for p in range(len(out)):
        out[p] = pearsonr(true[:, p], pred[:, p])[0]
    return out


class StreamingPearson:
    """torchmetrics.PearsonCorrCoef(num_outputs=V) update/compute restated (PARITY UNPINNED:
    torchmetrics is absent).  metrics/base.py:26-29 takes .mean() of compute()."""

    def __init__(self, num_outputs: int):
        z = lambda: torch.zeros(num_outputs, dtype=torch.float32)  # noqa: E731
        self.mean_x, self.mean_y, self.var_x, self.var_y, self.corr_xy, self.n_total = z(), z(), z(), z(), z(), z()

    def update(self, preds: torch.Tensor, target: torch.Tensor) -> None:
        n_obs = preds.shape[0]
        cond = bool(self.n_total.mean() > 0) or n_obs == 1
        if cond:
            mx_new = (self.n_total * self.mean_x + preds.sum(0)) / (self.n_total + n_obs)
            my_new = (self.n_total * self.mean_y + target.sum(0)) / (self.n_total + n_obs)
        else:
            mx_new, my_new = preds.mean(0), target.mean(0)
        self.n_total = self.n_total + n_obs
        if cond:
            self.var_x = self.var_x + ((preds - mx_new) * (preds - self.mean_x)).sum(0)
            self.var_y = self.var_y + ((target - my_new) * (target - self.mean_y)).sum(0)
        else:
            self.var_x = self.var_x + preds.var(0) * (n_obs - 1)
            self.var_y = self.var_y + target.var(0) * (n_obs - 1)
        self.corr_xy = self.corr_xy + ((preds - mx_new) * (target - self.mean_y)).sum(0)
        self.mean_x, self.mean_y = mx_new, my_new

    def compute_per_output(self) -> torch.Tensor:
        nb = self.n_total
        var_x, var_y, corr_xy = self.var_x / (nb - 1), self.var_y / (nb - 1), self.corr_xy / (nb - 1)
        return torch.clamp(corr_xy / (var_x * var_y).sqrt(), -1.0, 1.0)

    def compute(self) -> torch.Tensor:
        return self.compute_per_output().mean()


# ----------------------------------------------------------------------------
# Feature-cache layer grouping -- data_utils/data_utils/features/text.py:129-149
# (identical in audio.py:123-143 and video.py:147-167)
# ----------------------------------------------------------------------------


def aggregate_layers(latents: np.ndarray, layers: tp.Sequence[float], layer_aggregation: str | None) -> np.ndarray:
    idx = np.unique([int(i * (latents.shape[0] - 1)) for i in layers]).tolist()
    if len(idx) == 1:
        return latents[idx[0]][None, :] if layer_aggregation is None else latents[idx[0]]
    if layer_aggregation == "group_mean":
        idx[-1] += 1
        return np.stack([latents[a:b].mean(0) for a, b in zip(idx[:-1], idx[1:])])
    if layer_aggregation is None:
        return latents[idx]
    raise ValueError(f"Unknown layer aggregation: {layer_aggregation}")


# ----------------------------------------------------------------------------
# Deterministic parameters / synthetic inputs shared by golden maker, tests, bench
# ----------------------------------------------------------------------------


def fill_params_(module: nn.Module, seed: int = 0) -> None:
    """Overwrite every parameter with a seeded, name-keyed draw so that two
    differently-constructed modules with equal state_dict keys get equal values,
    independent of construction order.  Scales follow the reference inits:
    Linear U(-1/sqrt(fan_in), 1/sqrt(fan_in)); predictor N(0,1)/sqrt(C)
    (common.py:37-42); time_pos_embed N(0,1) (model.py:106); ScaleNorm g and
    residual_scale perturbed around their init of 1 so they are exercised."""
    import zlib

    sd = module.state_dict()
    for name, p in sd.items():
        if name.endswith("inv_freq"):
            continue
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        if name.endswith(".g"):
            base = float(p.flatten()[0])  # 1 (2.x) or dim**-0.5 (legacy)
            v = base * (1.0 + 0.1 * (torch.rand(p.shape, generator=g) - 0.5))
        elif name.endswith("residual_scale"):
            v = 1.0 + 0.2 * (torch.rand(p.shape, generator=g) - 0.5)
        elif name == "time_pos_embed":
            v = torch.randn(p.shape, generator=g)
        elif name.startswith("predictor."):
            c = sd["predictor.weights"].shape[1]
            v = torch.randn(p.shape, generator=g) / c**0.5
        elif name == "subject_embed.weight":
            v = torch.randn(p.shape, generator=g)
        elif p.ndim == 2:  # Linear weight [out, in]
            bound = 1.0 / math.sqrt(p.shape[1])
            v = (torch.rand(p.shape, generator=g) * 2 - 1) * bound
        elif p.ndim == 1:  # Linear bias: fan_in from the sibling weight
            w = sd[name[: -len("bias")] + "weight"]
            bound = 1.0 / math.sqrt(w.shape[1])
            v = (torch.rand(p.shape, generator=g) * 2 - 1) * bound
        else:
            raise RuntimeError(f"unhandled parameter {name} {tuple(p.shape)}")
        p.copy_(v.to(p.dtype))


def synthetic_batch(
    B: int, T: int, feature_dims: dict[str, tuple[int, int] | None], n_subjects: int, seed: int = 0,
    dtype: torch.dtype = torch.float32,
) -> dict[str, torch.Tensor]:
    """N(0,1) features [B,L,D,T]; subject ids 0..S-1 repeating."""
    g = torch.Generator().manual_seed(seed)
    data: dict[str, torch.Tensor] = {}
    for m, tup in feature_dims.items():
        if tup is None:
            continue
        L, D = tup
        data[m] = torch.randn(B, L, D, T, generator=g).to(dtype)
    data["subject_id"] = (torch.arange(B) % n_subjects).view(B, 1)
    return data
