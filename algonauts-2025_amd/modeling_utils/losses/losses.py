"""HIP-backed losses of the TRIBE path.

`PearsonLoss` mirrors /root/reference/modeling_utils/modeling_utils/losses/losses.py:11-42
(1 - per-column Pearson with eps 1e-8, mean | sum over columns).  `MSELoss` is the HIP
counterpart of `torch.nn.MSELoss()` selected by defaults.py:125.

Both accept what the reference passes -- two [N, V] matrices, columns = voxels -- and, as the
fast path used by BrainModule, the un-flattened [B, V, T'] pair via `forward_bvt` (the '(b t) d'
flatten of pl_module.py:54-55 is a pure re-indexing of the same sums and is never materialised).
Both are differentiable (HIP backward kernels, modeling_utils/autograd.py).
"""

from __future__ import annotations

import torch
from torch import nn

from tribe_hip import ops


def _as_bvt(x: torch.Tensor, dim: int) -> torch.Tensor:
    """[N, V] (voxels along `dim`) -> strided [1, V, N] view, no copy."""
    if x.ndim != 2:
        x = x.transpose(0, dim).reshape(x.shape[dim], -1).t()  # reference semantics for >2-D inputs (copying)
        dim = 1
    v = x if dim == 1 else x.t()
    return v.t().unsqueeze(0)  # [1, V, N]


class PearsonLoss(nn.Module):
    def __init__(self, reduction: str = "mean", dim: int = 1):
        super().__init__()
        self.reduction = reduction
        self.dim = dim

    def forward_bvt(self, pred: torch.Tensor, true: torch.Tensor) -> torch.Tensor:
        if self.reduction not in ("mean", "sum"):
            raise ValueError(f"Invalid reduction: {self.reduction}")
        if torch.is_grad_enabled() and pred.requires_grad:
            from ..autograd import PearsonLossFn

            return PearsonLossFn.apply(pred.float(), true.float(), self.reduction)
        return ops.pearson_loss(pred.float(), true.float(), self.reduction)

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        return self.forward_bvt(_as_bvt(x.float(), self.dim), _as_bvt(y.float(), self.dim))


class MSELoss(nn.Module):
    """mean((pred - true)^2) over all elements (torch.nn.MSELoss(reduction='mean'))."""

    def __init__(self, reduction: str = "mean"):
        super().__init__()
        if reduction != "mean":
            raise NotImplementedError("the HIP MSELoss implements reduction='mean' (the only one the reference configures)")
        self.reduction = reduction

    def forward(self, pred: torch.Tensor, true: torch.Tensor) -> torch.Tensor:
        if pred.shape != true.shape:
            raise ValueError(f"MSELoss: shape mismatch {tuple(pred.shape)} vs {tuple(true.shape)}")
        if torch.is_grad_enabled() and pred.requires_grad:
            from ..autograd import MSE

            return MSE.apply(pred.float(), true.float())
        return ops.mse(pred.float().contiguous(), true.float().contiguous())

    forward_bvt = forward
