"""Streaming per-voxel Pearson metrics, HIP-backed.

Mirror of /root/reference/modeling_utils/modeling_utils/metrics/base.py:
  * `MultidimPearsonCorrCoef(num_outputs)` (base.py:26-29): `update(preds[N, V], target[N, V])`,
    `compute()` -> mean over outputs of the per-output Pearson r, `reset()`;
  * `GroupedMetric(metric_name, kwargs)` (base.py:39-91): one sub-metric per group id,
    `update(preds, target, groups)`, `compute()` -> {group_id: value}.
The reference inherits the running mean/var/cov update of `torchmetrics.PearsonCorrCoef`
(third-party, absent here -> "parity unpinned" for the streaming form); this build accumulates the
f64 sufficient statistics {Sx, Sy, Sxx, Syy, Sxy, n} per (group, voxel) on the GPU in one kernel
(tribe_pearson_stats_update) -- algebraically the same r, pinned against scipy.stats.pearsonr,
which is what the reference's own evaluation uses (main.py:459-477).  The pandas groupby of
base.py:67-78 is replaced by passing the group index of every batch row to the kernel.
`sync()` all-reduces the statistics over the process group (RCCL) for multi-GPU evaluation.
"""

from __future__ import annotations

import typing as tp

import pydantic
import torch
from torch import nn

from tribe_hip import ops


class _PearsonState(nn.Module):
    def __init__(self, num_outputs: int, n_groups: int = 1):
        super().__init__()
        self.num_outputs, self.n_groups = num_outputs, n_groups
        self.stats: torch.Tensor | None = None  # f64 [G, V, 6], created on first update (device follows the data)

    def _ensure(self, device: torch.device, n_groups: int) -> None:
        if self.stats is None or self.stats.device != device:
            self.stats = torch.zeros(max(n_groups, self.n_groups), self.num_outputs, 6, dtype=torch.float64, device=device)
        elif self.stats.shape[0] < n_groups:
            grown = torch.zeros(n_groups, self.num_outputs, 6, dtype=torch.float64, device=device)
            grown[: self.stats.shape[0]] = self.stats
            self.stats = grown
        self.n_groups = self.stats.shape[0]

    @staticmethod
    def _as_bvt(x: torch.Tensor) -> torch.Tensor:
        if x.ndim == 3:
            return x.float()
        if x.ndim != 2:
            raise ValueError(f"expected [N, V] or [B, V, T], got {tuple(x.shape)}")
        return x.float().t().unsqueeze(0)  # [1, V, N] strided view of the flattened matrix

    def update_bvt(self, preds: torch.Tensor, target: torch.Tensor, group: torch.Tensor | None = None, n_groups: int = 1) -> None:
        p, t = self._as_bvt(preds), self._as_bvt(target)
        if p.shape[1] != self.num_outputs:
            raise ValueError(f"expected {self.num_outputs} outputs, got {p.shape[1]}")
        self._ensure(p.device, n_groups)
        ops.pearson_stats_update(self.stats, p, t, group)

    def sync(self, group: tp.Any = None) -> None:
        import torch.distributed as dist

        if self.stats is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.stats, op=dist.ReduceOp.SUM, group=group)

    def per_output(self) -> torch.Tensor:
        if self.stats is None:
            raise RuntimeError("compute() called before update()")
        return ops.pearson_from_stats(self.stats)  # [G, V]

    def reset(self) -> None:
        self.stats = None


class MultidimPearsonCorrCoef(_PearsonState):
    def __init__(self, num_outputs: int = 1, **_: tp.Any):
        super().__init__(num_outputs, 1)

    def update(self, preds: torch.Tensor, target: torch.Tensor) -> None:
        self.update_bvt(preds, target)

    def compute(self) -> torch.Tensor:
        return self.per_output()[0].mean()


class OnlinePearsonCorr(MultidimPearsonCorrCoef):
    """metrics.py:16-63: same statistic with `dim` / `reduction` options."""

    def __init__(self, dim: int = 0, reduction: str | None = "mean"):
        super().__init__(1)
        self.dim, self.reduction, self._initialized = dim, reduction, False

    def update(self, preds: torch.Tensor, target: torch.Tensor) -> None:
        if self.dim == 1:
            preds, target = preds.T, target.T
        if not self._initialized:
            self.num_outputs, self._initialized = preds.shape[1], True
        super().update(preds, target)

    def compute(self) -> torch.Tensor:
        r = self.per_output()[0]
        return r.mean() if self.reduction == "mean" else r.sum() if self.reduction == "sum" else r

    def reset(self) -> None:
        self._initialized = False
        super().reset()


_BASE_METRICS = {"MultidimPearsonCorrCoef": MultidimPearsonCorrCoef, "OnlinePearsonCorr": OnlinePearsonCorr}


class GroupedMetric(nn.Module):
    def __init__(self, metric_name: str, kwargs: dict[str, tp.Any] | None = None) -> None:
        super().__init__()
        assert metric_name in _BASE_METRICS, f"Metric {metric_name} not found"
        self.base_metric_cls = _BASE_METRICS[metric_name]
        self.metric_kwargs = kwargs or {}
        self._state: _PearsonState | None = None
        self._ids: list[str] = []  # group keys in first-seen order (groupby(sort=False), base.py:67)
        self._index: dict[str, int] = {}

    def update(self, preds: torch.Tensor, target: torch.Tensor, groups: tp.Optional[torch.Tensor] = None) -> None:
        """preds/target [N, V] with groups [N], or (fast path) [B, V, T] with groups [B] / [B, 1]."""
        n_rows = preds.shape[0]
        if groups is None:
            groups = torch.zeros(n_rows, dtype=torch.int64)
        groups = groups.flatten()
        assert len(groups) == n_rows, f"Groups must be the same shape as preds/target, got {groups.shape} and {preds.shape}"
        labels = groups.tolist()
        for lab in labels:
            key = str(lab)
            if key not in self._index:
                self._index[key] = len(self._ids)
                self._ids.append(key)
        slot = torch.tensor([self._index[str(lab)] for lab in labels], dtype=torch.int64, device=preds.device)
        if self._state is None:
            self._state = _PearsonState(preds.shape[1])
        if preds.ndim == 2:  # [N, V]: every row is its own "batch row" of length T = 1
            preds, target = preds.float().unsqueeze(-1), target.float().unsqueeze(-1)
        self._state.update_bvt(preds, target, slot, len(self._ids))

    def compute(self) -> dict[str, float]:
        if self._state is None:
            return {}
        r = self._state.per_output()  # [G, V]
        means = r.mean(dim=1).tolist()
        return {gid: means[self._index[gid]] for gid in self._ids}

    def sync(self, group: tp.Any = None) -> None:
        if self._state is not None:
            self._state.sync(group)

    def reset(self) -> None:
        if self._state is not None:
            self._state.reset()

    def __repr__(self) -> str:
        return f"GroupedMetric({self.base_metric_cls.__name__})"


class BaseMetricConfig(pydantic.BaseModel):
    model_config = pydantic.ConfigDict(extra="forbid")
    log_name: str
    name: str

    def build(self) -> nn.Module:
        raise NotImplementedError


class MultidimPearsonCorrCoefConfig(BaseMetricConfig):
    """TorchMetricConfig shape of the reference (base.py:112-127): {log_name, name, kwargs}."""

    name: tp.Literal["MultidimPearsonCorrCoef"] = "MultidimPearsonCorrCoef"
    kwargs: dict[str, tp.Any] = {}

    def build(self) -> nn.Module:
        return MultidimPearsonCorrCoef(**self.kwargs)


class GroupedMetricConfig(BaseMetricConfig):
    name: tp.Literal["GroupedMetric"] = "GroupedMetric"
    metric_name: str
    kwargs: dict[str, tp.Any] | None = None

    def build(self) -> nn.Module:
        return GroupedMetric(metric_name=self.metric_name, kwargs=self.kwargs)
