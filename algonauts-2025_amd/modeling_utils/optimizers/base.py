"""Optimiser / LR-schedule configs of the training loop, HIP-backed where a kernel exists.

Mirror of /root/reference/modeling_utils/modeling_utils/optimizers/base.py:25-96 (`TorchOptimizerConfig`,
`TorchLRSchedulerConfig`, `LightningOptimizerConfig`): same field names, same `build` signatures, same result
dictionary (`{"optimizer": ..., "lr_scheduler": {"scheduler": ..., "interval": ...}}`), so the `optim` block of
grids/defaults.py:126-141 validates and builds unchanged.  What differs is what gets built: `name="Adam"` / `"AdamW"`
resolve to `modeling_utils.optim.HipAdam` (one HIP launch per step, torch.optim.Adam's arithmetic and state_dict keys)
whenever the parameters live on the GPU; every other optimiser name, and every scheduler, is torch's own -- schedulers
only edit `param_groups` on the host, which HipAdam reads at each step.
"""

from __future__ import annotations

import inspect
import typing as tp

import pydantic
import torch
from torch import optim


def _known(base: type) -> dict[str, type]:
    """Public classes of torch.optim(.lr_scheduler) deriving from `base`, by name."""
    module = optim.lr_scheduler if base is optim.lr_scheduler.LRScheduler else optim
    found = {}
    for name in dir(module):
        obj = getattr(module, name)
        if inspect.isclass(obj) and issubclass(obj, base) and obj is not base and not name.startswith("_"):
            found[name] = obj
    return found


def _check_kwargs(target: type, kwargs: dict[str, tp.Any], implied: tuple[str, ...]) -> None:
    """Fail at config time on unknown or missing constructor arguments (the reference validates the same way, base.py:45-47)."""
    sig = inspect.signature(target.__init__)
    has_var_kw = any(p.kind is inspect.Parameter.VAR_KEYWORD for p in sig.parameters.values())
    unknown = [k for k in kwargs if k not in sig.parameters]
    if unknown and not has_var_kw:
        raise ValueError(f"{target.__name__} does not take {unknown}")
    missing = [n for n, p in sig.parameters.items()
               if n != "self" and p.default is inspect.Parameter.empty and p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)
               and n not in kwargs and n not in implied]
    if missing:
        raise ValueError(f"{target.__name__} needs {missing}")


class _NamedConfig(pydantic.BaseModel):
    """`{name: ...}` with unknown keys rejected -- the shape of every config block in this family."""

    model_config = pydantic.ConfigDict(extra="forbid")
    name: str

    @classmethod
    def _catalogue(cls) -> dict[str, type]:
        return {}

    @pydantic.field_validator("name")
    @classmethod
    def _name_is_known(cls, v: str) -> str:
        known = cls._catalogue()
        if known and v not in known:
            raise ValueError(f"unknown name {v!r} for {cls.__name__}")
        return v


class BaseLRSchedulerConfig(_NamedConfig):
    def build(self, optimizer: optim.Optimizer) -> optim.lr_scheduler.LRScheduler:
        raise NotImplementedError


class TorchLRSchedulerConfig(BaseLRSchedulerConfig):
    """`{name, kwargs}` (base.py:62-78); `build(optimizer, **build_kwargs)` lets the caller add `total_steps` (pl_module.py:141-143)."""

    kwargs: dict[str, tp.Any] = {}

    @classmethod
    def _catalogue(cls) -> dict[str, type]:
        return _known(optim.lr_scheduler.LRScheduler)

    def model_post_init(self, _ctx: tp.Any) -> None:
        _check_kwargs(self._catalogue()[self.name], self.kwargs, implied=("optimizer",))

    def build(self, optimizer: optim.Optimizer, **build_kwargs: tp.Any) -> optim.lr_scheduler.LRScheduler:
        merged = {**self.kwargs, **build_kwargs}
        return self._catalogue()[self.name](optimizer, **merged)


class BaseOptimizerConfig(_NamedConfig):
    def build(self, params: tp.Iterable[torch.Tensor]) -> optim.Optimizer:
        raise NotImplementedError


class TorchOptimizerConfig(BaseOptimizerConfig):
    """`{name, lr, kwargs}` (base.py:34-50).  Adam / AdamW over GPU parameters run as one HIP launch per step."""

    lr: float
    kwargs: dict[str, tp.Any] = {}
    HIP_BACKED: tp.ClassVar[tuple[str, ...]] = ("Adam", "AdamW")

    @classmethod
    def _catalogue(cls) -> dict[str, type]:
        return _known(optim.Optimizer)

    def model_post_init(self, _ctx: tp.Any) -> None:
        if "lr" in self.kwargs:
            raise ValueError("lr should be defined as a base parameter instead of within kwargs.")
        _check_kwargs(self._catalogue()[self.name], self.kwargs, implied=("params", "lr"))

    def _on_hip(self, params: list[torch.Tensor]) -> bool:
        plain = set(self.kwargs) <= {"betas", "eps", "weight_decay"}
        on_gpu = bool(params) and all(isinstance(p, torch.Tensor) and p.is_cuda and p.dtype == torch.float32 for p in params)
        return self.name in self.HIP_BACKED and plain and on_gpu

    def build(self, params: tp.Iterable[torch.Tensor]) -> optim.Optimizer:
        params = list(params)
        if not self._on_hip(params):
            return self._catalogue()[self.name](params, lr=self.lr, **self.kwargs)
        from modeling_utils.optim import HipAdam

        extra = dict(self.kwargs)
        if self.name == "AdamW":
            extra.setdefault("weight_decay", 1e-2)     # torch.optim.AdamW's default
        return HipAdam(params, lr=self.lr, decoupled_weight_decay=self.name == "AdamW", **extra)


class LightningOptimizerConfig(pydantic.BaseModel):
    """Optimiser + optional schedule in the dictionary shape Lightning's `configure_optimizers` expects (base.py:81-96)."""

    model_config = pydantic.ConfigDict(extra="forbid")
    name: tp.Literal["LightningOptimizer"] = "LightningOptimizer"
    optimizer: TorchOptimizerConfig
    scheduler: TorchLRSchedulerConfig | None = None
    interval: tp.Literal["step", "epoch"] = "step"

    def build(self, params: tp.Iterable[torch.Tensor], **scheduler_build_kwargs: tp.Any) -> dict[str, tp.Any]:
        opt = self.optimizer.build(params)
        if self.scheduler is None:
            return {"optimizer": opt}
        schedule = self.scheduler.build(opt, **scheduler_build_kwargs)
        return {"optimizer": opt, "lr_scheduler": {"scheduler": schedule, "interval": self.interval}}
