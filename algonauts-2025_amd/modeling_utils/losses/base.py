"""Loss configs (mirror of /root/reference/modeling_utils/modeling_utils/losses/base.py:26-59):
`{name, kwargs}` -> `.build()` -> nn.Module(pred, true) -> scalar.  The reference derives the
custom-loss configs with `convert_to_pydantic`; here the one custom loss is spelled out."""

from __future__ import annotations

import inspect
import typing as tp

import pydantic
from torch import nn
from torch.nn.modules.loss import _Loss

from . import losses


def _all_subclasses(cls: type) -> set[type]:
    out = set()
    for sub in cls.__subclasses__():
        out.add(sub)
        out |= _all_subclasses(sub)
    return out


TORCHLOSS_NAMES = sorted({c.__name__ for c in _all_subclasses(_Loss)})


class BaseLossConfig(pydantic.BaseModel):
    model_config = pydantic.ConfigDict(extra="forbid")
    name: str

    def build(self) -> nn.Module:
        raise NotImplementedError


class PearsonLossConfig(BaseLossConfig):
    name: tp.Literal["PearsonLoss"] = "PearsonLoss"
    reduction: str = "mean"
    dim: int = 1

    def build(self) -> nn.Module:
        return losses.PearsonLoss(reduction=self.reduction, dim=self.dim)


class TorchLossConfig(BaseLossConfig):
    name: tp.Literal[tuple(TORCHLOSS_NAMES)]  # type: ignore[valid-type]
    kwargs: dict[str, tp.Any] = {}

    def model_post_init(self, log__: tp.Any) -> None:
        super().model_post_init(log__)
        params = inspect.signature(getattr(nn, self.name).__init__).parameters
        unknown = set(self.kwargs) - set(params)
        if unknown:
            raise ValueError(f"Unknown kwargs for nn.{self.name}: {sorted(unknown)}")

    def build(self, **kwargs: tp.Any) -> nn.Module:
        if overlap := set(self.kwargs) & set(kwargs):
            raise ValueError(f"Build kwargs overlap with config kwargs for keys: {overlap}.")
        kwargs = self.kwargs | kwargs
        if self.name == "MSELoss" and kwargs.get("reduction", "mean") == "mean" and not (set(kwargs) - {"reduction"}):
            return losses.MSELoss()  # the default loss (defaults.py:125) runs in HIP
        # SmoothL1Loss / HuberLoss etc. (run_ensemble.py grids): stock torch modules, outside the HIP scope
        return getattr(nn, self.name)(**kwargs)
