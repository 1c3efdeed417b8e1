"""Loss registry (mirror of /root/reference/modeling_utils/modeling_utils/losses/__init__.py)."""
import typing as tp

import pydantic

from .base import BaseLossConfig, PearsonLossConfig, TorchLossConfig  # noqa: F401
from .losses import MSELoss, PearsonLoss  # noqa: F401

LossConfig = tp.Annotated[tp.Union[PearsonLossConfig, TorchLossConfig], pydantic.Field(discriminator="name")]
