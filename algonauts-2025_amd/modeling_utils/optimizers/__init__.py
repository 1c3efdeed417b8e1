"""Optimiser configs (reference: modeling_utils/modeling_utils/optimizers/__init__.py)."""

from .base import BaseLRSchedulerConfig, BaseOptimizerConfig, LightningOptimizerConfig, TorchLRSchedulerConfig, TorchOptimizerConfig

OptimizerConfig = TorchOptimizerConfig

__all__ = ["BaseLRSchedulerConfig", "BaseOptimizerConfig", "LightningOptimizerConfig", "OptimizerConfig", "TorchLRSchedulerConfig",
           "TorchOptimizerConfig"]
