"""BrainModule: the per-batch step of the reference's LightningModule on the HIP path.

Mirror of /root/reference/algonauts2025/pl_module.py:19-144 (`forward`, `_run_step`,
`training_step`, `validation_step`, `test_step`, `on_*_epoch_end`).  `lightning` is not a
dependency here: the class is a plain `nn.Module` exposing the same methods, so a Lightning
user subclasses `(BrainModule, pl.LightningModule)` while the benchmark and
the tests drive the steps directly.  `self.log` calls are recorded in `self.logged`.

_run_step arithmetic (pl_module.py:46-107) without the two transposing copies: predictions and
targets stay [B, V, T']; loss and metric kernels index them as the '(b t) d' flatten would.
"""

from __future__ import annotations

import typing as tp
from pathlib import Path

import torch
from torch import nn

from data_utils.dataloader import SegmentData


class BrainModule(nn.Module):
    def __init__(self, model: nn.Module, loss: nn.Module, optim_config: tp.Any, metrics: dict[str, tp.Any],
                 max_epochs: int = 100, checkpoint_path: Path | None = None, config: dict[str, tp.Any] | None = None) -> None:
        super().__init__()
        self.model = model
        self.checkpoint_path = checkpoint_path
        self.config = config
        self.optim_config = optim_config
        self.max_epochs = max_epochs
        self.loss = loss
        self.metrics = metrics
        self.logged: dict[str, tp.Any] = {}

    def log(self, name: str, value: tp.Any, **kwargs: tp.Any) -> None:
        self.logged[name] = value

    def log_dict(self, values: dict[str, tp.Any], **kwargs: tp.Any) -> None:
        self.logged.update(values)

    def forward(self, batch: SegmentData) -> torch.Tensor:
        return self.model(batch)

    def _loss(self, y_pred: torch.Tensor, y_true: torch.Tensor) -> torch.Tensor:
        if hasattr(self.loss, "forward_bvt"):
            return self.loss.forward_bvt(y_pred, y_true)
        # foreign loss module: materialise the reference's "b d t -> (b t) d" flatten (pl_module.py:54-55)
        V = y_pred.shape[1]
        return self.loss(y_pred.permute(0, 2, 1).reshape(-1, V), y_true.permute(0, 2, 1).reshape(-1, V))

    def _run_step(self, batch: SegmentData, batch_idx: int, step_name: str):
        y_true = batch.data["fmri"]  # B, D, T
        y_pred = self.forward(batch)  # B, D, T
        subject_id = batch.data["subject_id"]
        loss = self._loss(y_pred, y_true.to(y_pred.dtype))

        if hasattr(self.model, "compute_contrastive_loss"):
            contrastive_losses = self.model.compute_contrastive_loss(batch)
            if contrastive_losses:
                weight = getattr(self.model.config, "contrastive_weight", 0.0)
                total = 0.0
                for name, c_loss in contrastive_losses.items():
                    self.log(f"{step_name}/contrastive/{name}", c_loss, batch_size=y_pred.shape[0])
                    total = total + c_loss
                loss = loss + weight * (total / max(1, len(contrastive_losses)))
        self.log(f"{step_name}/loss", loss, batch_size=y_pred.shape[0])

        for metric_name, metric in self.metrics.items():
            if not metric_name.startswith(step_name):
                continue
            if "grouped" in metric.__class__.__name__.lower():
                # per-row groups of the flattened view == the sample's subject id repeated T' times (pl_module.py:52)
                metric.update(y_pred, y_true, groups=subject_id)
            else:
                if "retrieval" in metric_name:
                    metric.update(y_pred.mean(dim=-1), y_true.mean(dim=-1))
                else:
                    metric.update(y_pred, y_true)
                self.log(metric_name, metric)
        return loss, y_pred.detach().cpu(), y_true.detach().cpu()

    def on_val_or_test_epoch_end(self, step_name: str) -> None:
        for metric_name, metric in self.metrics.items():
            if metric_name.startswith(step_name) and "grouped" in metric.__class__.__name__.lower():
                self.log_dict({metric_name + "/" + k: v for k, v in metric.compute().items()})

    def on_validation_epoch_end(self) -> None:
        self.on_val_or_test_epoch_end("val")

    def on_test_epoch_end(self) -> None:
        self.on_val_or_test_epoch_end("test")

    def training_step(self, batch: SegmentData, batch_idx: int):
        """pl_module.py:126-128: returns the loss tensor; `loss.backward()` then runs the HIP backward kernels through
        the autograd functions of modeling_utils/autograd.py (MSE / Pearson loss + optional InfoNCE alignment)."""
        loss, _, _ = self._run_step(batch, batch_idx, step_name="train")
        return loss

    def validation_step(self, batch: SegmentData, batch_idx: int):
        _, y_pred, y_true = self._run_step(batch, batch_idx, step_name="val")
        return y_pred, y_true

    def test_step(self, batch: SegmentData, batch_idx: int):
        _, y_pred, y_true = self._run_step(batch, batch_idx, step_name="test")
        return y_pred, y_true
