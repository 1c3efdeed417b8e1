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
    """Same constructor arguments and hooks as the reference module; every tensor op below runs through the HIP kernels."""

    STEP_NAMES = ("train", "val", "test")

    def __init__(self, model: nn.Module, loss: nn.Module, optim_config: tp.Any, metrics: dict[str, tp.Any],
                 max_epochs: int = 100, checkpoint_path: Path | None = None, config: dict[str, tp.Any] | None = None) -> None:
        super().__init__()
        self.model, self.loss, self.metrics = model, loss, metrics
        self.optim_config, self.max_epochs = optim_config, max_epochs
        self.checkpoint_path, self.config = checkpoint_path, config
        self.logged: dict[str, tp.Any] = {}   # what a Lightning logger would have received

    # -- logging shims (Lightning provides these on a LightningModule) ---------------------------------
    def log(self, name: str, value: tp.Any, **kwargs: tp.Any) -> None:
        self.logged[name] = value

    def log_dict(self, values: dict[str, tp.Any], **kwargs: tp.Any) -> None:
        self.logged.update(values)

    # -- the step ---------------------------------------------------------------------------------------------
    def forward(self, batch: SegmentData) -> torch.Tensor:
        return self.model(batch)

    def _primary_loss(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """pl_module.py:54-56 without the two transposing copies when the loss understands [B, V, T'] directly."""
        fused = getattr(self.loss, "forward_bvt", None)
        if fused is not None:
            return fused(pred, target)
        n_out = pred.shape[1]   # foreign loss module: materialise the "b d t -> (b t) d" flatten it expects
        return self.loss(pred.permute(0, 2, 1).reshape(-1, n_out), target.permute(0, 2, 1).reshape(-1, n_out))

    def _alignment_term(self, batch: SegmentData, stage: str, n: int) -> torch.Tensor | float:
        """pl_module.py:58-67: weight * mean of the per-modality InfoNCE terms (0 when the branch is off)."""
        compute = getattr(self.model, "compute_contrastive_loss", None)
        terms = compute(batch) if compute is not None else None
        if not terms:
            return 0.0
        for modality, value in terms.items():
            self.log(f"{stage}/contrastive/{modality}", value, batch_size=n)
        weight = getattr(self.model.config, "contrastive_weight", 0.0)
        return weight * (sum(terms.values()) / max(1, len(terms)))

    def _feed_metrics(self, stage: str, pred: torch.Tensor, target: torch.Tensor, subjects: torch.Tensor) -> None:
        """pl_module.py:70-105.  Grouped metrics take the sample's subject id (the reference repeats it T' times for the flattened
        rows, :52); retrieval metrics see time-averaged inputs."""
        for key, metric in self.metrics.items():
            if not key.startswith(stage):
                continue
            if _is_grouped(metric):
                metric.update(pred, target, groups=subjects)
                continue
            if "retrieval" in key:
                metric.update(pred.mean(dim=-1), target.mean(dim=-1))
            else:
                metric.update(pred, target)
            self.log(key, metric)

    def _run_step(self, batch: SegmentData, batch_idx: int, step_name: str, outputs: bool = True):
        """pl_module.py:107-124.  `outputs=False` (training_step, which only returns the loss) skips the host copies of the prediction and
        the target: 130 MB over PCIe and a device synchronisation per step at B = 16 that nothing reads."""
        target = batch.data["fmri"]                 # [B, V, T']
        pred = self.forward(batch)                  # [B, V, T']
        n = pred.shape[0]
        loss = self._primary_loss(pred, target.to(pred.dtype)) + self._alignment_term(batch, step_name, n)
        self.log(f"{step_name}/loss", loss, batch_size=n)
        self._feed_metrics(step_name, pred, target, batch.data["subject_id"])
        if not outputs:
            return loss, None, None
        return loss, pred.detach().cpu(), target.detach().cpu()

    def training_step(self, batch: SegmentData, batch_idx: int):
        """pl_module.py:126-128: returns the loss tensor; `loss.backward()` then runs the HIP backward kernels through
        the autograd functions of modeling_utils/autograd.py (MSE / Pearson loss + optional InfoNCE alignment)."""
        return self._run_step(batch, batch_idx, step_name="train", outputs=False)[0]

    def validation_step(self, batch: SegmentData, batch_idx: int):
        return self._run_step(batch, batch_idx, step_name="val")[1:]

    def test_step(self, batch: SegmentData, batch_idx: int):
        return self._run_step(batch, batch_idx, step_name="test")[1:]

    # -- epoch ends: grouped metrics report one value per subject ---------------------------------------------------
    def on_val_or_test_epoch_end(self, step_name: str) -> None:
        for key, metric in self.metrics.items():
            if key.startswith(step_name) and _is_grouped(metric):
                self.log_dict({f"{key}/{group}": value for group, value in metric.compute().items()})

    def on_validation_epoch_end(self) -> None:
        self.on_val_or_test_epoch_end("val")

    def on_test_epoch_end(self) -> None:
        self.on_val_or_test_epoch_end("test")

    def configure_optimizers(self, total_steps: int | None = None) -> tp.Any:
        """pl_module.py:138-144: the optimiser over the trainable parameters.  An `optim_config` with a `build(params, total_steps=...)`
        method (the reference's) is used as is; otherwise Adam(lr=1e-4) as in grids/defaults.py:126-133, as one HIP launch per
        step (modeling_utils.optim.HipAdam)."""
        params = [p for p in self.parameters() if p.requires_grad]
        build = getattr(self.optim_config, "build", None)
        if build is not None:
            if total_steps is None:
                # Lightning calls configure_optimizers() with no argument; the reference then sizes OneCycleLR with
                # self.trainer.estimated_stepping_batches (pl_module.py:139-143)
                trainer = getattr(self, "trainer", None)
                total_steps = getattr(trainer, "estimated_stepping_batches", None)
            return build(params, total_steps=total_steps)
        from modeling_utils.optim import HipAdam

        return HipAdam(params, lr=1e-4, weight_decay=0.0)


def _is_grouped(metric: tp.Any) -> bool:
    return "grouped" in type(metric).__name__.lower()
