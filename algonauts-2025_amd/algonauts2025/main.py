"""Entry points of the reference driver that touch the hot path.

The reference's `Experiment` / `Data` (/root/reference/algonauts2025/main.py:63-511) are a
control plane (exca job infra, Lightning Trainer, WandB, dataset building) and are out of scope
(SURVEY.md section 8); what is kept is the evaluation routine that DEFINES the parity metric:
`compute_multidim_pearson` (main.py:459-477) -- per-parcel Pearson r between concatenated
predictions and targets over a loader -- computed here from GPU-side f64 sufficient statistics
instead of 1000 scipy calls on host copies.
"""

from __future__ import annotations

import typing as tp

import numpy as np
import torch

from modeling_utils.metrics.base import _PearsonState


def compute_multidim_pearson(brain_module: tp.Any, loader: tp.Iterable, device: str | torch.device = "cuda",
                             sync_distributed: bool = True) -> np.ndarray:
    """float32 [n_outputs]; rows of the '(b t) d' flatten over all batches of `loader` (main.py:459-477)."""
    state: _PearsonState | None = None
    brain_module.eval()
    brain_module.to(device)
    with torch.inference_mode():
        for batch in loader:
            batch = batch.to(device)
            y_true = batch.data["fmri"].squeeze(-1)
            y_pred = brain_module(batch)
            if state is None:
                state = _PearsonState(y_pred.shape[1])
            state.update_bvt(y_pred, y_true.to(torch.float32))
    if state is None:
        raise ValueError("empty loader")
    if sync_distributed:
        state.sync()
    return state.per_output()[0].cpu().numpy().astype(np.float32)


class Experiment:
    """Placeholder for the reference's pydantic `Experiment` (main.py:206-511): orchestration only, out of scope."""

    def __init__(self, *args: tp.Any, **kwargs: tp.Any) -> None:
        raise NotImplementedError("Experiment (exca / Lightning / WandB control plane) is outside the MI355X hot-path build; "
                                  "use FmriEncoderConfig.build(...) + BrainModule + compute_multidim_pearson")
