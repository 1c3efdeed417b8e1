"""Ensemble averaging of submissions (/root/reference/algonauts2025/grids/average_submissions.py:18-131).

Same functions, arguments and file conventions; the array work runs on the GPU:
  * `np.mean(preds, axis=0)`                 -> tribe_group_mean_fwd (one group over the N submissions)
  * `np.sum(preds * weights, axis=0)`        -> tribe_weighted_sum_fwd (per-voxel f32 weights or per-submission f64 weights)
  * `np.corrcoef(preds)` for the diverse subset -> tribe_corr_matrix_fwd (f64)
with the summation order and rounding of numpy, so the averaged arrays are bit-identical to the reference's for a given
order of submissions (tests/test_gpu_ensemble.py).  The softmax of N scores and the greedy subset choice are a few
hundred flops and stay on the host.  Reference quirks kept: the per-voxel softmax runs over dim=1 (voxels, :98), and
only the first subject's predictions feed the diversity estimate (:46).

`submissions` may be passed in memory (`average_predictions`); `average_submissions(grid_path, ...)` keeps the reference's
folder protocol (`<run>/submission.zip`, `metrics.csv`, `pearson.npy`).  submission files are pickled dicts (the
competition's format): they are only read with `trust_pickle=True`.
"""

from __future__ import annotations

import os
import typing as tp
import zipfile
from collections import defaultdict
from pathlib import Path

import numpy as np
import torch

from tribe_hip import ops


def select_diverse_subset(C: np.ndarray, k: int) -> list[int]:
    """Greedy choice of k mutually least-correlated rows (:18-35): start from the smallest total |corr|, then add the
    candidate (ascending index, first minimum wins) with the smallest summed |corr| to the chosen ones."""
    A = np.abs(np.asarray(C))
    selected = [int(np.argmin(A.sum(axis=0)))]
    while len(selected) < k:
        best, best_total = -1, None
        for c in sorted(set(range(A.shape[0])) - set(selected)):
            total = sum(A[c, s] for s in selected)
            if best_total is None or total < best_total:
                best, best_total = c, total
        selected.append(best)
    return selected


def _stack(predictions: list[dict], sub: str, chunk: str, device: torch.device) -> torch.Tensor:
    return torch.stack([torch.as_tensor(np.asarray(p[sub][chunk], dtype=np.float32)) for p in predictions]).to(device)


def get_k_most_diverse_indices(predictions: list[dict], k: int, device: str | torch.device = "cuda") -> np.ndarray:
    """:37-53 -- correlation between submissions over all chunks of the FIRST subject."""
    device = torch.device(device)
    sub = next(iter(predictions[0].keys()))
    preds = torch.cat([_stack(predictions, sub, chunk, device) for chunk in predictions[0][sub].keys()], dim=1)
    preds = preds.reshape(preds.shape[0], -1).contiguous()
    assert preds.shape[0] == len(predictions)
    corr = ops.corr_matrix(preds).cpu().numpy()
    return np.array(select_diverse_subset(corr, k))


def ensemble_weights(scores: tp.Sequence[float] | None, pearsons: tp.Sequence[np.ndarray] | None, per_voxel_weights: bool,
                     temperature: float) -> np.ndarray:
    """:95-104.  per-voxel: softmax over dim=1 of pearsons / temperature, f32 [N, 1, V]; else softmax of the scores, f64 [N, 1, 1]."""
    if per_voxel_weights:
        p = torch.Tensor(np.asarray(pearsons)) / temperature
        return p.softmax(dim=1).unsqueeze(1).numpy()
    s = np.array(scores, dtype=np.float64)
    w = np.exp(s / temperature) / np.sum(np.exp(s / temperature))
    return w[:, None, None]


def average_predictions(predictions: list[dict], weights: np.ndarray | None = None, weigh_by_score: bool = False,
                        device: str | torch.device = "cuda") -> dict[str, dict[str, np.ndarray]]:
    """:107-118 for submissions already in memory."""
    device = torch.device(device)
    n = len(predictions)
    lo = torch.zeros(1, dtype=torch.int32, device=device)
    hi = torch.full((1,), n, dtype=torch.int32, device=device)
    w_col = w_set = None
    if weigh_by_score:
        if weights is None:
            raise ValueError("weigh_by_score needs weights")
        if weights.shape[-1] > 1:
            w_col = torch.from_numpy(np.ascontiguousarray(weights.reshape(n, -1), dtype=np.float32)).to(device)
        else:
            w_set = torch.from_numpy(np.ascontiguousarray(weights.reshape(n), dtype=np.float64)).to(device)
    out: dict[str, dict[str, np.ndarray]] = defaultdict(dict)
    for sub in predictions[0].keys():
        for chunk in predictions[0][sub].keys():
            preds = _stack(predictions, sub, chunk, device)              # [N, T, V]
            if weigh_by_score:
                avg = ops.weighted_sum(preds, w_column=w_col, w_set=w_set)
            else:
                avg = ops.group_mean(preds[None], lo, hi)[0, 0]
            out[sub][chunk] = avg.cpu().numpy()
    return out


def average_submissions(grid_path: Path, weigh_by_score: bool = False, per_voxel_weights: bool = False, temperature: float = 1.0,
                        max_runs: int | None = None, k_most_diverse: int | None = None, trust_pickle: bool = False,
                        device: str | torch.device = "cuda") -> dict[str, dict[str, np.ndarray]]:
    """:55-131.  Runs are taken in sorted folder order (the reference's order is whatever `os.listdir` and its thread
    pool produce; the averaged values depend on it only in the last bits of the f32 sums)."""
    import pandas as pd

    grid_path = Path(grid_path)
    if not trust_pickle:
        raise ValueError("submission.zip files hold pickled dicts; pass trust_pickle=True to read runs you produced yourself")
    paths = []
    for folder in sorted(os.listdir(grid_path)):
        if max_runs is not None and len(paths) == max_runs:
            break
        run = grid_path / folder
        if run.is_dir():
            if (run / "submission.zip").exists():
                paths.append(run / "submission.zip")
            if (run / "submission.npy").exists():
                os.remove(run / "submission.npy")
    predictions, scores, pearsons = [], [], []
    for path in paths:
        try:
            submission = np.load(path, allow_pickle=True)["submission"].item()
        except Exception:
            print(f"Error loading submission from {path}")
            continue
        predictions.append(submission)
        scores.append(pd.read_csv(path.with_name("metrics.csv")))
        pearsons.append(np.load(path.with_name("pearson.npy")) if path.with_name("pearson.npy").exists() else None)
    if k_most_diverse is not None:
        indices = get_k_most_diverse_indices(predictions, k_most_diverse, device=device)
        predictions, scores = [predictions[i] for i in indices], [scores[i] for i in indices]
    weights = ensemble_weights([s["val/pearson"].item() for s in scores] if not per_voxel_weights else None, pearsons, per_voxel_weights,
                               temperature)
    averaged = average_predictions(predictions, weights, weigh_by_score, device=device)
    submission_path = grid_path / "submission.npy"
    np.save(submission_path, dict(averaged))
    with zipfile.ZipFile(submission_path.with_suffix(".zip"), "w") as zipf:
        zipf.write(submission_path, arcname=submission_path.name)
    print(f"Saved average submission to {submission_path.with_suffix('.zip')}")
    return averaged
