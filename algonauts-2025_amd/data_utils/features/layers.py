"""Layer selection / grouping of cached extractor hidden states.

Host-side restatement of `_aggregate_layers`, which the reference repeats in each
extractor (/root/reference/data_utils/data_utils/features/text.py:129-149,
audio.py:123-143, video.py:147-167): `layers` are fractions of the depth; with
"group_mean" consecutive selected indices delimit groups that are averaged.
Pure index arithmetic on a small [n_states, D(, T)] array -- stays on the host.
"""

from __future__ import annotations

import typing as tp

import numpy as np


def layer_indices(n_states: int, layers: tp.Sequence[float]) -> list[int]:
    return sorted({int(f * (n_states - 1)) for f in layers})


def aggregate_layers(latents: np.ndarray, layers: tp.Sequence[float], layer_aggregation: str | None) -> np.ndarray:
    idx = layer_indices(latents.shape[0], layers)
    if len(idx) == 1:
        only = latents[idx[0]]
        return only[None, :] if layer_aggregation is None else only
    if layer_aggregation is None:
        return latents[idx]
    if layer_aggregation != "group_mean":
        raise ValueError(f"Unknown layer aggregation: {layer_aggregation}")
    bounds = idx[:-1] + [idx[-1] + 1]
    return np.stack([latents[lo:hi].mean(0) for lo, hi in zip(bounds[:-1], bounds[1:])])
