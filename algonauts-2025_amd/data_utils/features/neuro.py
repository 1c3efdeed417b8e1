"""fMRI target feature on the HBM feature store.

Mirror of the reference plugin `Fmri` (/root/reference/data_utils/data_utils/features/neuro.py:25-153): one recording per segment
(`events[:1]`, :81), z-scored per voxel over time (`nilearn.signal.clean(..., standardize="zscore_sample")`, :108-125), placed on
the timeline 4.47 s (3 TRs) before the event start (:141-153) and cut to the segment window on a 1/1.49 Hz grid.  The window
arithmetic is the host mirror of `TimedArray` (data_utils/base.py, pinned by G10 / G11: `fmri_seg*`), the cut itself one gather
launch; `__call__` returns f32 [V, T'] on the GPU.  `event.read()` (nibabel) stays host-side as in the reference; nilearn is not a
dependency: `zscore_sample` is restated ((x - mean) / std with ddof = 1 along time, constant voxels -> 0) -- parity unpinned
(nilearn absent offline; no reference fixture)."""

from __future__ import annotations

import typing as tp

import numpy as np

from .plugin import HbmFeaturePlugin


class Fmri(HbmFeaturePlugin):
    name: tp.Literal["Fmri"] = "Fmri"
    layers: list[float] = [1.0]                         # unused: a recording has no layer axis (kept off the cache uid like the others)
    _EVENT_TYPE: tp.ClassVar[str] = "Fmri"
    _KIND: tp.ClassVar[str] = "target"
    _FREQUENCY: tp.ClassVar[float] = 1 / 1.49
    _START_SHIFT: tp.ClassVar[float] = 4.47
    _FIRST_EVENT_ONLY: tp.ClassVar[bool] = True

    def _exclude_from_cache_uid(self) -> list[str]:
        return ["offset"]

    def _item_uid(self, event: tp.Any) -> str:
        return str(event.filepath)

    def _preprocess_event(self, event: tp.Any) -> np.ndarray:
        rec = event.read()
        data = np.asarray(rec.get_fdata() if hasattr(rec, "get_fdata") else rec, dtype=np.float64)   # [V..., T]
        flat = data.reshape(-1, data.shape[-1])
        mean = flat.mean(axis=1, keepdims=True)
        std = flat.std(axis=1, ddof=1, keepdims=True)
        z = np.where(std > np.finfo(np.float64).eps, (flat - mean) / np.where(std == 0, 1.0, std), 0.0)
        return z.reshape(data.shape).astype(np.float32)

    def _compute(self, events: list[tp.Any]) -> tp.Iterator[np.ndarray]:
        for event in events:
            yield self._preprocess_event(event)
