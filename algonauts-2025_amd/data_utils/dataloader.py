"""Batch container handed to `FmriEncoder.forward`.

Mirrors the reference boundary type `SegmentData`
(/root/reference/data_utils/data_utils/dataloader.py:27-53): same two fields, same
validation errors, same `.to(device)` and the same refusal of dict-style access.
`data` maps 'text' | 'audio' | 'video' -> [B, L, D, T] (or [B, D, T]),
'fmri' -> [B, V, T'], 'subject_id' -> int64 [B, 1].
"""

from __future__ import annotations

import dataclasses
import typing as tp

import torch


@dataclasses.dataclass
class SegmentData:
    data: tp.Dict[str, torch.Tensor]
    segments: tp.List[tp.Any]

    def __post_init__(self) -> None:
        if not isinstance(self.data, dict):
            raise TypeError(f"'features' need to be a dict, got: {type(self.data)}")
        if not self.data:
            raise ValueError(f"No data in {self}")
        if not isinstance(self.segments, list):
            raise TypeError(f"'segments' needs to be a list, got {self.segments}")
        batch_size = next(iter(self.data.values())).shape[0]
        if len(self.segments) != batch_size:
            raise RuntimeError(f"Incoherent batch size {batch_size} for {len(self.segments)} segments in {self}")

    def to(self, device: str | torch.device) -> "SegmentData":
        return SegmentData(data={k: v.to(device) for k, v in self.data.items()}, segments=self.segments)

    def __getitem__(self, key: str) -> None:
        raise RuntimeError("New SegmentData batch is not a dict, use batch.data instead")
