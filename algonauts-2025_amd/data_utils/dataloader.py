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


# ---------------------------------------------------------------------------------------------------------------------
# SegmentDataset: the caller of the feature plugins (dataloader.py:56-187 of the reference)
# ---------------------------------------------------------------------------------------------------------------------
import collections.abc  # noqa: E402
import warnings  # noqa: E402


def validate_features(features: tp.Mapping[str, tp.Any]) -> tp.Mapping[str, tp.Any]:
    if not features:
        return {}
    if not isinstance(features, collections.abc.Mapping):
        raise ValueError(f"Only dict of features are supported, got {type(features)}")
    return features


def get_pad_lengths(feats: tp.Mapping[str, tp.Any], pad_duration: float | None) -> tp.Dict[str, int]:
    """dataloader.py:69-87: `pad_duration` seconds on each feature's own grid (features without a sampling rate -- the subject
    label, frequency 0 -- are left alone)."""
    from .base import Frequency

    pad_lengths: tp.Dict[str, int] = {}
    if pad_duration is None:
        return pad_lengths
    for name, f in feats.items():
        freq = getattr(f, "frequency", None)
        if freq:
            pad_lengths[name] = Frequency(freq).to_ind(pad_duration)
    return pad_lengths


def _pad_to(tensor: torch.Tensor, pad_len: int | None) -> torch.Tensor:
    if pad_len is None:
        return tensor
    if pad_len < tensor.shape[-1]:
        warnings.warn("Pad duration is shorter than segment duration, cropping.", UserWarning)
        return tensor[..., :pad_len]   # the time axis (the reference's `tensor[:, :pad_len]` would crop D of a [L, D, T] tensor: a slip)
    return torch.nn.functional.pad(tensor, (0, pad_len - tensor.shape[-1]))


def _apply_feature(segment: tp.Any, feature: tp.Any) -> torch.Tensor:
    return feature(segment.ns_events, start=segment.start, duration=segment.duration, trigger=getattr(segment, "_trigger", None))


class SegmentDataset(torch.utils.data.Dataset):
    """Segments x features -> `SegmentData` items, as the reference's `SegmentDataset` (dataloader.py:111-187): `__getitem__` calls
    every feature on the segment's events, pads / crops to `pad_duration`, adds the batch axis; `collate_fn` concatenates.

    With this build's plugins the item tensors already live on the GPU (the features cut them out of HBM-resident states), so
    DataLoader worker processes are neither needed nor possible: `build_dataloader` runs in the calling process
    (`num_workers` is forced to 0), and when every feature is bound to ONE `HbmFeatureStore`, `gpu_batches` skips the
    per-segment tensors altogether (one gather launch per feature and batch, `GpuSegmentLoader`)."""

    def __init__(self, features: tp.Mapping[str, tp.Any], segments: tp.Sequence[tp.Any], pad_duration: float | None = None) -> None:
        self.features = validate_features(features)
        self.segments = segments
        self.pad_duration = pad_duration
        self._pad_lengths = get_pad_lengths(self.features, pad_duration)

    def collate_fn(self, batches: tp.List[SegmentData]) -> SegmentData:
        if not batches:
            return _empty_batch()
        if len(batches) == 1:
            return batches[0]
        if not batches[0].data:
            raise ValueError(f"No feature in first batch: {batches[0]}")
        features = {}
        for name in batches[0].data:
            data = [b.data[name] for b in batches]
            try:
                features[name] = torch.cat(data, axis=0)
            except Exception:
                raise RuntimeError(f"Failed to collate data with shapes {[d.shape for d in data]}\\n"
                                   "Do you need specifying padding in SegmentDataset?")
        return SegmentData(data=features, segments=[s for b in batches for s in b.segments])

    def __len__(self) -> int:
        return len(self.segments)

    def __getitem__(self, idx: int) -> SegmentData:
        seg = self.segments[idx]
        out: tp.Dict[str, torch.Tensor] = {}
        for name, feat in self.features.items():
            data = _pad_to(_apply_feature(seg, feat), self._pad_lengths.get(name, None))
            out[name] = data[None, ...]
        return SegmentData(data=out, segments=[seg])

    def build_dataloader(self, **kwargs: tp.Any) -> torch.utils.data.DataLoader:
        kwargs["num_workers"] = 0          # item tensors are CUDA tensors cut from HBM-resident states: no worker processes
        kwargs.pop("prefetch_factor", None)
        kwargs.pop("persistent_workers", None)
        return torch.utils.data.DataLoader(self, collate_fn=self.collate_fn, **kwargs)

    def as_one_batch(self, num_workers: int = 0) -> SegmentData:
        loader = self.build_dataloader(batch_size=max(1, len(self)), shuffle=False)
        return self.collate_fn(list(loader))

    def gpu_batches(self, batch_size: int) -> tp.Iterator[SegmentData]:
        """The fast path: every feature a plugin bound to the same HbmFeatureStore (plus optionally a SubjectEncoder under
        'subject_id') -> batches with the projector operands written directly (bf16 `PackedFeature`s)."""
        from .gpu_loader import GpuSegmentLoader

        stores = {id(getattr(f, "_store", None)): getattr(f, "_store", None) for n, f in self.features.items() if n != "subject_id"}
        if len(stores) != 1 or None in stores.values():
            raise ValueError("gpu_batches needs every feature bound to one HbmFeatureStore (feature.bind(store, name=key))")
        store = next(iter(stores.values()))
        subj = self.features.get("subject_id")
        loader = GpuSegmentLoader(store, pad_duration=self.pad_duration, subject_index=getattr(subj, "subject_index", None) or None)
        names = [n for n in self.features if n != "subject_id"]
        for i in range(0, len(self.segments), batch_size):
            yield loader.batch(list(self.segments[i:i + batch_size]), names=names)


def _empty_batch() -> SegmentData:
    batch = SegmentData.__new__(SegmentData)     # the reference returns an empty container here; its validator would refuse it
    batch.data, batch.segments = {}, []
    return batch
