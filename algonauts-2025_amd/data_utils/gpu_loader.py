"""Segment batches straight from HBM-resident extractor outputs.

What the reference does per segment and per feature on the host (numpy, inside DataLoader workers):
    feature(events, start, duration)           text.py:85-124, audio.py:78-120, neuro.py:60-106
      -> cached states of every event           exca MapInfra caches, [n_states, D(, T_event)]
      -> TimedArray.overlap(start, duration)    base.py:198-211
      -> _aggregate_layers                      text.py:129-149 (group_mean over layer groups)
      -> out += piece                           base.py:130-162
    collate -> fp32 [B, L, D, T] -> H2D copy -> "b (l d) t -> b t (l d)" transpose on the device (model.py:146-155)
is split here into
    * once per event (`HbmFeatureStore.put`): upload the states, aggregate layers on the GPU (tribe_group_mean_fwd),
      keep f32 [L*D, T_event] (sampled features) / one f32 [n_words, L*D] table (word features) in HBM -- the whole
      Algonauts feature cache is a few tens of GB, 288 GB of HBM3E hold it with room to spare;
    * once per segment (`GpuSegmentLoader.plan`): the index decisions of TimedArray (data_utils/base.py of this build,
      vectorised; pinned by tests/golden/g10, g11), cached per segment object;
    * once per batch and modality: ONE launch (tribe_segment_gather_fwd / tribe_word_bag_fwd) that writes the bf16
      [B*T, C_pad] rows the projector GEMM reads -- no fp32 batch, no H2D copy, no transpose pass.
The result is bit-identical to bf16(reference fp32 tensor) (tests/test_gpu_loader.py).
"""

from __future__ import annotations

import dataclasses
import typing as tp

import numpy as np
import torch

from tribe_hip import _lib, ops

from .base import overlap_window
from .dataloader import SegmentData
from .features.layers import layer_indices


@dataclasses.dataclass
class FeatureSpec:
    """The part of a feature plugin's config that shapes its tensor (text.py:42-47, audio.py:27-40, neuro.py:25-30)."""

    name: str                                  # key in SegmentData.data
    kind: tp.Literal["sampled", "words", "target"]
    event_type: str                            # "Word" | "Sound" | "Video" | "Fmri"
    frequency: float = 2.0                     # output grid in Hz (2.0 for the stimulus features, 1/1.49 for fMRI)
    layers: tp.Sequence[float] = (0.5, 0.75, 1.0)
    layer_aggregation: str | None = "group_mean"
    pass_event_duration: bool = False          # video.py:176-182 hands event.duration to the array (validated), audio.py:241 not
    start_shift: float = 0.0                   # neuro.py:150: the recording is read 4.47 s early
    first_event_only: bool = False             # neuro.py:81: events[:1]

    @staticmethod
    def defaults() -> list["FeatureSpec"]:
        return [
            FeatureSpec("text", "words", "Word"),
            FeatureSpec("audio", "sampled", "Sound"),
            FeatureSpec("video", "sampled", "Video", pass_event_duration=True),
            FeatureSpec("fmri", "target", "Fmri", frequency=1 / 1.49, start_shift=4.47, first_event_only=True),
        ]


class PackedFeature:
    """A feature batch already in the projector's operand layout: bf16 [B*T, C_pad] rows (time-major, channels
    contiguous, zero padded), plus the logical [B, L, D, T] shape the reference tensor would have had."""

    def __init__(self, packed: torch.Tensor, B: int, L: int, D: int, T: int) -> None:
        if packed.dtype != torch.bfloat16 or packed.ndim != 2 or packed.shape[0] != B * T or packed.shape[1] < L * D:
            raise ValueError(f"PackedFeature: {tuple(packed.shape)} {packed.dtype} does not hold [B={B}, L={L}, D={D}, T={T}]")
        self.packed, self.B, self.L, self.D, self.T = packed, B, L, D, T

    @property
    def shape(self) -> torch.Size:
        return torch.Size((self.B, self.L, self.D, self.T))

    ndim = 4
    dtype = torch.bfloat16

    @property
    def device(self) -> torch.device:
        return self.packed.device

    def to(self, device: tp.Any) -> "PackedFeature":
        return PackedFeature(self.packed.to(device), self.B, self.L, self.D, self.T)

    def unpack(self) -> torch.Tensor:
        """The reference-layout tensor f32 [B, L, D, T] (values rounded to bf16)."""
        C = self.L * self.D
        return self.packed.view(self.B, self.T, -1)[:, :, :C].permute(0, 2, 1).reshape(self.B, self.L, self.D, self.T).float()


def layer_groups(n_states: int, layers: tp.Sequence[float], layer_aggregation: str | None) -> tuple[list[int], list[int]]:
    """[lo, hi) layer ranges whose means `_aggregate_layers` returns (text.py:129-149); ranges of one = index select."""
    idx = layer_indices(n_states, layers)
    if len(idx) == 1 or layer_aggregation is None:
        return idx, [i + 1 for i in idx]
    if layer_aggregation != "group_mean":
        raise ValueError(f"Unknown layer aggregation: {layer_aggregation}")
    bounds = idx[:-1] + [idx[-1] + 1]
    return bounds[:-1], bounds[1:]


def default_event_key(event: tp.Any) -> tp.Hashable:
    """Cache key of an event: media events by file (+ offset), as the reference's `item_uid`s do (audio.py:253,
    neuro.py:132); words by identity of their (timeline, start, text)."""
    path = getattr(event, "filepath", "")
    if path:
        return (str(path), float(getattr(event, "offset", 0.0) or 0.0))
    return (getattr(event, "timeline", ""), float(event.start), float(event.duration), getattr(event, "text", ""))


@dataclasses.dataclass
class _Resident:
    array: torch.Tensor      # f32 [C, T_event] in HBM
    n: int                   # T_event


class HbmFeatureStore:
    """Layer-aggregated extractor outputs, resident in HBM, addressed by (feature name, event key)."""

    def __init__(self, specs: tp.Sequence[FeatureSpec], device: str | torch.device = "cuda",
                 event_key: tp.Callable[[tp.Any], tp.Hashable] = default_event_key) -> None:
        self.specs = {s.name: s for s in specs}
        self.device = torch.device(device)
        self.event_key = event_key
        self._sampled: dict[tuple[str, tp.Hashable], _Resident] = {}
        self._word_rows: dict[str, dict[tp.Hashable, int]] = {}
        self._word_chunks: dict[str, list[torch.Tensor]] = {}
        self._word_table: dict[str, torch.Tensor] = {}
        self.channels: dict[str, tuple[int, int]] = {}   # name -> (L, D)

    # -- filling ------------------------------------------------------------------------------------
    def _aggregate(self, spec: FeatureSpec, states: torch.Tensor) -> torch.Tensor:
        """states f32 [batch, n_states, plane...] on the device -> [batch, L, plane...]."""
        lo, hi = layer_groups(states.shape[1], spec.layers, spec.layer_aggregation)
        lo_t = torch.tensor(lo, dtype=torch.int32, device=self.device)
        hi_t = torch.tensor(hi, dtype=torch.int32, device=self.device)
        return ops.group_mean(states, lo_t, hi_t)

    def _note_channels(self, name: str, L: int, D: int) -> None:
        if self.channels.setdefault(name, (L, D)) != (L, D):
            raise ValueError(f"feature {name!r}: arrays of [L={L}, D={D}] after {self.channels[name]}")

    def _upload(self, a: tp.Any) -> torch.Tensor:
        t = torch.as_tensor(a)
        return t.to(self.device, dtype=torch.float32, non_blocking=True).contiguous()   # fp64 caches (video.py:230) are narrowed here

    def put(self, name: str, event: tp.Any, states: tp.Any) -> None:
        """One event's cached extractor output: [n_states, D, T_event] (sampled), [n_states, D] (word) or [V, T_event] (target)."""
        spec = self.specs[name]
        key = self.event_key(event)
        x = self._upload(states)
        if spec.kind == "target":
            if x.ndim != 2:
                raise ValueError(f"target {name!r}: expected [V, T], got {tuple(x.shape)}")
            self._note_channels(name, 1, x.shape[0])
            self._sampled[(name, key)] = _Resident(x, x.shape[1])
        elif spec.kind == "sampled":
            if x.ndim == 2:
                x = x[None]
            if x.ndim != 3:
                raise ValueError(f"feature {name!r}: expected [n_states, D, T], got {tuple(x.shape)}")
            agg = self._aggregate(spec, x[None])[0]            # [L, D, T]
            self._note_channels(name, agg.shape[0], agg.shape[1])
            self._sampled[(name, key)] = _Resident(agg.reshape(-1, agg.shape[-1]), agg.shape[-1])
        else:
            self.put_words(name, [event], x[None])

    def put_words(self, name: str, events: tp.Sequence[tp.Any], states: tp.Any) -> None:
        """Word latents [n_words, n_states, D] of `events` (same order)."""
        spec = self.specs[name]
        x = self._upload(states)
        if x.ndim != 3 or x.shape[0] != len(events):
            raise ValueError(f"feature {name!r}: expected [{len(events)}, n_states, D], got {tuple(x.shape)}")
        agg = self._aggregate(spec, x)                           # [n_words, L, D]
        self._note_channels(name, agg.shape[1], agg.shape[2])
        rows = self._word_rows.setdefault(name, {})
        base = len(rows)
        fresh = []
        for i, e in enumerate(events):
            k = self.event_key(e)
            if k not in rows:
                rows[k] = base + len(fresh)
                fresh.append(i)
        if fresh:
            sel = agg if len(fresh) == len(events) else agg[torch.tensor(fresh, device=self.device)]
            self._word_chunks.setdefault(name, []).append(sel.reshape(len(fresh), -1))
            self._word_table.pop(name, None)

    # -- lookup -------------------------------------------------------------------------------------
    def resident(self, name: str, event: tp.Any) -> _Resident:
        try:
            return self._sampled[(name, self.event_key(event))]
        except KeyError:
            raise KeyError(f"feature {name!r}: no cached array for event {event!r} (call store.put first)") from None

    def word_row(self, name: str, event: tp.Any) -> int:
        try:
            return self._word_rows[name][self.event_key(event)]
        except KeyError:
            raise KeyError(f"feature {name!r}: no cached latent for word {event!r} (call store.put_words first)") from None

    def word_table(self, name: str) -> torch.Tensor:
        if name not in self._word_table:
            self._word_table[name] = torch.cat(self._word_chunks[name], dim=0).contiguous()
            self._word_chunks[name] = [self._word_table[name]]
        return self._word_table[name]

    def nbytes(self) -> int:
        n = sum(r.array.numel() * 4 for r in self._sampled.values())
        return n + sum(c.numel() * 4 for chunks in self._word_chunks.values() for c in chunks)


@dataclasses.dataclass
class _SegmentPlan:
    """Index decisions for one (segment, feature): either pieces of resident arrays or (step, word row) pairs."""

    n_out: int
    pieces: np.ndarray | None = None     # FEATURE_PIECE_DTYPE records
    steps: np.ndarray | None = None      # int32 destination step per (word, covered step) pair, in event order
    rows: np.ndarray | None = None       # int32 word-table row of the same pairs


def _events_of(segment: tp.Any, event_type: str) -> list[tp.Any]:
    return [e for e in segment.ns_events if getattr(e, "type", e.__class__.__name__) == event_type]


class GpuSegmentLoader:
    """Builds `SegmentData` batches on the GPU from an `HbmFeatureStore`.

    `pad_duration` mirrors SegmentDataset's padding / cropping to a common length (dataloader.py:69-98)."""

    def __init__(self, store: HbmFeatureStore, pad_duration: float | None = None, subject_index: dict[str, int] | None = None) -> None:
        self.store = store
        self.pad_duration = pad_duration
        self.subject_index = subject_index
        self._plans: dict[tuple[int, str], tuple[tp.Any, _SegmentPlan]] = {}   # the segment is kept so its id stays unique

    # -- per segment: TimedArray arithmetic on the host ---------------------------------------------
    def plan(self, segment: tp.Any, spec: FeatureSpec) -> _SegmentPlan:
        key = (id(segment), spec.name)
        hit = self._plans.get(key)
        if hit is not None and hit[0] is segment:
            return hit[1]
        freq = float(spec.frequency)
        n_out = max(1, int(round(segment.duration * freq)))                       # base.py:83-89
        out_dur = n_out / freq
        events = _events_of(segment, spec.event_type)
        if spec.first_event_only:
            events = events[:1]
        if spec.kind == "words":
            plan = self._plan_words(segment, spec, events, n_out, out_dur)
        else:
            plan = self._plan_sampled(segment, spec, events, n_out, out_dur)
        self._plans[key] = (segment, plan)
        return plan

    def clear_plans(self) -> None:
        self._plans.clear()

    def _plan_words(self, segment: tp.Any, spec: FeatureSpec, events: list[tp.Any], n_out: int, out_dur: float) -> _SegmentPlan:
        if not events:
            return _SegmentPlan(n_out, steps=np.zeros(0, np.int32), rows=np.zeros(0, np.int32))
        w_start = np.asarray([e.start for e in events], dtype=np.float64)
        w_dur = np.asarray([e.duration for e in events], dtype=np.float64)
        # `out += word` (base.py:144-151): the output array's slice for the word's window ...
        valid, _, _, first, count = overlap_window(spec.frequency, segment.start, n_out, out_dur, w_start, w_dur)
        # ... and the word's own (frequency 0) overlap with the output window must both exist (base.py:164-178)
        lo = np.maximum(segment.start, w_start)
        hi = np.minimum(segment.start + out_dur, w_start + w_dur)
        valid &= ~(hi < lo) & ~((hi == lo) & (w_dur != 0) & bool(out_dur))
        rows = np.asarray([self.store.word_row(spec.name, e) for e in events], dtype=np.int32)
        first, count, rows = first[valid], count[valid], rows[valid]
        reps = np.repeat(np.arange(len(first)), count)
        offs = np.arange(int(count.sum())) - np.repeat(np.cumsum(count) - count, count)
        return _SegmentPlan(n_out, steps=(first[reps] + offs).astype(np.int32), rows=rows[reps])

    def _plan_sampled(self, segment: tp.Any, spec: FeatureSpec, events: list[tp.Any], n_out: int, out_dur: float) -> _SegmentPlan:
        freq = float(spec.frequency)
        rec = np.zeros(len(events), dtype=_lib.FEATURE_PIECE_DTYPE)
        keep = 0
        for e in events:
            res = self.store.resident(spec.name, e)
            n = res.n
            ev_start = e.start - spec.start_shift
            if spec.pass_event_duration or spec.kind == "target":   # constructor validation (base.py:96-108)
                expected = max(1, int(round(e.duration * freq)))
                if abs(n - expected) > 2:
                    raise ValueError(f"Data has incorrect (last) dimension {(n,)} for duration {e.duration} and frequency {freq} "
                                     f"(expected {expected})")
            ev_dur = n / freq
            if spec.kind == "target":
                p_start, p_n, p_first = ev_start, n, 0                # neuro.py:141-153: the whole recording is the piece
            else:
                valid, s0, _, first, count = overlap_window(freq, ev_start, n, ev_dur, segment.start, segment.duration)
                if not bool(valid):                                   # audio.py:245-248: fall back to the first sample
                    valid, s0, _, first, count = overlap_window(freq, ev_start, n, ev_dur, ev_start, 0.0)
                p_start, p_n, p_first = float(s0), int(count), int(first)
            p_dur = p_n / freq
            a_valid, _, _, a_first, a_count = overlap_window(freq, segment.start, n_out, out_dur, p_start, p_dur)
            b_valid, _, _, b_first, b_count = overlap_window(freq, p_start, p_n, p_dur, segment.start, out_dur)
            if not (bool(a_valid) and bool(b_valid)):
                continue
            a_count, b_count = int(a_count), int(b_count)
            if a_count != b_count and b_count != 1:
                raise ValueError(f"operands could not be broadcast together with shapes ({a_count},) ({b_count},)")   # numpy's error in base.py:156
            rec[keep] = (res.array.data_ptr(), n, p_first + int(b_first), b_count, int(a_first), a_count)
            keep += 1
        return _SegmentPlan(n_out, pieces=rec[:keep])

    # -- per batch: one launch per feature ------------------------------------------------------------
    def _steps(self, spec: FeatureSpec, plans: list[_SegmentPlan]) -> int:
        if self.pad_duration is not None:
            return int(round(self.pad_duration * float(spec.frequency)))        # dataloader.py:69-98
        lens = {p.n_out for p in plans}
        if len(lens) != 1:
            raise RuntimeError(f"Failed to collate data with lengths {sorted(lens)}\nDo you need specifying padding in SegmentDataset?")
        return lens.pop()

    def feature(self, spec: FeatureSpec, segments: tp.Sequence[tp.Any], exact: bool = False) -> PackedFeature | torch.Tensor:
        """One modality of a batch.  Default: the bf16 projector operand (`PackedFeature`); `exact=True`: the reference-layout
        f32 [B, C, T] tensor with no rounding (what a feature plugin's `__call__` returns, text.py:85-124)."""
        store, dev = self.store, self.store.device
        plans = [self.plan(s, spec) for s in segments]
        B, T = len(segments), self._steps(spec, plans)
        L, D = store.channels[spec.name]
        if spec.kind == "words":
            ptr = np.zeros(B * T + 1, dtype=np.int64)
            idx = []
            for b, p in enumerate(plans):
                inside = p.steps < T                                                  # cropping by pad_duration
                steps, rows = p.steps[inside], p.rows[inside]
                order = np.argsort(steps, kind="stable")                              # per step, keep event order
                np.add.at(ptr, b * T + steps + 1, 1)
                idx.append(rows[order])
            np.cumsum(ptr, out=ptr)
            idx_np = np.concatenate(idx) if idx else np.zeros(0, np.int32)
            ptr_t, idx_t = torch.from_numpy(ptr.astype(np.int32)).to(dev), torch.from_numpy(idx_np.astype(np.int32)).to(dev)
            if exact:
                rows = ops.word_bag(store.word_table(spec.name), ptr_t, idx_t, B * T, f32=True)          # [B*T, C]
                return ops.transpose_f32(rows.view(B, T, L * D))                                          # [B, C, T]
            packed = ops.word_bag(store.word_table(spec.name), ptr_t, idx_t, B * T)
            return PackedFeature(packed, B, L, D, T)
        pieces = []
        seg_ptr = np.zeros(B + 1, dtype=np.int32)
        for b, p in enumerate(plans):
            rec = p.pieces
            if len(rec) and int((rec["dst_first"] + rec["dst_count"]).max()) > T:     # cropping by pad_duration
                rec = rec.copy()
                over = rec["dst_first"] + rec["dst_count"] - T
                rec["dst_count"] -= np.maximum(over, 0)
                rec["src_count"] = np.where(rec["src_count"] == 1, 1, rec["dst_count"])
                rec = rec[rec["dst_count"] > 0]
            pieces.append(rec)
            seg_ptr[b + 1] = seg_ptr[b] + len(rec)
        table = np.concatenate(pieces) if pieces else np.zeros(0, _lib.FEATURE_PIECE_DTYPE)
        if len(table) == 0:
            table = np.zeros(1, _lib.FEATURE_PIECE_DTYPE)                             # keep the device pointer non-null
        else:
            bad = (table["src_first"] < 0) | (table["src_first"] + table["src_count"] > table["ld"]) | (table["dst_first"] < 0) | \
                  (table["dst_first"] + table["dst_count"] > T) | ((table["src_count"] != table["dst_count"]) & (table["src_count"] != 1))
            if bad.any():
                raise RuntimeError(f"segment plan out of bounds for feature {spec.name!r}: {table[bad][:3]}")
        pieces_t = torch.from_numpy(table.view(np.uint8).reshape(-1)).to(dev)
        seg_t = torch.from_numpy(seg_ptr).to(dev)
        C = L * D
        if spec.kind == "target" or exact:
            return ops.segment_gather(pieces_t, seg_t, B, C, T, packed=False)
        return PackedFeature(ops.segment_gather(pieces_t, seg_t, B, C, T, packed=True), B, L, D, T)

    def subject_ids(self, segments: tp.Sequence[tp.Any]) -> torch.Tensor:
        """SubjectEncoder (subject.py:84-149): label of the segment's FIRST event -> index among the sorted labels."""
        if self.subject_index is None:
            raise ValueError("Must pass subject_index (label -> index) before using the subject feature.")
        ids = []
        for s in segments:
            e = s.ns_events[0]
            label = getattr(e, "subject", None) or e.extra["subject"]
            ids.append([self.subject_index[label]])
        return torch.tensor(ids, dtype=torch.long, device=self.store.device)

    def batch(self, segments: tp.Sequence[tp.Any], names: tp.Sequence[str] | None = None) -> SegmentData:
        data: dict[str, tp.Any] = {}
        for name in (names if names is not None else self.store.specs):
            data[name] = self.feature(self.store.specs[name], segments)
        if self.subject_index is not None:
            data["subject_id"] = self.subject_ids(segments)
        return SegmentData(data=data, segments=list(segments))
