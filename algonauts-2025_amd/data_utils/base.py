"""Time-axis bookkeeping of the feature pipeline: `Frequency` and `TimedArray`.

Host-side mirror of /root/reference/data_utils/data_utils/base.py:39-211 (same class and method names,
same rounding, same error messages where callers can see them).  Every feature of the reference is cut
out of its cached extractor output by this arithmetic (`overlap`) and summed into the segment's output
array (`+=`), so the GPU segment loader (gpu_loader.py) must reproduce its index decisions bit for bit;
the scalar form lives here, the vectorised form used for batches is `overlap_window`.

Index rules restated (base.py:41-61,164-196):
  * seconds -> samples is round-half-to-even of `seconds * frequency` (Python `round` / `numpy.round`);
  * an overlap query never returns an empty slice of a sampled array: a rounded length <= 0 becomes 1;
  * the slice is pulled back inside the array when start + length would run past its end.
"""

from __future__ import annotations

import typing as tp

import numpy as np


class Frequency(float):
    """A sampling rate in Hz with the two conversions the pipeline uses (base.py:39-61)."""

    def to_ind(self, seconds: tp.Any) -> tp.Any:
        if isinstance(seconds, np.ndarray):
            return np.round(seconds * self).astype(int)
        return int(round(seconds * self))

    def to_sec(self, index: tp.Any) -> tp.Any:
        return index / self


def overlap_window(frequency: float, arr_start: float, arr_len: int, arr_duration: float, q_start: tp.Any,
                   q_duration: tp.Any) -> tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Vectorised `TimedArray._overlap_slice` (base.py:164-196) of ONE array against many query windows.

    Returns (valid, out_start_sec, out_duration_sec, first_index, n_index); for `frequency == 0` the two
    index outputs are -1 (the reference returns slice None there).  Invalid rows have unspecified values."""
    q_start = np.asarray(q_start, dtype=np.float64)
    q_duration = np.asarray(q_duration, dtype=np.float64)
    if np.any(q_duration < 0):
        raise ValueError(f"duration should be >=0, got duration={float(q_duration.min())}")
    lo = np.maximum(q_start, arr_start)
    hi = np.minimum(q_start + q_duration, arr_start + arr_duration)
    valid = ~(hi < lo)
    valid &= ~((hi == lo) & bool(arr_duration) & (q_duration != 0))
    if not frequency:
        none = np.full(lo.shape, -1, dtype=np.int64)
        return valid, lo, hi - lo, none, none
    freq = float(frequency)
    first = np.asarray(np.round((lo - arr_start) * freq)).astype(np.int64)   # Frequency.to_ind, array form
    count = np.asarray(np.round((hi - lo) * freq)).astype(np.int64)
    count = np.where(count <= 0, 1, count)
    first = np.where(first > arr_len - count, arr_len - count, first)
    if np.any(valid & (first < 0)):
        raise RuntimeError(f"Fail for start={q_start} duration={q_duration} on array of {arr_len} samples at {arr_start}")
    return valid, first / freq + arr_start, count / freq, first, count


class TimedArray:
    """An array whose last axis is time (`frequency` > 0) or a value that holds for `duration` (`frequency` 0).

    Mirrors base.py:64-211: constructor validation, `overlap`, and in-place accumulation with `+=`."""

    def __init__(self, *, frequency: float, start: float, data: np.ndarray | None = None, duration: float | None = None,
                 aggregation: str = "sum") -> None:
        self.frequency = Frequency(frequency)
        self.start = start
        self.aggregation = aggregation
        if duration is not None and duration < 0:
            raise ValueError(f"duration should be None or >=0, got {duration}")
        expected = max(1, self.frequency.to_ind(duration)) if (frequency and duration is not None) else 0
        if data is None:
            if duration is None:
                raise ValueError("Missing data or duration")
            data = np.zeros((0, expected)) if frequency else np.zeros((0,))
        self.data = data
        if frequency and duration is not None:
            if not self.data.shape[-1]:
                raise ValueError(f"Last dimension is empty but frequency is not null (shape={self.data.shape})")
            if abs(data.shape[-1] - expected) > 2:
                raise ValueError(f"Data has incorrect (last) dimension {data.shape} for duration {duration} and "
                                 f"frequency {frequency} (expected {expected})")
        if frequency:
            self.duration = self.frequency.to_sec(data.shape[-1])
        elif duration is None:
            raise ValueError(f"duration must be provided if {frequency=}")
        else:
            self.duration = duration
        self._overlapping_data_count: np.ndarray | None = None
        if aggregation == "average":
            self._overlapping_data_count = np.zeros(self.data.shape[-1] if self.frequency else 1, dtype=int)
        elif aggregation != "sum":
            raise ValueError(f"Unknown {aggregation=}")

    def __repr__(self) -> str:
        fields = ",".join(f"{f}={getattr(self, f)}" for f in "frequency,start,duration,aggregation,data".split(","))
        return f"{self.__class__.__name__}({fields})"

    def _overlap_slice(self, start: float, duration: float) -> tuple[float, float, slice | None] | None:
        valid, o_start, o_dur, first, count = overlap_window(self.frequency, self.start, self.data.shape[-1] if self.frequency else 0,
                                                             self.duration, start, duration)
        if not bool(valid):
            return None
        if not self.frequency:
            return float(o_start), float(o_dur), None
        return float(o_start), float(o_dur), slice(int(first), int(first) + int(count))

    def overlap(self, start: float, duration: float) -> "TimedArray | None":
        found = self._overlap_slice(start, duration)
        if found is None:
            return None
        o_start, o_dur, sl = found
        return TimedArray(frequency=self.frequency, start=o_start, duration=o_dur, data=self.data[..., sl])

    def __iadd__(self, other: "TimedArray") -> "TimedArray":
        if other.frequency and self.frequency != other.frequency:
            if abs(self.frequency - other.frequency) * max(self.duration, other.duration) >= 0.5:
                raise ValueError(f"Cannot add with different (non-0) frequencies ({other.frequency} and {self.frequency})")
        if not self.data.size:  # first contribution fixes the leading shape
            lead = other.data.shape[: (-1 if other.frequency else None)]
            if self.frequency:
                lead += (self.data.shape[-1],)
            self.data = np.zeros(lead, dtype=other.data.dtype)
        mine: slice | None = None
        theirs: slice | None = None
        if self.frequency:
            a = self._overlap_slice(other.start, other.duration)
            b = other._overlap_slice(self.start, self.duration)
            if a is None or b is None:
                return self
            mine, theirs = a[-1], b[-1]
        if self._overlapping_data_count is None:
            self.data[..., mine] += other.data[..., theirs]
        else:
            counts = self._overlapping_data_count[..., mine]
            keep = counts / (1.0 + counts)
            self.data[..., mine] *= keep
            self.data[..., mine] += (1 - keep) * other.data[..., theirs]
            counts += 1
        return self
