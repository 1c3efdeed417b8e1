"""Cutting timelines into model windows.

Host mirror of /root/reference/data_utils/data_utils/segments.py:21-82,163-265 for event LISTS (the DataFrame entry
points -- `validate_events`, `find_overlap` on frames -- belong to the study loaders, outside this build's scope):
`Segment`, `_prepare_strided_windows`, `SegmentCreator.select`, `iter_segments` / `list_segments` with the
reference's hard-wired geometry: windows of 149 s every 149 s (100 TRs of 1.49 s), starting 4.47 s (3 TRs) before
the timeline's first event, the last incomplete window kept (segments.py:182-200)."""

from __future__ import annotations

import collections
import dataclasses
import typing as tp

import numpy as np

WINDOW_S = 149.0     # stride == duration (segments.py:189-195)
LEAD_S = 4.47        # windows start this long before the first event (the fMRI read shift of neuro.py:150)


@dataclasses.dataclass
class Segment:
    """A window of one timeline and the events that touch it (segments.py:21-82)."""

    start: float
    duration: float
    _index: tp.Any = None
    ns_events: list[tp.Any] = dataclasses.field(default_factory=list)
    _trigger: tp.Any = None

    @property
    def stop(self) -> float:
        return self.start + self.duration

    def subsegment(self, start: float, duration: float) -> "Segment":
        assert start >= 0, "Start is relative to the segment start and must be non-negative"
        lo, hi = self.start + start, self.start + start + duration
        keep = [i for i, e in enumerate(self.ns_events) if e.start <= hi and e.start + e.duration >= lo]   # closed ends, as the reference
        index = np.array([self._index[i] for i in keep]) if self._index is not None else None
        return Segment(start=lo, duration=duration, _index=index, ns_events=[self.ns_events[i] for i in keep], _trigger=self._trigger)

    def _to_feature(self) -> dict[str, tp.Any]:
        return {"start": self.start, "duration": self.duration, "events": self.ns_events, "trigger": self._trigger}


def _prepare_strided_windows(start: float, stop: float, stride: float, duration: float, drop_incomplete: bool = True) -> tuple[np.ndarray, np.ndarray]:
    """segments.py:163-176: window starts start, start + stride, ... <= stop (+1e-8); all of length `duration`."""
    if drop_incomplete:
        stop -= duration
    starts = np.arange(start, stop + 1e-8, stride)
    return starts, np.full_like(starts, fill_value=duration)


def _events_of(obj: tp.Any) -> list[tp.Any]:
    """helpers.extract_events for the two inputs the hot path uses: a list of events or a list of segments
    (segments contribute each event object once, helpers.py:56-60)."""
    if not isinstance(obj, (list, tuple)):
        raise NotImplementedError(f"Conversion of {type(obj)} is not supported")
    if not obj:
        return []
    if isinstance(obj[0], Segment) or hasattr(obj[0], "ns_events"):
        seen: dict[int, tp.Any] = {}
        for seg in obj:
            for e in seg.ns_events:
                seen.setdefault(id(e), e)
        return list(seen.values())
    return list(obj)


class SegmentCreator:
    """Overlap selection on one timeline (segments.py:229-265)."""

    def __init__(self, events: list[tp.Any]) -> None:
        timelines = {e.timeline for e in events}
        if len(timelines) > 1:
            raise ValueError(f"Cannot create {self.__class__.__name__} on several timelines, got {timelines}")
        self.events = np.empty(len(events), dtype=object)
        self.events[:] = events
        self.starts = np.array([e.start for e in events])
        self.indices = np.array([getattr(e, "_index", None) for e in events])
        self.stops = np.array([e.duration for e in events]) + self.starts

    @classmethod
    def from_obj(cls, obj: tp.Any) -> dict[str, "SegmentCreator"]:
        per_timeline: dict[str, list[tp.Any]] = collections.defaultdict(list)
        for e in _events_of(obj):
            per_timeline[e.timeline].append(e)
        return {tl: cls(evs) for tl, evs in per_timeline.items()}

    def select(self, start: float, duration: float) -> Segment:
        hit = (self.starts < start + duration) & (self.stops > start)      # open ends: touching events are excluded
        return Segment(ns_events=list(self.events[hit]), start=start, duration=duration, _index=self.indices[hit])


def iter_segments(events: tp.Any, start_jitter: float = 0.0) -> tp.Iterator[Segment]:
    """segments.py:179-200 (and callbacks.py:25-44 when `start_jitter` != 0)."""
    for creator in SegmentCreator.from_obj(events).values():
        starts, durations = _prepare_strided_windows(creator.starts.min() - LEAD_S + start_jitter, creator.stops.max() - LEAD_S + start_jitter,
                                                     WINDOW_S, WINDOW_S, drop_incomplete=False)
        for s, d in zip(starts, durations):
            seg = creator.select(start=s, duration=d)
            seg._trigger = s
            yield seg


def list_segments(events: tp.Any) -> list[Segment]:
    return list(iter_segments(events))
