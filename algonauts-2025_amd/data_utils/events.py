"""Event records the segment loader reads: the fields of the reference's pydantic events
(/root/reference/data_utils/data_utils/events.py:25-346) that the hot path touches -- `start`, `duration`, `timeline`,
the class name as `type`, `filepath` / `offset` of media events, `text` of words, `subject` of recordings.
Plain dataclasses: media decoding (`read`), dataframe round trips and validation stay with the reference's study
loaders, which are outside this build's scope.  The loader duck-types, so the reference's own event objects work too."""

from __future__ import annotations

import dataclasses
import typing as tp


@dataclasses.dataclass(eq=False)
class Event:
    start: float
    timeline: str = ""
    duration: float = 0.0
    extra: dict[str, tp.Any] = dataclasses.field(default_factory=dict)

    @property
    def type(self) -> str:
        return self.__class__.__name__

    @property
    def stop(self) -> float:
        return self.start + self.duration

    def to_dict(self) -> dict[str, tp.Any]:
        out = {f.name: getattr(self, f.name) for f in dataclasses.fields(self) if f.name != "extra"}
        out["type"] = self.type
        out.update(self.extra)
        return out


@dataclasses.dataclass(eq=False)
class Word(Event):
    text: str = ""
    language: str = ""
    context: str = ""
    sentence: str = ""


@dataclasses.dataclass(eq=False)
class Sound(Event):
    filepath: str = ""
    frequency: float = 0.0
    offset: float = 0.0


@dataclasses.dataclass(eq=False)
class Video(Event):
    filepath: str = ""
    frequency: float = 0.0
    offset: float = 0.0


@dataclasses.dataclass(eq=False)
class Fmri(Event):
    filepath: str = ""
    frequency: float = 0.0
    subject: str = ""


def __getattr__(name: str) -> tp.Any:   # `Segment` lives in segments.py, as in the reference; kept importable from here
    if name == "Segment":
        from .segments import Segment
        return Segment
    raise AttributeError(name)
