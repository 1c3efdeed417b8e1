"""`extract_events` / `EventTypesHelper` for the inputs the feature plugins receive on the hot path.

Host mirror of /root/reference/data_utils/data_utils/helpers.py:18-66 and events.py:109-126: an event, a dict, a list of
events or a list of segments becomes a flat list of events, optionally filtered by event type (a type matches its
subclasses too).  The DataFrame entry point belongs to the study loaders (out of scope); a frame is accepted only in the
trivial way -- rows are turned into this build's event records through their `type` column."""

from __future__ import annotations

import inspect
import typing as tp

from . import events as _events


def _classes() -> dict[str, type]:
    return {n: c for n, c in vars(_events).items() if inspect.isclass(c) and issubclass(c, _events.Event)}


class EventTypesHelper:
    """events.py:109-126: `classes` to isinstance against, `names` of every matching (sub)class."""

    def __init__(self, event_types: str | type | tp.Sequence[str]) -> None:
        self.specified = event_types
        known = _classes()
        if inspect.isclass(event_types):
            self.classes: tuple[type, ...] = (event_types,)
        else:
            if isinstance(event_types, str):
                event_types = (event_types,)
            try:
                self.classes = tuple(known[x] for x in event_types)
            except KeyError as e:
                raise ValueError(f"{event_types} is an invalid event name, use one of {list(known)}") from e
        self.names = [n for n, c in known.items() if issubclass(c, self.classes)]

    def matches(self, event: tp.Any) -> bool:
        # duck-typed events (the reference's own pydantic events work too): compare the class name / `type` field
        return isinstance(event, self.classes) or getattr(event, "type", event.__class__.__name__) in self.names


def _from_dict(row: tp.Mapping[str, tp.Any]) -> tp.Any:
    known = _classes()
    cls = known.get(str(row.get("type", "")))
    if cls is None:
        raise ValueError(f"Unknown event type {row.get('type')!r}, use one of {list(known)}")
    import dataclasses

    names = {f.name for f in dataclasses.fields(cls)} - {"extra"}
    kwargs = {k: v for k, v in row.items() if k in names}
    extra = {k: v for k, v in row.items() if k not in names and k != "type"}
    return cls(**kwargs, extra=extra)


def extract_events(obj: tp.Any, types: tp.Any = None) -> list[tp.Any]:
    helper = types if isinstance(types, EventTypesHelper) or types is None else EventTypesHelper(types)
    if hasattr(obj, "iterrows") and hasattr(obj, "columns"):          # a pandas frame of events
        obj = [_from_dict(dict(r)) for _, r in obj.iterrows()]
    if isinstance(obj, dict):
        obj = [_from_dict(obj)]
    elif hasattr(obj, "start") and hasattr(obj, "duration") and not hasattr(obj, "ns_events"):
        obj = [obj]
    if not isinstance(obj, (list, tuple)):
        raise NotImplementedError(f"Conversion of {type(obj)} is not supported")
    if not obj:
        return []
    if hasattr(obj[0], "ns_events"):                                  # segments: each event object once (helpers.py:56-60)
        seen: dict[int, tp.Any] = {}
        for seg in obj:
            for e in seg.ns_events:
                seen.setdefault(id(e), e)
        obj = list(seen.values())
    if helper is not None:
        obj = [e for e in obj if helper.matches(e)]
    return list(obj)
