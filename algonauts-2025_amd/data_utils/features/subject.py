"""Subject label feature.

Mirror of the reference `SubjectEncoder` (/root/reference/data_utils/data_utils/features/subject.py:23-149): `prepare(events)` maps
the sorted set of `subject` labels to indices; `__call__` returns the index of the segment's FIRST event as an int64 tensor `[1]`
(the collated batch is `subject_id: int64 [B, 1]`, the index `SubjectLayers` and the subject embedding gather by).  Pure host
bookkeeping -- no kernel involved; `GpuSegmentLoader.subject_ids` is the batched form."""

from __future__ import annotations

import logging
import typing as tp

import pydantic
import torch

from ..helpers import EventTypesHelper, extract_events

logger = logging.getLogger(__name__)


class SubjectEncoder(pydantic.BaseModel):
    model_config = pydantic.ConfigDict(protected_namespaces=(), extra="forbid")
    name: tp.Literal["SubjectEncoder"] = "SubjectEncoder"
    frequency: float = 0.0
    _label_to_ind: dict[str, int] = pydantic.PrivateAttr(default_factory=dict)
    _missing_default: torch.Tensor | None = pydantic.PrivateAttr(default=None)
    _event_types_helper: tp.Any = pydantic.PrivateAttr(default=None)

    def model_post_init(self, log__: tp.Any) -> None:
        super().model_post_init(log__)
        self._event_types_helper = EventTypesHelper("Event")

    @staticmethod
    def _extract_event_field(event: tp.Any) -> str:
        if getattr(event, "subject", None):
            return event.subject
        return event.extra["subject"]

    def prepare(self, obj: tp.Any) -> None:
        events = extract_events(obj, types=self._event_types_helper)
        if not all(getattr(e, "subject", None) or "subject" in getattr(e, "extra", {}) for e in events):
            raise TypeError(f"Field subject not found in events for {self.__class__.__name__}")
        labels = {self._extract_event_field(e) for e in events}
        if len(labels) < 2:
            logger.warning(f"SubjectEncoder has only found one label: {labels}. This was probably not intended.")
        self._label_to_ind = {label: i for i, label in enumerate(sorted(labels))}
        if events:
            self(events[0], events[0].start, duration=0.001, trigger=events[0].to_dict())

    @property
    def subject_index(self) -> dict[str, int]:
        return dict(self._label_to_ind)

    def get_static(self, event: tp.Any) -> torch.Tensor:
        if not self._label_to_ind:
            raise ValueError("Must call subject_encoder.prepare(events) before using the feature.")
        return torch.tensor([self._label_to_ind[self._extract_event_field(event)]], dtype=torch.long)

    def __call__(self, events: tp.Any, start: float, duration: float, trigger: tp.Any = None) -> torch.Tensor:
        assert duration >= 0.0, f"{duration} must be >= 0."
        found = extract_events(events, types=self._event_types_helper)
        if not found and self._missing_default is not None:
            return self._missing_default
        if not found:
            raise ValueError(f"No Event found in segment for feature {self.__class__.__name__} and feature shape not populated "
                             '(you may need to call "prepare" on the feature).')
        tensor = self.get_static(found[0])
        if self._missing_default is None:
            self._missing_default = torch.zeros((), dtype=tensor.dtype)
        return tensor
