"""The feature-plugin surface of the reference's extractors, on HBM-resident states.

Every extractor of the reference (`LLAMA3p2` text.py:42-256, `Wav2VecBert` audio.py:27-263, `VJEPA2` video.py:56-236) is a
pydantic model with the SAME three entry points, repeated verbatim in each file:

    prepare(events)                              compute + cache the hidden states of every event (`_get_data`), then one
                                                 dry call to fix the missing-feature default (text.py:63-78)
    __call__(events, start, duration, trigger)   -> Tensor [L, D, T]: the cached states of the segment's events cut to the
                                                 window (`TimedArray.overlap`), layer-aggregated, summed into a 2 Hz grid
                                                 (text.py:80-124); invoked by SegmentDataset.__getitem__ (dataloader.py:101-108)
    _get_data(events)                            -> Iterator[np.ndarray [n_states, D(, T_event)]]: the model forward, cached
                                                 per item uid by exca's MapInfra

`HbmFeaturePlugin` implements them once for this build: `_get_data` runs the HIP extractor of the subclass (`_compute`) and
caches per item uid (RAM, optional folder); the states are uploaded and layer-aggregated on the GPU once (`HbmFeatureStore`);
`__call__` is the index arithmetic of `TimedArray` on the host (data_utils/base.py, pinned by G10 / G11) plus ONE gather launch
(`tribe_segment_gather_fwd` / `tribe_word_bag_f32_fwd` + `tribe_transpose_f32_fwd`) and returns the f32 [L, D, T] tensor ON
THE GPU -- bit-identical to the reference's host tensor (tests/test_gpu_plugins.py, through G11).  For batches, hand the
plugin's `store` to `GpuSegmentLoader`, which writes the projector operand directly and skips the per-segment tensors.
There is no CPU path: `device="cpu"` (or "auto" without a GPU) raises when data is requested."""

from __future__ import annotations

import types
import typing as tp

import numpy as np
import pydantic
import torch

from ..base import Frequency
from ..gpu_loader import FeatureSpec, GpuSegmentLoader, HbmFeatureStore
from ..helpers import EventTypesHelper, extract_events
from ..infra import MapInfra
from .layers import aggregate_layers, layer_indices


class HbmFeaturePlugin(pydantic.BaseModel):
    model_config = pydantic.ConfigDict(protected_namespaces=(), extra="forbid", arbitrary_types_allowed=True)

    layers: list[float] = [0.5, 0.75, 1.0]
    layer_aggregation: tp.Literal["group_mean"] | None = "group_mean"
    device: tp.Literal["auto", "cpu", "cuda"] = "auto"
    infra: MapInfra = MapInfra()

    # what distinguishes the three extractors as far as assembly goes
    _EVENT_TYPE: tp.ClassVar[str] = "Event"
    _KIND: tp.ClassVar[str] = "sampled"                 # "sampled": [n_states, D, T_event] at 2 Hz; "words": [n_states, D] held for the word
    _PASS_EVENT_DURATION: tp.ClassVar[bool] = False     # video.py:176-182 validates against event.duration, audio.py:241 does not
    _FREQUENCY: tp.ClassVar[float] = 2.0
    _START_SHIFT: tp.ClassVar[float] = 0.0              # neuro.py:150: the recording sits 4.47 s before its event
    _FIRST_EVENT_ONLY: tp.ClassVar[bool] = False        # neuro.py:81: events[:1]

    _event_types_helper: tp.Any = pydantic.PrivateAttr(default=None)
    _missing_default: torch.Tensor | None = pydantic.PrivateAttr(default=None)
    _store: tp.Any = pydantic.PrivateAttr(default=None)
    _loader: tp.Any = pydantic.PrivateAttr(default=None)
    _spec: tp.Any = pydantic.PrivateAttr(default=None)
    _ram: dict = pydantic.PrivateAttr(default_factory=dict)       # item uid -> np.ndarray (exca's keep_in_ram)
    _disk: tp.Any = pydantic.PrivateAttr(default=None)
    _resident: set = pydantic.PrivateAttr(default_factory=set)    # ids of events already uploaded
    _n_states: int | None = pydantic.PrivateAttr(default=None)

    def model_post_init(self, log__: tp.Any) -> None:
        super().model_post_init(log__)
        self._event_types_helper = EventTypesHelper(self._EVENT_TYPE)
        if self.device == "auto":
            self.device = "cuda" if torch.cuda.is_available() else "cpu"

    # -- reference helpers kept under their names ------------------------------------------------------------------------
    def _aggregate_layers(self, latents: np.ndarray) -> np.ndarray:
        return aggregate_layers(latents, self.layers, self.layer_aggregation)

    @classmethod
    def _exclude_from_cls_uid(cls) -> list[str]:
        return ["device"]

    def _exclude_from_cache_uid(self) -> list[str]:
        return ["device"] + ["layers", "layer_aggregation"]

    # -- to be provided by the extractor ---------------------------------------------------------------------------------
    def _item_uid(self, event: tp.Any) -> str:
        raise NotImplementedError

    def _compute(self, events: list[tp.Any]) -> tp.Iterator[np.ndarray]:
        """The extractor forward for `events` (HIP), one array per event, in order."""
        raise NotImplementedError

    # -- HBM residency ---------------------------------------------------------------------------------------------------
    def _require_gpu(self) -> None:
        if self.device != "cuda" or not torch.cuda.is_available():
            raise RuntimeError(f"{type(self).__name__}: this build runs its extractors and feature assembly on an MI355X only "
                               f"(device={self.device!r}, GPU visible: {torch.cuda.is_available()}); there is no CPU path")

    @property
    def spec(self) -> FeatureSpec:
        if self._spec is None:
            self._spec = self._make_spec(self.name)  # type: ignore[attr-defined]
        return self._spec

    def _make_spec(self, key: str) -> FeatureSpec:
        return FeatureSpec(key, self._KIND, self._EVENT_TYPE, frequency=self._FREQUENCY, layers=tuple(self.layers),
                           layer_aggregation=self.layer_aggregation, pass_event_duration=self._PASS_EVENT_DURATION,
                           start_shift=self._START_SHIFT, first_event_only=self._FIRST_EVENT_ONLY)

    @property
    def frequency(self) -> float:
        """Output grid in Hz (dataloader.py:69-87 reads it to turn `pad_duration` into a length)."""
        return self._FREQUENCY

    @property
    def store(self) -> HbmFeatureStore:
        if self._store is None:
            self._require_gpu()
            self.bind(HbmFeatureStore([self.spec], device="cuda"))
        return self._store

    def bind(self, store: HbmFeatureStore, name: str | None = None) -> "HbmFeaturePlugin":
        """Share one `HbmFeatureStore` between plugins (and a `GpuSegmentLoader`): the feature is filed under `name`
        (default: the spec already registered under this plugin's `name`, else it is added)."""
        key = name or self.spec.name
        if key != self.spec.name:
            self._spec = self._make_spec(key)
        store.specs.setdefault(key, self.spec)
        self._spec = store.specs[key]
        self._store, self._loader = store, GpuSegmentLoader(store)
        self._resident = set()
        return self

    def _disk_cache(self) -> tp.Any:
        if self.infra.folder is None:
            return None
        if self._disk is None:
            from pathlib import Path

            from ..cache_file import FeatureCacheFile

            self._disk = FeatureCacheFile(Path(self.infra.folder) / f"{type(self).__name__}-{self.infra.version}")
        return self._disk

    def _get_data(self, events: list[tp.Any]) -> tp.Iterator[np.ndarray]:
        """One `[n_states, D(, T_event)]` array per event, computed by the HIP extractor for items not cached yet
        (`infra.apply(item_uid=..., cache_type="MemmapArrayFile")` in the reference: text.py:199-203, audio.py:145-149)."""
        events = list(events)
        uids = [self._item_uid(e) for e in events]
        disk = self._disk_cache()
        force = self.infra.mode == "force"
        todo, seen = [], set()
        for e, u in zip(events, uids):
            if u in seen or (not force and (u in self._ram or (disk is not None and u in disk))):
                continue
            seen.add(u)
            todo.append((e, u))
        fresh: dict[str, np.ndarray] = {}
        if todo:
            if self.infra.mode == "read-only":
                raise RuntimeError(f"{type(self).__name__}: {len(todo)} item(s) missing from a read-only cache")
            self._require_gpu()
            for (e, u), arr in zip(todo, self._compute([e for e, _ in todo])):
                arr = np.asarray(arr)
                fresh[u] = arr
                if disk is not None:
                    disk[u] = arr
                if self.infra.keep_in_ram:
                    self._ram[u] = arr
        for u in uids:
            if u in fresh:
                yield fresh[u]
            elif u in self._ram:
                yield self._ram[u]
            else:
                yield disk[u]

    def _ensure_resident(self, events: list[tp.Any]) -> None:
        store = self.store
        new = [e for e in events if id(e) not in self._resident]
        if not new:
            return
        arrays = list(self._get_data(new))
        if self._n_states is None and arrays:
            self._n_states = int(arrays[0].shape[0])
        if self._KIND == "words":
            store.put_words(self.spec.name, new, np.stack(arrays))
        else:
            for e, a in zip(new, arrays):
                store.put(self.spec.name, e, a)
        self._resident.update(id(e) for e in new)

    # -- the reference's entry points ------------------------------------------------------------------------------------
    def prepare(self, obj: tp.Any) -> None:
        events = extract_events(obj, types=self._event_types_helper)
        self._ensure_resident(events)
        if events:
            self(events[0], start=events[0].start, duration=0.001, trigger=events[0].to_dict())

    def __call__(self, events: tp.Any, start: float, duration: float, trigger: float | dict[str, tp.Any] | None = None) -> torch.Tensor:
        assert duration >= 0.0, f"{duration} must be >= 0."
        events = extract_events(events, types=self._event_types_helper)
        freq = Frequency(self._FREQUENCY)
        if not events and self._missing_default is not None:
            default = self._missing_default
            n_times = max(1, freq.to_ind(duration))
            return default.unsqueeze(-1).repeat([1 for _ in range(default.ndim)] + [n_times])
        if not events:   # text.py:108-117 with nothing to add: the empty accumulator, shape (0, T)
            return torch.zeros(0, max(1, freq.to_ind(duration)), device="cuda" if self.device == "cuda" else "cpu")
        if self._FIRST_EVENT_ONLY:
            events = events[:1]
        self._ensure_resident(events)
        window = types.SimpleNamespace(ns_events=events, start=start, duration=duration)
        flat = self._loader.feature(self.spec, [window], exact=True)[0]                    # f32 [L*D, T] on the GPU
        self._loader.clear_plans()
        L, D = self.store.channels[self.spec.name]
        squeeze = self._KIND == "target" or (self.layer_aggregation is not None and len(layer_indices(self._n_states or 2, self.layers)) == 1)
        tensor = flat.view(D, -1) if squeeze else flat.view(L, D, -1)
        if self._missing_default is None:
            self._missing_default = torch.zeros(*tensor.shape[:-1], dtype=tensor.dtype, device=tensor.device)
        return tensor
