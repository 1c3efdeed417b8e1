"""On-disk feature cache: raw array bytes in one `.data` file + one JSON line per item.

The reference keeps extractor outputs in exca `MapInfra` caches (`cache_type="MemmapArrayFile"` for text / audio /
video, text.py:204-209, audio.py:253-257, video.py:191-196; `"NumpyMemmapArray"` for fMRI, neuro.py:131-135): per
item a record {filename, offset, shape, dtype} in a `*-info.jsonl` file pointing into a shared binary file that is
memory-mapped on load.  exca is not installed here and the reference ships no cache files, so the exact spelling of
its records cannot be checked offline -- PARITY UNPINNED for the file format.  This module therefore
  * defines this build's own format with exactly those four fields plus the item key (`write_cache`), and
  * reads any jsonl whose lines carry `filename`, `offset`, `shape`, `dtype` and a key under "#key" / "key" / "uid"
    (`iter_cache`), skipping lines without them (header / metadata lines),
so a cache written by the reference's stack is expected to load, but only this build's own files are tested.

Loading copies each memory-mapped array from the page cache to HBM once (`load_into_store` -> `HbmFeatureStore.put`);
nothing stays resident on the host beyond one item (one run of words).
"""

from __future__ import annotations

import json
import typing as tp
from pathlib import Path

import numpy as np

_KEYS = ("#key", "key", "uid")


def write_cache(folder: str | Path, items: tp.Mapping[str, np.ndarray] | tp.Iterable[tuple[str, np.ndarray]], name: str = "features") -> Path:
    """Append arrays to `<folder>/<name>.data`, one info line each to `<folder>/<name>-info.jsonl`.  Returns the info path."""
    folder = Path(folder)
    folder.mkdir(parents=True, exist_ok=True)
    data_path, info_path = folder / f"{name}.data", folder / f"{name}-info.jsonl"
    pairs = items.items() if hasattr(items, "items") else items
    with open(data_path, "ab") as fd, open(info_path, "a") as fi:
        for key, arr in pairs:
            arr = np.ascontiguousarray(arr)
            pad = (-fd.tell()) % 64                      # 64-byte aligned items: any dtype can be viewed in place
            fd.write(b"\0" * pad)
            rec = {"#key": str(key), "filename": data_path.name, "offset": fd.tell(), "shape": list(arr.shape), "dtype": str(arr.dtype)}
            fd.write(arr.tobytes())
            fi.write(json.dumps(rec) + "\n")
    return info_path


def iter_cache(folder: str | Path) -> tp.Iterator[tuple[str, np.ndarray]]:
    """(key, read-only memory-mapped array) for every item of every `*-info.jsonl` in `folder`; later lines win per key."""
    folder = Path(folder)
    maps: dict[str, np.memmap] = {}
    records: dict[str, dict] = {}
    for info in sorted(folder.glob("*-info.jsonl")):
        for line in info.read_text().splitlines():
            line = line.strip()
            if not line:
                continue
            try:
                rec = json.loads(line)
            except json.JSONDecodeError:
                continue
            if not isinstance(rec, dict) or not all(k in rec for k in ("filename", "offset", "shape", "dtype")):
                continue
            key = next((rec[k] for k in _KEYS if k in rec), None)
            if key is None:
                continue
            records[str(key)] = rec
    for key, rec in records.items():
        fn = rec["filename"]
        if fn not in maps:
            maps[fn] = np.memmap(folder / fn, mode="r", dtype=np.uint8)
        dtype = np.dtype(rec["dtype"])
        shape = tuple(int(s) for s in rec["shape"])
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        off = int(rec["offset"])
        if off < 0 or off + nbytes > maps[fn].shape[0]:
            raise ValueError(f"cache item {key!r}: bytes [{off}, {off + nbytes}) outside {fn} ({maps[fn].shape[0]} bytes)")
        yield key, maps[fn][off:off + nbytes].view(dtype).reshape(shape)


def load_into_store(store: tp.Any, name: str, folder: str | Path, events_by_key: tp.Mapping[str, tp.Any]) -> int:
    """Upload every cached item whose key names an event to `store` (an HbmFeatureStore).  Word features are uploaded
    in runs so that the layer aggregation runs once per run, not once per word.  Returns the number of items loaded."""
    spec = store.specs[name]
    n = 0
    if spec.kind == "words":
        run_events, run_arrays = [], []

        def flush() -> None:
            if run_events:
                store.put_words(name, list(run_events), np.stack(run_arrays))
                run_events.clear()
                run_arrays.clear()

        for key, arr in iter_cache(folder):
            ev = events_by_key.get(key)
            if ev is None:
                continue
            run_events.append(ev)
            run_arrays.append(np.asarray(arr))
            n += 1
            if len(run_events) >= 4096:
                flush()
        flush()
        return n
    for key, arr in iter_cache(folder):
        ev = events_by_key.get(key)
        if ev is None:
            continue
        store.put(name, ev, np.asarray(arr))
        n += 1
    return n


class FeatureCacheFile:
    """Dict-like view of one cache folder (key -> array) for the feature plugins' `infra.folder`: `in`, `[]` (memory-mapped
    read), `[] =` (append).  The index is read on first use and kept in step with this object's own writes."""

    def __init__(self, folder: str | Path) -> None:
        self.folder = Path(folder)
        self._items: dict[str, np.ndarray] | None = None

    def _index(self) -> dict[str, np.ndarray]:
        if self._items is None:
            self._items = dict(iter_cache(self.folder)) if self.folder.exists() else {}
        return self._items

    def __contains__(self, key: str) -> bool:
        return str(key) in self._index()

    def __getitem__(self, key: str) -> np.ndarray:
        return self._index()[str(key)]

    def __setitem__(self, key: str, arr: np.ndarray) -> None:
        write_cache(self.folder, [(str(key), arr)])
        self._items = None                      # re-map on next read: the data file grew

    def __len__(self) -> int:
        return len(self._index())
