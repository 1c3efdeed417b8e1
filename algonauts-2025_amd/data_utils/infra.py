"""`infra:` block of a feature plugin.

The reference's extractors carry `infra: MapInfra = MapInfra()` (exca; text.py:51, audio.py:37, video.py:69) and decorate
`_get_data` with `infra.apply(item_uid=..., cache_type="MemmapArrayFile")`: per-item results are cached in RAM and, when
`folder` is set, on disk.  exca is not a dependency of this build and cluster submission (`cluster="slurm"`, job arrays) is
control plane, out of scope.  This model keeps the FIELD a config file sets so reference configs validate, and honours the
three that shape the data path: `folder` (on-disk cache through data_utils/cache_file.py), `mode` and `keep_in_ram`.
Every other key of the reference's `infra` blocks is accepted and ignored."""

from __future__ import annotations

import typing as tp

import pydantic


class MapInfra(pydantic.BaseModel):
    model_config = pydantic.ConfigDict(extra="allow")
    folder: str | None = None                                                # None: RAM / HBM only
    mode: tp.Literal["cached", "force", "read-only"] = "cached"              # force: recompute even when cached
    keep_in_ram: bool = True
    cluster: str | None = None                                               # accepted; never submitted anywhere
    version: str = "0"
