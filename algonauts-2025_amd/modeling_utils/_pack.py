"""Version-keyed cache of device-side packed (bf16, re-laid-out) copies of fp32 parameters.

The fp32 tensors stay the single source of truth (state_dict / checkpoints / optimizer);
a packed copy is rebuilt only when a source tensor was written (torch bumps `_version` on
every in-place update, e.g. optimizer.step or load_state_dict) or moved."""

from __future__ import annotations

import typing as tp

import torch


class PackCache:
    def __init__(self) -> None:
        self._store: dict[str, tuple[tuple, tp.Any]] = {}

    @staticmethod
    def _sig(tensors: tp.Sequence[torch.Tensor | None]) -> tuple:
        return tuple(None if t is None else (t.data_ptr(), t._version, str(t.device), tuple(t.shape)) for t in tensors)

    def get(self, key: str, tensors: tp.Sequence[torch.Tensor | None], build: tp.Callable[[], tp.Any]) -> tp.Any:
        sig = self._sig(tensors)
        hit = self._store.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        value = build()
        self._store[key] = (sig, value)
        return value

    def clear(self) -> None:
        self._store.clear()


def f32c(t: torch.Tensor) -> torch.Tensor:
    """detached, contiguous fp32 view of a parameter (no copy when already so)."""
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
