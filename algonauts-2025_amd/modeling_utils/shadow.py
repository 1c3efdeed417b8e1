"""bf16 shadows of f32 parameters: the packed GEMM operand a Linear's forward reads.

The autograd path packs every f32 weight to bf16 when its version changed (modeling_utils/autograd.py `PACKS`, `QKV_PACKS`) -- after
every optimiser step that is a cast pass over all 950 M parameters (2.0 ms of the B = 16 step).  A pack that is a plain element-wise
cast (no padding: K % 64 == 0) registers itself here; `HipAdam.step` hands its pointer to the kernel, which writes the bf16 value next to
the f32 one, and then tells the owner which parameter version the shadow now holds.  Any other in-place write to the parameter bumps its
`_version` past that, so the pack is rebuilt the usual way (load_state_dict, SWA swap, manual edits)."""

from __future__ import annotations

import typing as tp
import weakref

import torch

# id(param) -> (weak ref to the parameter, bf16 tensor with the parameter's numel in its element order, callback(version))
_SHADOWS: dict[int, tuple[tp.Any, torch.Tensor, tp.Callable[[int], None]]] = {}


def register(param: torch.Tensor, bf16: torch.Tensor, on_update: tp.Callable[[int], None]) -> None:
    if bf16.dtype != torch.bfloat16 or bf16.numel() != param.numel() or not bf16.is_contiguous() or bf16.device != param.device:
        raise ValueError("shadow: the bf16 copy must be a contiguous tensor of the parameter's size on its device")
    key = id(param)
    _SHADOWS[key] = (weakref.ref(param, lambda _r, k=key: _SHADOWS.pop(k, None)), bf16, on_update)


def lookup(param: torch.Tensor) -> tuple[torch.Tensor, tp.Callable[[int], None]] | None:
    hit = _SHADOWS.get(id(param))
    if hit is None or hit[0]() is not param:
        return None
    return hit[1], hit[2]


def forget(param: torch.Tensor) -> None:
    _SHADOWS.pop(id(param), None)
