"""Optimiser step of the training loop on HIP.

The reference builds `torch.optim.Adam(lr=1e-4, weight_decay=0)` under a `OneCycleLR` schedule
(/root/reference/algonauts2025/grids/defaults.py:126-141, pl_module.py:138-144).  Stock torch Adam walks the ~150
parameter tensors with a dozen multi-tensor launches per step (≈ 19 ms of a 184 ms step on the full-size model);
`HipAdam` is a drop-in `torch.optim.Optimizer` (same param_groups / state_dict keys: `step`, `exp_avg`, `exp_avg_sq`, so
schedulers and checkpoints interchange) whose `step()` is ONE launch per parameter group (tribe_adam_step): 16 B read +
12 B written per parameter, HBM-bound."""

from __future__ import annotations

import typing as tp

import numpy as np
import torch

from tribe_hip import _lib
from tribe_hip._lib import check, lib

from . import shadow


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params: tp.Any, lr: float = 1e-3, betas: tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 decoupled_weight_decay: bool = False) -> None:
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError(f"Invalid Adam hyper-parameters: lr={lr} betas={betas} eps={eps} weight_decay={weight_decay}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled_weight_decay=decoupled_weight_decay))
        self._chunks: dict[tuple, tuple[torch.Tensor, torch.Tensor]] = {}

    def _work_list(self, gi: int, params: list[torch.Tensor]) -> tuple[torch.Tensor, torch.Tensor]:
        sig = (gi,) + tuple(p.numel() for p in params)
        hit = self._chunks.get(sig)
        if hit is None:
            chunk = int(lib().tribe_adam_chunk_elems())
            owner, start = [], []
            for i, n in enumerate(sig[1:]):
                s = np.arange(0, n, chunk, dtype=np.int64)
                owner.append(np.full(len(s), i, dtype=np.int32))
                start.append(s)
            dev = params[0].device
            if len(self._chunks) >= 16:      # parameter subsets change with modality dropout; keep the cache bounded
                self._chunks.pop(next(iter(self._chunks)))
            hit = (torch.from_numpy(np.concatenate(owner)).to(dev), torch.from_numpy(np.concatenate(start)).to(dev))
            self._chunks[sig] = hit
        return hit

    @torch.no_grad()
    def step(self, closure: tp.Callable[[], torch.Tensor] | None = None) -> torch.Tensor | None:
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            # torch.optim.Adam keeps a step count PER PARAMETER: one that had grad None on earlier steps (a projector whose
            # modality was dropped, model.py:133-141) starts its bias corrections at 1 when it first receives a gradient.
            # The kernel takes one step count per launch, so parameters are launched per distinct count (one launch in
            # the steady state, two or three while late starters exist).
            by_step: dict[int, list[torch.Tensor]] = {}
            keep_alive = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.TribeHipError("HipAdam: parameters must be contiguous f32 tensors on the GPU (no CPU fallback)")
                if p.grad.is_sparse:
                    raise RuntimeError("HipAdam does not support sparse gradients")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                by_step.setdefault(int(st["step"]), []).append(p)
            b1, b2 = group["betas"]
            for step, params in by_step.items():
                table = np.zeros(len(params), dtype=_lib.ADAM_TENSOR_DTYPE)
                shadows: list[tuple[torch.Tensor, tp.Callable[[int], None]]] = []
                for i, p in enumerate(params):
                    st = self.state[p]
                    g = p.grad if (p.grad.dtype == torch.float32 and p.grad.is_contiguous()) else p.grad.float().contiguous()
                    keep_alive.append(g)                      # a converted gradient must outlive the launch
                    sh = shadow.lookup(p)        # bf16 GEMM-operand copy kept by the autograd path: the kernel refreshes it in the same pass
                    if sh is not None:
                        shadows.append((p, sh[1]))
                    table[i] = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                                sh[0].data_ptr() if sh is not None else 0)
                owner, start = self._work_list(gi, params)
                table_t = torch.from_numpy(table.view(np.uint8).reshape(-1)).to(params[0].device)
                keep_alive.append(table_t)
                check(lib().tribe_adam_step(table_t.data_ptr(), owner.data_ptr(), start.data_ptr(), owner.numel(), float(group["lr"]), float(b1),
                                            float(b2), float(group["eps"]), float(group["weight_decay"]), step, int(group["decoupled_weight_decay"]),
                                            torch.cuda.current_stream().cuda_stream), "tribe_adam_step")
                # the kernel wrote through raw pointers: tell autograd (and this build's packed-weight caches, which key on
                # `_version`) that the parameters changed
                torch.autograd.graph.increment_version(params)
                for p, on_update in shadows:
                    on_update(p._version)
            del keep_alive      # torch's caching allocator keeps freed blocks ordered on the launch stream
        return loss
