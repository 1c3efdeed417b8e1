"""The fit loop around `BrainModule.training_step`: the part of `lightning.Trainer.fit` the reference's run relies on.

Reference: /root/reference/algonauts2025/main.py:351-405 builds `pl.Trainer(max_epochs, limit_train_batches, strategy=ddp...,
callbacks=[JitterWindows, StochasticWeightAveraging, ...])` and calls `trainer.fit(brain_module, train, val)`; the optimiser and
its `OneCycleLR` schedule come from `configure_optimizers` (pl_module.py:138-144, interval "step").  Lightning is not a dependency
of this build (and loggers, checkpoints, early stopping and progress bars are control plane, out of scope): this class is the
epoch / step / hook ordering only, with the MI355X pieces plugged in -- HipAdam (one launch per step), GradReducer (bucketed
RCCL all-reduce overlapped with backward, one process per GPU) and the HIP weight-averaging kernel behind the SWA callback.
Callbacks receive `(trainer, pl_module)` exactly as Lightning passes them.
"""

from __future__ import annotations

import typing as tp

import torch

from .distributed import GradReducer, world


class Trainer:
    def __init__(self, max_epochs: int, callbacks: tp.Sequence[tp.Any] = (), limit_train_batches: int | None = None,
                 device: str | torch.device = "cuda", bucket_bytes: int = 512 << 20, reduce_gradients: bool | None = None) -> None:
        if max_epochs < 1:
            raise ValueError("max_epochs must be >= 1")
        self.max_epochs, self.callbacks, self.limit_train_batches = max_epochs, list(callbacks), limit_train_batches
        self.device, self.bucket_bytes = device, bucket_bytes
        self.reduce_gradients = reduce_gradients                # None: whenever the process group has more than one rank
        self.current_epoch = 0
        self.global_step = 0
        self.optimizers: list[torch.optim.Optimizer] = []
        self.lr_scheduler: dict[str, tp.Any] | None = None     # {"scheduler", "interval"}: callbacks (SWA) may replace it
        self.train_dataloader: tp.Any = None
        self.reducer: GradReducer | None = None
        self.history: list[dict[str, float]] = []

    def _call(self, hook: str, module: tp.Any) -> None:
        for cb in self.callbacks:
            fn = getattr(cb, hook, None)
            if fn is not None:
                fn(self, module)

    def _n_batches(self, loader: tp.Any) -> int:
        n = len(loader)
        return n if self.limit_train_batches is None else min(n, self.limit_train_batches)

    @property
    def estimated_stepping_batches(self) -> int:
        return self._n_batches(self.train_dataloader) * self.max_epochs

    def fit(self, module: tp.Any, train_loader: tp.Any, val_loader: tp.Any = None) -> None:
        self.train_dataloader = train_loader
        module.to(self.device)
        module.trainer = self
        built = module.configure_optimizers(total_steps=self.estimated_stepping_batches)
        if isinstance(built, dict):
            self.optimizers, self.lr_scheduler = [built["optimizer"]], built.get("lr_scheduler")
        else:
            self.optimizers, self.lr_scheduler = [built], None
        opt = self.optimizers[0]
        if (world()[1] > 1) if self.reduce_gradients is None else self.reduce_gradients:
            self.reducer = GradReducer([p for g in opt.param_groups for p in g["params"]], bucket_bytes=self.bucket_bytes)
        self._call("on_fit_start", module)
        for epoch in range(self.max_epochs):
            self.current_epoch = epoch
            module.train()
            self._call("on_train_epoch_start", module)
            total, count = 0.0, 0
            for i, batch in enumerate(train_loader):
                if i >= self._n_batches(train_loader):
                    break
                if self.reducer is not None:
                    self.reducer.zero_grad()
                else:
                    opt.zero_grad(set_to_none=True)
                loss = module.training_step(batch.to(self.device), i)
                loss.backward()
                if self.reducer is not None:
                    self.reducer.finish()
                opt.step()
                if self.lr_scheduler is not None and self.lr_scheduler["interval"] == "step":
                    self.lr_scheduler["scheduler"].step()
                self.global_step += 1
                total, count = total + float(loss.detach()), count + 1
            if self.lr_scheduler is not None and self.lr_scheduler["interval"] == "epoch":
                self.lr_scheduler["scheduler"].step()
            self._call("on_train_epoch_end", module)
            record = {"epoch": epoch, "train/loss": total / max(1, count), "lr": opt.param_groups[0]["lr"]}
            if val_loader is not None:
                module.eval()
                with torch.no_grad():
                    for i, batch in enumerate(val_loader):
                        module.validation_step(batch.to(self.device), i)
                module.on_validation_epoch_end()
            self.history.append(record)
        self.current_epoch = self.max_epochs                   # Lightning leaves the counter one past the last epoch
        self._call("on_train_end", module)
        if self.reducer is not None:
            self.reducer.remove()
            self.reducer = None
