"""CPU, world_size 2, gloo: the N > 1 host path (sequence sharding, all-gather of predictions, all-reduce of the
Pearson sufficient statistics) gives exactly the single-process result.  The arithmetic that runs on the GPU in
production (predictions, statistics) is produced here by the CPU oracle -- only the distribution logic is under test."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tribe_ref


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, q):
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "algonauts-2025_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from algonauts2025 import distributed as D
        from modeling_utils.metrics.base import _PearsonState

        torch.manual_seed(0)
        B, V, T = 6, 11, 20
        g = torch.Generator().manual_seed(5)
        pred_all = torch.randn(B, V, T, generator=g)
        true_all = 0.3 * pred_all + torch.randn(B, V, T, generator=g)
        data = {"pred": pred_all, "fmri": true_all, "subject_id": (torch.arange(B) % 4).view(B, 1)}
        mine = D.shard_batch(data, rank, world)
        assert mine["pred"].shape[0] == B // world
        # (1) all-gather of predictions, restored to the original order
        buf, work = D.gather_predictions(mine["pred"], async_op=True)
        work.wait()
        restored = buf[torch.tensor(D.unshard_order(B, world))]
        ok_gather = torch.equal(restored, pred_all)
        # (2) all-reduce of the sufficient statistics == statistics of the whole data set
        local = tribe_ref.pearson_stats(tribe_ref.flatten_bt(mine["pred"]), tribe_ref.flatten_bt(mine["fmri"]))  # [5, V]
        stats = torch.zeros(1, V, 6, dtype=torch.float64)
        stats[0, :, :5] = local.t()
        stats[0, :, 5] = mine["pred"].shape[0] * T
        state = _PearsonState(V)
        state.stats = stats
        state.sync()
        full = tribe_ref.pearson_stats(tribe_ref.flatten_bt(pred_all), tribe_ref.flatten_bt(true_all))
        ok_stats = torch.allclose(state.stats[0, :, :5].t(), full, rtol=1e-12) and bool((state.stats[0, :, 5] == B * T).all())
        r = tribe_ref.pearson_from_stats(state.stats[0, :, :5].t(), B * T).numpy()
        ref = tribe_ref.scipy_pearson_columns(tribe_ref.flatten_bt(pred_all).numpy(), tribe_ref.flatten_bt(true_all).numpy())
        ok_r = bool(np.abs(r - ref).max() < 2e-6)
        q.put((rank, ok_gather, ok_stats, ok_r))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding_gather_and_stats():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == [0, 1]
    assert all(all(r[1:]) for r in results), results


def test_shard_indices_cover_and_partition():
    from algonauts2025 import distributed as D

    for n in (1, 4, 7, 16):
        for g in (1, 2, 3, 8):
            parts = [D.shard_indices(n, r, g) for r in range(g)]
            assert sorted(i for p in parts for i in p) == list(range(n))
    with pytest.raises(ValueError):
        D.shard_indices(4, 2, 2)
    inv = D.unshard_order(8, 4)
    order = [i for r in range(4) for i in D.shard_indices(8, r, 4)]
    assert [order[p] for p in inv] == list(range(8))


def _replica_worker(rank: int, world: int, port: int, q):
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "algonauts-2025_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from algonauts2025.grids.replicas import expand_grid, my_replicas, run_replicas

        configs = expand_grid({"seed": list(range(5)), "data.layers": [[0, 0.5, 1], [0.5, 1.0]]}, combinatorial=True)
        ran = []

        def fn(cfg):
            ran.append(cfg["seed"])
            return {"seed": cfg["seed"], "n_layers": len(cfg["data.layers"]), "rank": dist.get_rank()}

        out = run_replicas(fn, configs)
        q.put((rank, len(ran), my_replicas(len(configs), rank, world), out))
    finally:
        dist.destroy_process_group()


def test_replica_scheduling_two_ranks_and_grid_expansion():
    """BASELINE config 5 host logic: grid expansion as the reference's run_grid (utils.py:104-117), round-robin replicas over
    ranks, results gathered in configuration order on rank 0."""
    import itertools
    import random

    from algonauts2025.grids.replicas import apply_overrides, expand_grid, replica_name

    grid = {"a": [1, 2, 3], "b.c": ["x", "y"]}
    comb = expand_grid(grid, combinatorial=True)
    assert comb == [dict(zip(grid.keys(), v)) for v in itertools.product(*grid.values())] and len(comb) == 6
    assert expand_grid(grid) == [{"a": 1}, {"a": 2}, {"a": 3}, {"b.c": "x"}, {"b.c": "y"}]
    assert expand_grid(grid, combinatorial=True, n_randomly_sampled=4, rng=random.Random(3)) == random.Random(3).sample(comb, 4)
    with pytest.raises(AssertionError):
        expand_grid(grid, combinatorial=True, n_randomly_sampled=7)
    with pytest.raises(AssertionError):
        expand_grid({"a": 1})
    base = {"data": {"layers": [0.5], "other": 1}, "seed": 0}
    cfg = apply_overrides(base, {"data.layers": [0, 1], "loss.name": "PearsonLoss", "seed": 4})
    assert cfg == {"data": {"layers": [0, 1], "other": 1}, "seed": 4, "loss": {"name": "PearsonLoss"}} and base["seed"] == 0
    assert replica_name({"a": 1, "b": 2}) == replica_name({"b": 2, "a": 1}) != replica_name({"a": 1, "b": 3})

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_replica_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {r[0]: r for r in (q.get(timeout=180) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][1] == 5 and results[1][1] == 5                      # 10 replicas, 5 each
    assert results[0][2] == [0, 2, 4, 6, 8] and results[1][2] == [1, 3, 5, 7, 9]
    assert results[1][3] is None
    out = results[0][3]
    assert [o["seed"] for o in out] == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4] and [o["rank"] for o in out] == [0, 1] * 5
    assert [o["n_layers"] for o in out] == [3, 2] * 5


def _reducer_worker(rank: int, world: int, port: int, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from algonauts2025.distributed import GradReducer

        torch.manual_seed(0)                                     # same weights on every rank
        net = torch.nn.ModuleDict({"a": torch.nn.Linear(12, 16), "b": torch.nn.Linear(16, 16), "side": torch.nn.Linear(16, 16),
                                   "c": torch.nn.Linear(16, 3)})
        x = torch.randn(8, 12, generator=torch.Generator().manual_seed(1))
        y = torch.randn(8, 3, generator=torch.Generator().manual_seed(2))

        def loss_of(xb, yb, use_side):
            h = torch.relu(net["b"](torch.relu(net["a"](xb))))
            if use_side:
                h = h + net["side"](h)
            return ((net["c"](h) - yb) ** 2).mean()

        # single-process truth: mean over ranks of the per-rank losses (rank 1 skips the side branch, as modality dropout might)
        want = torch.autograd.grad((loss_of(x[0::2], y[0::2], True) + loss_of(x[1::2], y[1::2], False)) / 2, list(net.parameters()),
                                   allow_unused=True)
        reducer = GradReducer(net.parameters(), bucket_bytes=4 * 300)      # several small buckets, 'side' in the middle of one
        ok = len(reducer.buckets) >= 3
        for step in range(2):                                    # twice: zero_grad must restore the bucket views and the counters
            reducer.zero_grad()
            loss_of(x[rank::2], y[rank::2], use_side=(rank == 0)).backward()
            reducer.finish()
            for p, w in zip(net.parameters(), want):
                w = torch.zeros_like(p) if w is None else w
                ok = ok and bool(torch.allclose(p.grad, w, rtol=1e-5, atol=1e-7)) and p.grad.data_ptr() == reducer._view_of[id(p)].data_ptr()
        # an optimizer.zero_grad(set_to_none=True) in between is tolerated: the hook folds the fresh gradient back into its bucket
        reducer.zero_grad()
        for p in net.parameters():
            p.grad = None
        loss_of(x[rank::2], y[rank::2], use_side=(rank == 0)).backward()
        reducer.finish()
        side_w = net["side"].weight
        ok = ok and bool(torch.allclose(net["a"].weight.grad, want[0], rtol=1e-5, atol=1e-7))
        if rank == 1:                                            # never produced here, not re-attached either: the bucket slice carries rank 0's half
            ok = ok and side_w.grad is None and bool(torch.allclose(reducer._view_of[id(side_w)], want[4], rtol=1e-5, atol=1e-7))
        reducer.remove()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_two_ranks_matches_single_process_and_tolerates_unused_parameters():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)], results


def test_optimizer_configs_mirror_the_reference_block():
    """grids/defaults.py:126-141 `optim` block through modeling_utils.optimizers (base.py:25-96): validation and what gets built."""
    import pydantic

    from modeling_utils.optimizers import LightningOptimizerConfig, TorchOptimizerConfig

    block = {"optimizer": {"name": "Adam", "lr": 1e-4, "kwargs": {"weight_decay": 0.0}},
             "scheduler": {"name": "OneCycleLR", "kwargs": {"max_lr": 1e-4, "pct_start": 0.1}}}
    cfg = LightningOptimizerConfig(**block)
    assert cfg.name == "LightningOptimizer" and cfg.interval == "step"
    params = [torch.nn.Parameter(torch.zeros(4))]
    built = cfg.build(params, total_steps=50)
    assert isinstance(built["optimizer"], torch.optim.Adam)                 # CPU parameters: torch's own (HipAdam needs GPU tensors)
    sched = built["lr_scheduler"]
    assert isinstance(sched["scheduler"], torch.optim.lr_scheduler.OneCycleLR) and sched["interval"] == "step"
    assert sched["scheduler"].total_steps == 50
    assert "lr_scheduler" not in LightningOptimizerConfig(optimizer=block["optimizer"]).build(params)
    assert isinstance(TorchOptimizerConfig(name="SGD", lr=0.1, kwargs={"momentum": 0.9}).build(params), torch.optim.SGD)
    for bad in ({"optimizer": {"name": "Adam", "lr": 1e-4, "kwargs": {"lr": 1.0}}},
                {"optimizer": {"name": "NoSuchOptimizer", "lr": 1e-4}},
                {"optimizer": {"name": "Adam", "lr": 1e-4, "kwargs": {"not_an_argument": 1}}},
                {"optimizer": {"name": "Adam", "lr": 1e-4}, "scheduler": {"name": "OneCycleLR", "kwargs": {"pct_start": 0.1}}},   # max_lr missing
                {"optimizer": {"name": "Adam", "lr": 1e-4}, "extra_field": 1}):
        with pytest.raises(pydantic.ValidationError):
            LightningOptimizerConfig(**bad)


def _voxel_worker(rank: int, world: int, port: int, q):
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "algonauts-2025_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from algonauts2025 import distributed as D

        S, Cc, V, B, T = 3, 16, 13, 5, 9                      # 13 parcels over 2 ranks: blocks of 7 and 6
        g = torch.Generator().manual_seed(3)
        full = tribe_ref.SubjectLayersRef(Cc, V, S, bias=True)
        with torch.no_grad():
            full.weights.copy_(torch.randn(S, Cc, V, generator=g))
            full.bias.copy_(torch.randn(S, V, generator=g))
        x = torch.randn(B, Cc, T, generator=g)                # replicated encoder latents [B, C, T]
        subj = torch.tensor([[0], [2], [1], [1], [0]])
        want = full(x, subj)                                  # [B, V, T]
        sl = D.voxel_slice(V, rank, world)
        part = tribe_ref.SubjectLayersRef(Cc, sl.stop - sl.start, S, bias=True)   # the arithmetic the GPU head does on its block
        with torch.no_grad():
            part.weights.copy_(full.weights[:, :, sl])
            part.bias.copy_(full.bias[:, sl])
        got = D.gather_voxel_slabs(part(x, subj), V)
        q.put((rank, tuple(got.shape) == (B, V, T), bool(torch.equal(got, want)), (sl.start, sl.stop)))
    finally:
        dist.destroy_process_group()


def test_voxel_block_head_split_two_ranks():
    """north star's 'voxel-blocks shard across the GPUs with an all-gather of predictions': each rank computes its block of
    parcels from replicated latents, the slabs are gathered along V (uneven blocks padded for the collective)."""
    from algonauts2025 import distributed as D

    assert [(D.voxel_slice(1000, r, 8).start, D.voxel_slice(1000, r, 8).stop) for r in (0, 7)] == [(0, 125), (875, 1000)]
    assert [D.voxel_slice(5, r, 4).stop - D.voxel_slice(5, r, 4).start for r in range(4)] == [2, 2, 1, 0]
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_voxel_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[3] for r in results] == [(0, 7), (7, 13)]
    assert all(r[1] and r[2] for r in results), results
