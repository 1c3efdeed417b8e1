"""Replica scheduling for grids / ensembles on one 8-GPU node.

The reference fans a grid out as a slurm job array through exca (`run_grid`, /root/reference/modeling_utils/modeling_utils/utils.py:76-160;
`grids/run_grid.py`: layers x 5 seeds, `grids/run_ensemble.py`: up to 1000 sampled configurations) and later averages the runs'
submissions (`grids/average_submissions.py`).  The cluster plumbing is out of this build's scope; what maps onto one node is
the part below: the same grid expansion, one process per GPU taking every world_size-th configuration (replicas are
independent: no collective until the end), and the gather of their results on rank 0 -- BASELINE config 5
("N seeds x 4 subjects data-parallel over 8 GPUs").
"""

from __future__ import annotations

import hashlib
import itertools
import json
import random
import typing as tp


def expand_grid(grid: dict[str, list], combinatorial: bool = False, n_randomly_sampled: int | None = None,
                rng: random.Random | None = None) -> list[dict[str, tp.Any]]:
    """utils.py:104-117: the cartesian product of the value lists (combinatorial) or one configuration per single value, then an
    optional random sample without replacement (the reference draws from the global `random`; pass `rng` to make it repeatable)."""
    assert all(isinstance(v, list) for v in grid.values()), "Grid values must be lists."
    if combinatorial:
        configs = [dict(zip(grid.keys(), v)) for v in itertools.product(*grid.values())]
    else:
        configs = [{param: value} for param, values in grid.items() for value in values]
    if n_randomly_sampled is not None:
        assert n_randomly_sampled <= len(configs), "n_randomly_sampled must be less than the number of grid products"
        configs = (rng or random).sample(configs, n_randomly_sampled)
    return configs


def replica_name(params: dict[str, tp.Any]) -> str:
    """A stable folder / job name for a configuration (the reference uses exca's ConfDict.to_uid; any injective, order-free
    naming serves the same purpose here)."""
    body = json.dumps(params, sort_keys=True, default=str)
    return hashlib.sha1(body.encode()).hexdigest()[:12]


def apply_overrides(base_config: dict[str, tp.Any], params: dict[str, tp.Any]) -> dict[str, tp.Any]:
    """ConfDict.update with dotted keys ("data.layers": [...]) on a plain nested dict; returns a deep copy."""
    out = json.loads(json.dumps(base_config, default=str)) if base_config else {}
    for key, value in params.items():
        node = out
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = value
    return out


def my_replicas(n_configs: int, rank: int, world_size: int) -> list[int]:
    """Round-robin assignment: replica i runs on rank i % world_size."""
    return list(range(rank, n_configs, world_size))


def run_replicas(fn: tp.Callable[[dict[str, tp.Any]], tp.Any], configs: list[dict[str, tp.Any]]) -> list[tp.Any] | None:
    """Run `fn(config)` for this rank's share of `configs` and gather every result on rank 0 (returned in configuration order;
    None on the other ranks).  Works without torch.distributed (single process: everything runs here)."""
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    mine = {i: fn(configs[i]) for i in my_replicas(len(configs), rank, world)}
    if world == 1:
        return [mine[i] for i in range(len(configs))]
    gathered: list[tp.Any] = [None] * world if rank == 0 else []
    dist.gather_object(mine, gathered if rank == 0 else None, dst=0)
    if rank != 0:
        return None
    merged: dict[int, tp.Any] = {}
    for part in gathered:
        merged.update(part)
    return [merged[i] for i in range(len(configs))]
