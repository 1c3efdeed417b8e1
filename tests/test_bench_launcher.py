"""`python bench.py --gpus N` invoked plainly must start its own rank processes (one per GPU, the reference's
main.py:388-395 layout), print ONE JSON line from rank 0 and propagate failures.

CPU: the launcher is exercised with `--launcher-rehearsal` (stand-in step on host tensors: rendezvous, bf16 exchange,
reporting -- no kernel, value null).  GPU: the real two-rank run over gloo on the box's one GPU."""

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(args, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def test_plain_invocation_launches_two_ranks_and_prints_one_line():
    p, lines = _run(["--gpus", "2", "--backend", "gloo", "--launcher-rehearsal", "--steps", "3", "--warmup", "1"], 300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["exchange_ok"] is True
    assert out["value"] is None and "rehearsal" in out["data"]      # nothing measured, and the line says so
    assert out["config"]["parallelism"] == "dp2"


def test_rank_failure_propagates_nonzero_exit():
    # world size mismatch inside the children: every rank exits non-zero, the parent must not report success
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    code = ("import sys; sys.argv=['bench.py','--gpus','2','--launcher-rehearsal']; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.launch_ranks(2, ['--gpus', '3', '--launcher-rehearsal']))" % str(ROOT))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert "exited with code" in p.stderr


@pytest.mark.gpu
def test_two_rank_bench_over_gloo_on_one_gpu():
    """The real rank body: two fresh processes sharing the box's GPU, gloo exchange of bf16 predictions."""
    p, lines = _run(["--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--repeats", "1"], 900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["global_batch"] == 8
    assert "gloo" in out["config"]["workload"] and "RCCL" not in out["config"]["workload"]
    assert out["roofline"]["by_kernel"]["voxel_head"]["launches"] == 2
