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


def test_eight_rank_rehearsal():
    """The driver's 8-GPU scaling run must not also be the first 8-process run: same launcher, rendezvous and exchange with eight
    ranks over gloo on the host (each rank imports torch: ~8 x 0.5 GiB, fine on the 64-GiB build container)."""
    p, lines = _run(["--gpus", "8", "--backend", "gloo", "--launcher-rehearsal", "--steps", "2", "--warmup", "1"], 600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["exchange_ok"] is True and out["config"]["parallelism"] == "dp8"


def test_launcher_kills_ranks_that_ignore_sigterm(monkeypatch):
    """A rank stuck in a collective ignores SIGTERM: after the first failure the parent gives the others 30 s (1 s here), kills them and
    still returns the recorded non-zero code instead of hanging."""
    import importlib
    import time as _time

    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    script = ("import os, signal, sys, time\n"
              "signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
              "sys.exit(7) if os.environ['RANK'] == '0' else time.sleep(600)\n")

    class FakePopen(subprocess.Popen):
        def __init__(self, cmd, env=None, **kw):
            super().__init__([sys.executable, "-c", script], env=env, **kw)

    monkeypatch.setattr(bench.subprocess, "Popen", FakePopen)
    real_monotonic = _time.monotonic
    monkeypatch.setattr(bench.time, "monotonic", lambda: real_monotonic() * 30.0)   # the 30-s grace period passes in one
    t0 = real_monotonic()
    rc = bench.launch_ranks(2, [])
    assert rc == 7 and real_monotonic() - t0 < 60


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
