"""Golden vectors for ensemble averaging.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_ensemble.py

EXECUTES the three functions of the reference's `algonauts2025/grids/average_submissions.py` (compiled out of the file
with `ast`: the module's relative import of `.defaults` needs the whole experiment stack) on a small synthetic grid
of runs written by this script (`<run>/submission.zip`, `metrics.csv`, `pearson.npy`).  Two names in the function's
namespace are replaced by order-preserving stand-ins so that the result is reproducible: `os.listdir` returns sorted
names, and the thread pool runs its jobs in submission order (the reference's own order is whatever the file system
and thread timing give; the code under test is untouched).

Writes g13_ensemble.npz: the runs' predictions / scores / per-voxel Pearson and, per mode, the averaged arrays.
"""

from __future__ import annotations

import ast
import os
import tempfile
import types
import zipfile
from collections import defaultdict
from pathlib import Path

import numpy as np
import pandas as pd
import torch
from tqdm import tqdm

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REF = Path("/root/reference")


class _SerialFuture:
    def __init__(self, value):
        self._value = value

    def result(self):
        return self._value


class _SerialPool:
    def __init__(self, max_workers=None):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def submit(self, fn, *args):
        return _SerialFuture(fn(*args))


def _as_completed(futures):
    return list(futures)


_os = types.SimpleNamespace(listdir=lambda p: sorted(os.listdir(p)), path=os.path, remove=os.remove)


def load_functions() -> dict:
    tree = ast.parse((REF / "algonauts2025/grids/average_submissions.py").read_text())
    ns = {"os": _os, "zipfile": zipfile, "defaultdict": defaultdict, "ThreadPoolExecutor": _SerialPool, "as_completed": _as_completed,
          "Path": Path, "np": np, "pd": pd, "tqdm": tqdm, "torch": torch}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef)]
    exec(compile(ast.fix_missing_locations(ast.Module(body=body, type_ignores=[])), "average_submissions.py", "exec"), ns)
    return ns


def main() -> None:
    fns = load_functions()
    rng = np.random.default_rng(13)
    n_runs, V = 5, 10
    chunks = {"sub-01": {"s07e01a": 23, "s07e01b": 17}, "sub-02": {"s07e01a": 23, "s07e01b": 17}}
    base = {s: {c: rng.standard_normal((t, V)).astype(np.float32) for c, t in cs.items()} for s, cs in chunks.items()}
    out: dict[str, np.ndarray] = {}
    scores = rng.uniform(0.18, 0.24, n_runs)
    pearsons = rng.uniform(0.0, 0.5, (n_runs, V)).astype(np.float32)
    out["scores"], out["pearsons"] = scores, pearsons
    with tempfile.TemporaryDirectory(dir=ROOT / "gpurun_out") as tmp:
        grid = Path(tmp)
        for r in range(n_runs):
            run = grid / f"run{r:02d}"
            run.mkdir()
            sub = {s: {c: (a + (0.3 + 0.2 * r) * rng.standard_normal(a.shape)).astype(np.float32) for c, a in cs.items()} for s, cs in base.items()}
            for s, cs in sub.items():
                for c, a in cs.items():
                    out[f"run{r}__{s}__{c}"] = a
            np.save(run / "submission.npy", sub)
            with zipfile.ZipFile(run / "submission.zip", "w") as z:
                z.write(run / "submission.npy", arcname="submission.npy")
            pd.DataFrame({"val/pearson": [scores[r]]}).to_csv(run / "metrics.csv", index=False)
            np.save(run / "pearson.npy", pearsons[r])
            # what the reference will see: pandas' default float parser may land one ulp off the written value
            out.setdefault("scores_as_read", np.zeros(n_runs))[r] = pd.read_csv(run / "metrics.csv")["val/pearson"].item()
        (grid / "not_a_run.txt").write_text("x")
        modes = {"mean": dict(), "score": dict(weigh_by_score=True, temperature=0.3),
                 "voxel": dict(weigh_by_score=True, per_voxel_weights=True, temperature=0.3), "first3": dict(max_runs=3),
                 "diverse3_score": dict(weigh_by_score=True, temperature=0.5, k_most_diverse=3)}
        for name, kw in modes.items():
            fns["average_submissions"](grid, **kw)
            got = np.load(grid / "submission.npy", allow_pickle=True).item()     # written one line above by this process
            for s, cs in got.items():
                for c, a in cs.items():
                    out[f"{name}__{s}__{c}"] = np.asarray(a)
            os.remove(grid / "submission.npy")
            os.remove(grid / "submission.zip")
    # the subset choice on its own
    C = np.corrcoef(rng.standard_normal((7, 400)) + 0.5 * rng.standard_normal((1, 400)))
    out["diverse_C"] = C
    out["diverse_k4"] = np.asarray(fns["select_diverse_subset"](C, 4))
    np.savez_compressed(HERE / "g13_ensemble.npz", **out)
    print("wrote g13_ensemble.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
