"""Golden vectors for the steps either side of the model: window segmentation, the
per-epoch window jitter and the submission writer.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_segments.py

EXECUTES the reference's own `data_utils` package files (`base.py`, `utils.py`, `events.py`, `helpers.py`,
`segments.py`: numpy / pandas / pydantic only) through a package shell whose `__path__` points at the reference
directory -- the package `__init__` is skipped because it pulls in the study loaders -- and the two callbacks of
`algonauts2025/callbacks.py`, whose methods are compiled out of the file with `ast` (the module imports `lightning`,
absent here).  `Path.home` is redirected into this repository's scratch directory while base.py loads (it creates
`~/.cache/data_utils`).  Only Word / Fmri / plain events are used: Sound / Video events insist on existing media files.

Writes g12_segments.npz (inputs + expected windows / selections / submission arrays).
"""

from __future__ import annotations

import ast
import importlib
import sys
import tempfile
import types
from pathlib import Path
from unittest import mock

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REF = Path("/root/reference")


def load_reference_data_utils():
    scratch = ROOT / "gpurun_out" / "_home"
    scratch.mkdir(parents=True, exist_ok=True)
    shell = types.ModuleType("data_utils")
    shell.__path__ = [str(REF / "data_utils" / "data_utils")]
    sys.modules["data_utils"] = shell
    with mock.patch("pathlib.Path.home", return_value=scratch):
        seg = importlib.import_module("data_utils.segments")
        importlib.import_module("data_utils.helpers")
    return seg, sys.modules["data_utils.events"]


class _IntOverlap(ast.NodeTransformer):
    """`overlap_trs = 0.0` -> `overlap_trs = 0` (see main(): the float makes `pred[overlap_trs:]` a TypeError)."""

    def visit_Assign(self, node: ast.Assign) -> ast.AST:
        if len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) and node.targets[0].id == "overlap_trs":
            node.value = ast.Constant(0)
        return node


def extract_methods(path: Path, cls: str, names: list[str], ns: dict, int_overlap: bool = False) -> dict:
    tree = ast.parse(path.read_text())
    if int_overlap:
        tree = _IntOverlap().visit(tree)
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name in names:
                    m = ast.Module(body=[sub], type_ignores=[])
                    scope = dict(ns)
                    exec(compile(ast.fix_missing_locations(m), str(path), "exec"), scope)
                    out[sub.name] = scope[sub.name]
    return out


def main() -> None:
    seg_mod, ev_mod = load_reference_data_utils()
    rng = np.random.default_rng(12)
    out: dict[str, np.ndarray] = {}

    # ---- events of two timelines (movie chunks), unsorted on purpose --------------------------------------------
    events, rows = [], []
    for ti, (tl, n_words, t0, length) in enumerate([("friends:s01e01a", 70, 0.0, 700.0), ("friends:s01e01b", 25, 12.5, 310.0)]):
        rec = ev_mod.Fmri(start=t0, duration=length, timeline=tl, filepath=f"h5:{tl}", frequency=1 / 1.49, subject=f"algonauts/sub-0{ti + 1}",
                          extra={"chunk": f"chunk:e01{'ab'[ti]}"})
        events.append(rec)
        rows.append((ti, 0, t0, length))
        ws = np.sort(rng.uniform(t0, t0 + length, n_words)).round(2)
        wd = rng.uniform(0.0, 1.0, n_words).round(2)
        for s, d in zip(ws, wd):
            # the study loaders stamp every row of a timeline with its subject / chunk (read by callbacks.py:61-62)
            events.append(ev_mod.Word(start=float(s), duration=float(d), timeline=tl, text="w",
                                      extra={"subject": f"algonauts/sub-0{ti + 1}", "chunk": f"chunk:e01{'ab'[ti]}"}))
            rows.append((ti, 1, float(s), float(d)))
    order = rng.permutation(len(events))
    events = [events[i] for i in order]
    out["events"] = np.asarray([rows[i] for i in order], dtype=np.float64)       # columns: timeline id, kind (0 Fmri, 1 Word), start, duration

    def dump(prefix: str, segments) -> None:
        ids = {id(e): i for i, e in enumerate(events)}
        out[f"{prefix}_start"] = np.asarray([s.start for s in segments])
        out[f"{prefix}_duration"] = np.asarray([s.duration for s in segments])
        out[f"{prefix}_trigger"] = np.asarray([s._trigger for s in segments], dtype=np.float64)
        out[f"{prefix}_count"] = np.asarray([len(s.ns_events) for s in segments])
        out[f"{prefix}_members"] = np.asarray([ids[id(e)] for s in segments for e in s.ns_events], dtype=np.int64)

    segments = seg_mod.list_segments(events)
    dump("windows", segments)

    # strided windows helper on its own
    for i, (a, b, st, du, drop) in enumerate([(0.0, 10.0, 2.5, 5.0, True), (-4.47, 700.0 - 4.47, 149.0, 149.0, False), (3.0, 3.0, 1.0, 1.0, False)]):
        s, d = seg_mod._prepare_strided_windows(a, b, st, du, drop_incomplete=drop)
        out[f"strided{i}_args"] = np.asarray([a, b, st, du, float(drop)])
        out[f"strided{i}_starts"], out[f"strided{i}_durations"] = s, d

    # subsegment of the first window
    sub = segments[0].subsegment(20.0, 30.5)
    dump("subsegment", [sub])

    # ---- JitterWindows.on_train_epoch_start (callbacks.py:25-44) ------------------------------------------------
    cb = extract_methods(REF / "algonauts2025/callbacks.py", "JitterWindows", ["on_train_epoch_start"],
                         {"np": np, "SegmentCreator": seg_mod.SegmentCreator, "_prepare_strided_windows": seg_mod._prepare_strided_windows})
    me = types.SimpleNamespace(start_jitter_amount=10.0, duration_jitter_amount=0.0)
    trainer = types.SimpleNamespace(train_dataloader=types.SimpleNamespace(dataset=types.SimpleNamespace(segments=list(segments))))
    np.random.seed(5)
    cb["on_train_epoch_start"](me, trainer, None)
    dump("jitter", trainer.train_dataloader.dataset.segments)
    out["jitter_seed_amount"] = np.asarray([5.0, 10.0])

    # ---- Benchmark submission writer (callbacks.py:47-103) --------------------------------------------------------
    V = 6
    bm = extract_methods(REF / "algonauts2025/callbacks.py", "Benchmark", ["on_test_epoch_start", "on_test_batch_end", "on_test_epoch_end"],
                         {"np": np, "Path": Path})
    with tempfile.TemporaryDirectory(dir=ROOT / "gpurun_out") as tmp:
        tmp = Path(tmp)
        want_samples = {}
        for ti in range(2):
            subject = f"sub-0{ti + 1}"
            n_seg = sum(1 for s in segments if s.ns_events and s.ns_events[0].timeline.endswith("ab"[ti]))
            want_samples[subject] = {f"s07e01{'ab'[ti]}": 100 * n_seg - 37}
            d = tmp / f"algonauts_2025.competitors/fmri/{subject}/target_sample_number"
            d.mkdir(parents=True)
            np.save(d / f"{subject}_friends-s7_fmri_samples.npy", want_samples[subject])
        me = types.SimpleNamespace(root_data_dir=tmp, submission_dict={})
        trainer = types.SimpleNamespace(logger=types.SimpleNamespace(save_dir=str(tmp)))
        preds = rng.standard_normal((len(segments), V, 100)).astype(np.float32)
        out["bench_preds"] = preds
        # As shipped, callbacks.py:56 sets `overlap_trs = 0.0`; with numpy >= 1.12 the slice `pred[overlap_trs:]` taken for
        # every window after the first of a chunk (callbacks.py:73) raises TypeError.  Record that, then produce the
        # expected arrays with that ONE literal turned into the integer 0 in the syntax tree (the evident intent: no
        # overlap between consecutive windows, stride == duration); nothing else of the reference's code is altered.
        bm["on_test_epoch_start"](me, trainer, None)
        try:
            bm["on_test_batch_end"](me, trainer, None, (torch.from_numpy(preds[:3]), None), types.SimpleNamespace(segments=segments[:3]), 0)
            out["bench_reference_raises_typeerror"] = np.asarray(False)
        except TypeError:
            out["bench_reference_raises_typeerror"] = np.asarray(True)
        bm = extract_methods(REF / "algonauts2025/callbacks.py", "Benchmark", ["on_test_epoch_start", "on_test_batch_end", "on_test_epoch_end"],
                             {"np": np, "Path": Path}, int_overlap=True)
        me.submission_dict = {}
        bm["on_test_epoch_start"](me, trainer, None)
        out["bench_batches"] = np.asarray([0, 3, 4, len(segments)])
        for b0, b1 in zip(out["bench_batches"][:-1], out["bench_batches"][1:]):
            batch = types.SimpleNamespace(segments=segments[b0:b1])
            bm["on_test_batch_end"](me, trainer, None, (torch.from_numpy(preds[b0:b1]), None), batch, 0)
        bm["on_test_epoch_end"](me, trainer, None)
        for subject, chunks in me.submission_dict.items():
            for chunk, arr in chunks.items():
                out[f"bench_result__{subject}__{chunk}"] = np.asarray(arr)
                out[f"bench_samples__{subject}__{chunk}"] = np.asarray(want_samples[subject][chunk])
        assert (tmp / "submission.zip").exists()

    np.savez_compressed(HERE / "g12_segments.npz", **out)
    print("wrote g12_segments.npz:", len(segments), "windows,", len(out), "arrays")


if __name__ == "__main__":
    main()
