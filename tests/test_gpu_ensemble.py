"""GPU legs of the steps after the model (csrc/features.hip through the C ABI): submission rows, ensemble averaging and
the correlation matrix behind the diverse-subset choice, against golden vectors from the reference's own callbacks.py /
average_submissions.py (g12, g13).  Bar: bit-exact for the averaged arrays (numpy's summation order and rounding is
reproduced), 1e-12 for the f64 correlation matrix (atomics make its summation order free)."""

import types
import zipfile
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def g13():
    return np.load(GOLD / "g13_ensemble.npz")


def runs_of(g13):
    runs = []
    for r in range(5):
        sub: dict = {}
        for k in g13.files:
            if k.startswith(f"run{r}__"):
                _, s, c = k.split("__")
                sub.setdefault(s, {})[c] = g13[k]
        runs.append(sub)
    return runs


def check(avg, g13, name):
    n = 0
    for k in g13.files:
        if k.startswith(f"{name}__"):
            _, s, c = k.split("__")
            assert avg[s][c].dtype == g13[k].dtype and np.array_equal(avg[s][c], g13[k]), k
            n += 1
    assert n == 4


def test_transpose_and_submission_writer_on_gpu(tmp_path):
    from tests.test_segments_host import build_events  # noqa: F401  (same events as the CPU test)
    from algonauts2025.callbacks import Benchmark
    from data_utils.segments import list_segments
    from tribe_hip import ops

    x = torch.randn(3, 1000, 77, device="cuda")
    assert torch.equal(ops.transpose_f32(x), x.transpose(1, 2).contiguous())
    g12 = np.load(GOLD / "g12_segments.npz")
    segments = list_segments(build_events(g12))
    samples = {}
    for k in g12.files:
        if k.startswith("bench_samples__"):
            _, subject, chunk = k.split("__")
            samples.setdefault(subject, {})[chunk] = int(g12[k])
    bm = Benchmark(target_sample_number=samples)
    trainer = types.SimpleNamespace(logger=types.SimpleNamespace(save_dir=str(tmp_path)))
    bm.on_test_epoch_start(trainer, None)
    preds = torch.from_numpy(g12["bench_preds"]).cuda()
    edges = g12["bench_batches"]
    for b0, b1 in zip(edges[:-1], edges[1:]):
        bm.on_test_batch_end(trainer, None, (preds[b0:b1], None), types.SimpleNamespace(segments=segments[b0:b1]), 0)
    bm.on_test_epoch_end(trainer, None)
    for k in g12.files:
        if k.startswith("bench_result__"):
            _, subject, chunk = k.split("__")
            assert np.array_equal(bm.submission_dict[subject][chunk], g12[k])


def test_average_predictions_bit_exact(g13):
    from algonauts2025.grids.average_submissions import average_predictions, ensemble_weights

    runs = runs_of(g13)
    check(average_predictions(runs), g13, "mean")
    w = ensemble_weights(g13["scores_as_read"], None, False, 0.3)
    check(average_predictions(runs, w, weigh_by_score=True), g13, "score")
    wv = ensemble_weights(None, list(g13["pearsons"]), True, 0.3)
    check(average_predictions(runs, wv, weigh_by_score=True), g13, "voxel")
    check(average_predictions(runs[:3]), g13, "first3")


def test_average_submissions_folder_protocol(g13, tmp_path):
    import pandas as pd

    from algonauts2025.grids.average_submissions import average_submissions

    for r, sub in enumerate(runs_of(g13)):
        run = tmp_path / f"run{r:02d}"
        run.mkdir()
        np.save(run / "submission.npy", sub)
        with zipfile.ZipFile(run / "submission.zip", "w") as z:
            z.write(run / "submission.npy", arcname="submission.npy")
        pd.DataFrame({"val/pearson": [g13["scores"][r]]}).to_csv(run / "metrics.csv", index=False)
        np.save(run / "pearson.npy", g13["pearsons"][r])
    with pytest.raises(ValueError):
        average_submissions(tmp_path)                                        # pickled inputs need the explicit opt-in
    check(average_submissions(tmp_path, trust_pickle=True), g13, "mean")
    assert (tmp_path / "submission.zip").exists() and not (tmp_path / "run00" / "submission.npy").exists()
    check(average_submissions(tmp_path, weigh_by_score=True, temperature=0.3, trust_pickle=True), g13, "score")
    check(average_submissions(tmp_path, weigh_by_score=True, per_voxel_weights=True, temperature=0.3, trust_pickle=True), g13, "voxel")
    check(average_submissions(tmp_path, max_runs=3, trust_pickle=True), g13, "first3")
    check(average_submissions(tmp_path, weigh_by_score=True, temperature=0.5, k_most_diverse=3, trust_pickle=True), g13, "diverse3_score")


@pytest.mark.parametrize("N,K", [(7, 100003), (64, 4096), (2, 50)])
def test_corr_matrix_vs_numpy(N, K):
    from tribe_hip import ops

    rng = np.random.default_rng(N)
    x = (rng.standard_normal((N, K)) + 0.7 * rng.standard_normal((1, K)) + 3.0).astype(np.float32)
    got = ops.corr_matrix(torch.from_numpy(x).cuda()).cpu().numpy()
    want = np.corrcoef(x)
    assert np.abs(got - want).max() < 1e-12
