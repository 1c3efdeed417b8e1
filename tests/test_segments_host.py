"""Window segmentation, window jitter, submission writer and ensemble weights on CPU, against golden vectors produced by
the reference's own segments.py / callbacks.py / average_submissions.py (tests/golden/make_golden_segments.py,
make_golden_ensemble.py -> g12, g13)."""

import types
from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def g12():
    return np.load(GOLD / "g12_segments.npz")


@pytest.fixture(scope="module")
def g13():
    return np.load(GOLD / "g13_ensemble.npz")


def build_events(g12):
    from data_utils.events import Fmri, Word

    names = ["friends:s01e01a", "friends:s01e01b"]
    events = []
    for ti, kind, start, dur in g12["events"]:
        ti = int(ti)
        extra = {"subject": f"algonauts/sub-0{ti + 1}", "chunk": f"chunk:e01{'ab'[ti]}"}
        if kind == 0:
            events.append(Fmri(start=start, duration=dur, timeline=names[ti], filepath=f"h5:{names[ti]}", frequency=1 / 1.49,
                               subject=extra["subject"], extra={"chunk": extra["chunk"]}))
        else:
            events.append(Word(start=start, duration=dur, timeline=names[ti], text="w", extra=extra))
    return events


def check_windows(g12, prefix, segments, events):
    ids = {id(e): i for i, e in enumerate(events)}
    assert np.array_equal(np.asarray([s.start for s in segments]), g12[f"{prefix}_start"])
    assert np.array_equal(np.asarray([s.duration for s in segments]), g12[f"{prefix}_duration"])
    assert np.array_equal(np.asarray([s._trigger for s in segments], dtype=np.float64), g12[f"{prefix}_trigger"], equal_nan=True)
    assert np.array_equal(np.asarray([len(s.ns_events) for s in segments]), g12[f"{prefix}_count"])
    assert np.array_equal(np.asarray([ids[id(e)] for s in segments for e in s.ns_events]), g12[f"{prefix}_members"])


def test_windows_match_reference(g12):
    from data_utils.segments import _prepare_strided_windows, list_segments

    events = build_events(g12)
    segments = list_segments(events)
    check_windows(g12, "windows", segments, events)
    for i in range(3):
        a, b, st, du, drop = g12[f"strided{i}_args"]
        s, d = _prepare_strided_windows(a, b, st, du, drop_incomplete=bool(drop))
        assert np.array_equal(s, g12[f"strided{i}_starts"]) and np.array_equal(d, g12[f"strided{i}_durations"])
    sub = segments[0].subsegment(20.0, 30.5)
    sub._trigger = segments[0]._trigger
    check_windows(g12, "subsegment", [sub], events)
    with pytest.raises(ValueError):
        from data_utils.segments import SegmentCreator
        SegmentCreator(events)                                  # two timelines in one creator


def test_jitter_windows_match_reference(g12):
    from algonauts2025.callbacks import JitterWindows
    from data_utils.segments import list_segments

    events = build_events(g12)
    dataset = types.SimpleNamespace(segments=list_segments(events))
    trainer = types.SimpleNamespace(train_dataloader=types.SimpleNamespace(dataset=dataset))
    seed, amount = g12["jitter_seed_amount"]
    np.random.seed(int(seed))
    JitterWindows(start_jitter_amount=float(amount)).on_train_epoch_start(trainer, None)
    check_windows(g12, "jitter", dataset.segments, events)


def test_submission_writer_matches_reference(g12, tmp_path):
    from algonauts2025.callbacks import Benchmark
    from data_utils.segments import list_segments

    assert bool(g12["bench_reference_raises_typeerror"])         # the shipped float `overlap_trs` cannot slice (see callbacks.py of this build)
    events = build_events(g12)
    segments = list_segments(events)
    samples = {}
    for k in g12.files:
        if k.startswith("bench_samples__"):
            _, subject, chunk = k.split("__")
            samples.setdefault(subject, {})[chunk] = int(g12[k])
    bm = Benchmark(target_sample_number=samples)
    trainer = types.SimpleNamespace(logger=types.SimpleNamespace(save_dir=str(tmp_path)))
    bm.on_test_epoch_start(trainer, None)
    preds = torch.from_numpy(g12["bench_preds"])
    edges = g12["bench_batches"]
    for b0, b1 in zip(edges[:-1], edges[1:]):
        bm.on_test_batch_end(trainer, None, (preds[b0:b1], None), types.SimpleNamespace(segments=segments[b0:b1]), 0)
    bm.on_test_epoch_end(trainer, None)
    n = 0
    for k in g12.files:
        if k.startswith("bench_result__"):
            _, subject, chunk = k.split("__")
            assert np.array_equal(bm.submission_dict[subject][chunk], g12[k])
            n += 1
    assert n == 2 and (tmp_path / "submission.zip").exists()
    saved = np.load(tmp_path / "submission.zip", allow_pickle=True)["submission"].item()      # file written by this test
    assert set(saved) == set(bm.submission_dict)
    bad = Benchmark(target_sample_number={s: {c: 10**6 for c in v} for s, v in samples.items()})
    bad.submission_dict = {s: {c: [a] for c, a in v.items()} for s, v in bm.submission_dict.items()}
    with pytest.raises(ValueError):
        bad.on_test_epoch_end(trainer, None)                     # fewer predictions than the competition expects
    with pytest.raises(ValueError):
        Benchmark(root_data_dir=tmp_path)._samples("sub-01")     # pickled samples file is refused without trust_pickle


def test_diverse_subset_and_weights_match_reference(g13):
    from algonauts2025.grids.average_submissions import ensemble_weights, select_diverse_subset

    assert select_diverse_subset(g13["diverse_C"], 4) == g13["diverse_k4"].tolist()
    w = ensemble_weights(g13["scores_as_read"], None, per_voxel_weights=False, temperature=0.3)
    assert w.shape == (5, 1, 1) and w.dtype == np.float64 and abs(w.sum() - 1) < 1e-12
    wv = ensemble_weights(None, list(g13["pearsons"]), per_voxel_weights=True, temperature=0.3)
    assert wv.shape == (5, 1, 10) and wv.dtype == np.float32
    assert np.allclose(wv.sum(axis=2), 1.0, atol=1e-6)           # the reference's softmax runs over voxels (dim=1), kept
    # the golden "score" average is reproduced on the host from these weights (numpy order), which pins the weight formula
    runs = [g13[f"run{r}__sub-01__s07e01a"] for r in range(5)]
    assert np.array_equal(np.sum(np.array(runs) * w, axis=0), g13["score__sub-01__s07e01a"])
    assert np.array_equal(np.sum(np.array(runs) * wv, axis=0), g13["voxel__sub-01__s07e01a"])
