"""Time-axis arithmetic of the feature pipeline on CPU: the oracle restatement (oracle/timeline_ref.py) and the product's
host mirror (data_utils/base.py) against golden vectors produced by the reference's own base.py
(tests/golden/make_golden_timeline.py -> g10, g11)."""

from pathlib import Path

import numpy as np
import pytest

from oracle import timeline_ref as tl

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def g10():
    z = np.load(GOLD / "g10_overlap_slices.npz")
    return z["table"], list(z["columns"])


@pytest.fixture(scope="module")
def g11():
    return np.load(GOLD / "g11_segment_assembly.npz")


def _layer_cfgs(g11):
    cfgs = []
    for row, gm in zip(g11["layer_cfg_layers"], g11["layer_cfg_group_mean"]):
        cfgs.append(([float(v) for v in row if v >= 0], "group_mean" if gm else None))
    return cfgs


def test_oracle_overlap_slice_matches_reference_table(g10):
    table, cols = g10
    assert cols[:6] == ["frequency", "arr_start", "arr_len", "arr_duration", "q_start", "q_duration"]
    n_valid = 0
    for f, a0, n, adur, q0, qd, valid, o0, od, first, count, raised in table:
        if raised:
            with pytest.raises(RuntimeError):
                tl.overlap_slice(f, a0, int(n), adur, q0, qd)
            continue
        got = tl.overlap_slice(f, a0, int(n), adur, q0, qd)
        if not valid:
            assert got is None
            continue
        n_valid += 1
        assert got is not None
        assert got[0] == o0 and got[1] == od and got[2] == int(first) and got[3] == int(count)
    assert n_valid > 200


def test_product_timed_array_matches_reference_table(g10):
    from data_utils.base import TimedArray, overlap_window

    table, _ = g10
    for f, a0, n, adur, q0, qd, valid, o0, od, first, count, raised in table:
        if f:
            ta = TimedArray(frequency=f, start=a0, data=np.zeros((2, int(n)), dtype=np.float32))
        else:
            ta = TimedArray(frequency=0, start=a0, duration=adur, data=np.zeros((2,), dtype=np.float32))
        assert ta.duration == adur
        if raised:
            with pytest.raises(RuntimeError):
                ta._overlap_slice(q0, qd)
            continue
        got = ta._overlap_slice(q0, qd)
        if not valid:
            assert got is None
            continue
        assert got[0] == o0 and got[1] == od
        if f:
            assert (got[2].start, got[2].stop - got[2].start) == (int(first), int(count))
        else:
            assert got[2] is None
    # the vectorised form agrees with the scalar one, one array against all queries that share its parameters
    sel = table[(table[:, 0] == 2.0) & (table[:, 11] == 0)]
    for row in sel[:40]:
        f, a0, n, adur = row[:4]
        same = sel[(sel[:, 1] == a0) & (sel[:, 2] == n)]
        valid, o0, od, first, count = overlap_window(f, a0, int(n), adur, same[:, 4], same[:, 5])
        assert np.array_equal(valid, same[:, 6].astype(bool))
        v = valid
        assert np.array_equal(o0[v], same[v, 7]) and np.array_equal(od[v], same[v, 8])
        assert np.array_equal(first[v], same[v, 9].astype(np.int64)) and np.array_equal(count[v], same[v, 10].astype(np.int64))


def test_product_timed_array_iadd_and_overlap():
    from data_utils.base import TimedArray

    rng = np.random.default_rng(0)
    data = rng.standard_normal((3, 40)).astype(np.float32)
    src = TimedArray(frequency=2.0, start=5.0, data=data)
    out = TimedArray(aggregation="sum", start=10.0, frequency=2.0, duration=8.0)
    out += src.overlap(10.0, 8.0)
    assert out.data.shape == (3, 16) and np.array_equal(out.data, data[:, 10:26])
    word = TimedArray(frequency=0, start=11.2, duration=0.4, data=np.arange(3, dtype=np.float32))
    out += word
    expect = data[:, 10:26].copy()
    expect[:, 2:3] += np.arange(3, dtype=np.float32)[:, None]
    assert np.array_equal(out.data, expect)
    with pytest.raises(ValueError):
        TimedArray(frequency=2.0, start=0.0, data=np.zeros((2, 10)), duration=20.0)
    with pytest.raises(ValueError):
        out += TimedArray(frequency=3.0, start=10.0, data=np.zeros((3, 24), dtype=np.float32))
    avg = TimedArray(aggregation="average", start=0.0, frequency=2.0, duration=2.0)
    avg += TimedArray(frequency=2.0, start=0.0, data=np.full((1, 4), 2.0))
    avg += TimedArray(frequency=2.0, start=0.0, data=np.full((1, 4), 4.0))
    assert np.allclose(avg.data, 3.0)


@pytest.mark.parametrize("flavour", ["audio", "video"])
def test_oracle_dense_assembly_matches_reference(g11, flavour):
    ev_start, ev_dur = g11["dense_ev_start"], g11["dense_ev_dur"]
    states = [g11["dense_states0"], g11["dense_states1"]]
    for ci, (layers, la) in enumerate(_layer_cfgs(g11)):
        for si, (s0, sd) in enumerate(zip(g11["dense_seg_start"], g11["dense_seg_dur"])):
            events = [(float(ev_start[e]), states[e], None if flavour == "audio" else float(ev_dur[e])) for e in range(2)]
            got = tl.assemble_dense(events, float(s0), float(sd), layers, la)
            want = g11[f"dense_{flavour}_cfg{ci}_seg{si}"]
            assert got.shape == want.shape and got.dtype == want.dtype
            assert np.array_equal(got, want), (flavour, ci, si)


def test_oracle_word_and_fmri_assembly_match_reference(g11):
    for ci, (layers, la) in enumerate(_layer_cfgs(g11)):
        for si, (s0, sd) in enumerate(zip(g11["word_seg_start"], g11["word_seg_dur"])):
            got = tl.assemble_words(g11["word_start"], g11["word_dur"], g11["word_states"], float(s0), float(sd), layers, la)
            want = g11[f"word_cfg{ci}_seg{si}"]
            assert got.shape == want.shape and np.array_equal(got, want), (ci, si)
    for si, (s0, sd) in enumerate(zip(g11["fmri_seg_start"], g11["fmri_seg_dur"])):
        got = tl.assemble_fmri(g11["fmri_data"], float(g11["fmri_start"]), float(s0), float(sd))
        want = g11[f"fmri_seg{si}"]
        assert got.shape == want.shape and np.array_equal(got, want), si
