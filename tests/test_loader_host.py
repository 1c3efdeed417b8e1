"""Host side of the GPU segment loader on CPU: the per-segment plans (which slice of which cached array lands where)
replayed with numpy must reproduce the reference's tensors (g11 golden vectors from the reference's base.py).
The kernels that normally execute the plans are covered by tests/test_gpu_loader.py."""

from pathlib import Path

import numpy as np
import pytest
import torch

from oracle.tribe_ref import aggregate_layers

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def g11():
    return np.load(GOLD / "g11_segment_assembly.npz")


def layer_cfgs(g11):
    return [([float(v) for v in row if v >= 0], "group_mean" if gm else None)
            for row, gm in zip(g11["layer_cfg_layers"], g11["layer_cfg_group_mean"])]


def _replay_pieces(plan, arrays_by_ptr, C):
    out = np.zeros((C, plan.n_out), dtype=np.float32)
    for pc in plan.pieces:
        src = arrays_by_ptr[int(pc["src"])]
        assert src.shape[1] == pc["ld"]
        sl = src[:, pc["src_first"]:pc["src_first"] + pc["src_count"]]
        out[:, pc["dst_first"]:pc["dst_first"] + pc["dst_count"]] += sl
    return out


def _cpu_store(specs):
    from data_utils.gpu_loader import HbmFeatureStore

    return HbmFeatureStore(specs, device="cpu")


@pytest.mark.parametrize("flavour", ["audio", "video"])
def test_sampled_plans_reproduce_reference(g11, flavour):
    from data_utils.events import Segment, Sound, Video
    from data_utils.gpu_loader import FeatureSpec, GpuSegmentLoader, _Resident

    Ev = Sound if flavour == "audio" else Video
    for ci, (layers, la) in enumerate(layer_cfgs(g11)):
        spec = FeatureSpec(flavour, "sampled", Ev.__name__, layers=layers, layer_aggregation=la, pass_event_duration=flavour == "video")
        store = _cpu_store([spec])
        events, by_ptr = [], {}
        for e in range(2):
            ev = Ev(start=float(g11["dense_ev_start"][e]), duration=float(g11["dense_ev_dur"][e]), filepath=f"movie{e}.mkv")
            agg = aggregate_layers(g11[f"dense_states{e}"], layers, la)
            agg = agg.reshape(-1, agg.shape[-1])
            t = torch.from_numpy(np.ascontiguousarray(agg))
            store._sampled[(flavour, store.event_key(ev))] = _Resident(t, t.shape[1])   # what store.put leaves in HBM
            by_ptr[t.data_ptr()] = t.numpy()
            events.append(ev)
        loader = GpuSegmentLoader(store)
        for si, (s0, sd) in enumerate(zip(g11["dense_seg_start"], g11["dense_seg_dur"])):
            want = g11[f"dense_{flavour}_cfg{ci}_seg{si}"]
            seg = Segment(start=float(s0), duration=float(sd), ns_events=list(events))
            plan = loader.plan(seg, spec)
            got = _replay_pieces(plan, by_ptr, agg.shape[0])
            assert got.shape[-1] == want.shape[-1]
            assert np.array_equal(got.reshape(want.shape), want), (flavour, ci, si)
            assert loader.plan(seg, spec) is plan                                       # cached per segment object


def test_word_plans_reproduce_reference(g11):
    from data_utils.events import Segment, Word
    from data_utils.gpu_loader import FeatureSpec, GpuSegmentLoader

    words = [Word(start=float(s), duration=float(d), text=f"w{i}", timeline="tl") for i, (s, d) in
             enumerate(zip(g11["word_start"], g11["word_dur"]))]
    for ci, (layers, la) in enumerate(layer_cfgs(g11)):
        spec = FeatureSpec("text", "words", "Word", layers=layers, layer_aggregation=la)
        store = _cpu_store([spec])
        store._word_rows["text"] = {store.event_key(w): i for i, w in enumerate(words)}
        table = np.stack([aggregate_layers(g11["word_states"][i], layers, la).reshape(-1) for i in range(len(words))])
        loader = GpuSegmentLoader(store)
        for si, (s0, sd) in enumerate(zip(g11["word_seg_start"], g11["word_seg_dur"])):
            want = g11[f"word_cfg{ci}_seg{si}"]
            plan = loader.plan(Segment(start=float(s0), duration=float(sd), ns_events=list(words)), spec)
            got = np.zeros((table.shape[1], plan.n_out), dtype=np.float32)
            for step, row in zip(plan.steps, plan.rows):                                # event order, as `out += word`
                got[:, step] += table[row]
            assert np.array_equal(got.reshape(want.shape), want), (ci, si)


def test_fmri_plans_reproduce_reference(g11):
    from data_utils.events import Fmri, Segment
    from data_utils.gpu_loader import FeatureSpec, GpuSegmentLoader, _Resident

    spec = [s for s in FeatureSpec.defaults() if s.name == "fmri"][0]
    store = _cpu_store([spec])
    data = torch.from_numpy(g11["fmri_data"].copy())
    rec = Fmri(start=float(g11["fmri_start"]), duration=data.shape[1] * 1.49, filepath="sub-01_task.h5", frequency=1 / 1.49, subject="sub-01")
    store._sampled[("fmri", store.event_key(rec))] = _Resident(data, data.shape[1])
    loader = GpuSegmentLoader(store)
    for si, (s0, sd) in enumerate(zip(g11["fmri_seg_start"], g11["fmri_seg_dur"])):
        want = g11[f"fmri_seg{si}"]
        plan = loader.plan(Segment(start=float(s0), duration=float(sd), ns_events=[rec]), spec)
        got = _replay_pieces(plan, {data.data_ptr(): data.numpy()}, data.shape[0])
        assert np.array_equal(got, want), si


def test_layer_groups_and_store_errors():
    from data_utils.gpu_loader import FeatureSpec, HbmFeatureStore, PackedFeature, layer_groups
    from tribe_hip import _lib

    assert layer_groups(29, [0.5, 0.75, 1.0], "group_mean") == ([14, 21], [21, 29])     # text.py:129-149 on 29 Llama states
    assert layer_groups(25, [0.0, 0.5, 1.0], None) == ([0, 12, 24], [1, 13, 25])
    assert layer_groups(41, [1.0], "group_mean") == ([40], [41])
    with pytest.raises(ValueError):
        layer_groups(9, [0.0, 1.0], "median")
    store = HbmFeatureStore(FeatureSpec.defaults(), device="cpu")
    with pytest.raises(_lib.TribeHipError):                                              # no CPU fallback for the aggregation kernel
        store.put("audio", type("E", (), {"filepath": "a.wav", "offset": 0.0})(), np.zeros((25, 8, 10), np.float32))
    with pytest.raises(KeyError):
        store.resident("audio", type("E", (), {"filepath": "b.wav", "offset": 0.0})())
    with pytest.raises(ValueError):
        PackedFeature(torch.zeros(6, 8, dtype=torch.bfloat16), B=2, L=2, D=8, T=3)
    pf = PackedFeature(torch.arange(6 * 64, dtype=torch.float32).view(6, 64).bfloat16(), B=2, L=2, D=3, T=3)
    assert pf.shape == (2, 2, 3, 3) and pf.unpack().shape == (2, 2, 3, 3)
    assert float(pf.unpack()[1, 1, 2, 0]) == float(pf.packed[3, 5])                      # [b, l, d, t] <- row b*T + t, column l*D + d


def test_cache_file_round_trip_and_foreign_records(tmp_path):
    import json

    from data_utils.cache_file import iter_cache, load_into_store, write_cache

    rng = np.random.default_rng(3)
    items = {"movie0.mkv": rng.standard_normal((5, 6, 17)).astype(np.float32), "movie1.mkv": rng.standard_normal((5, 6, 3)),
             "w:hello": rng.standard_normal((7, 4)).astype(np.float16)}
    info = write_cache(tmp_path, items, name="vjepa2")
    write_cache(tmp_path, {"movie0.mkv": items["movie0.mkv"] + 1}, name="vjepa2")            # re-written item: the later line wins
    # lines in the shape an exca-style writer is expected to leave: a header line and a record with another key name
    with open(info, "a") as f:
        f.write(json.dumps({"cache_type": "MemmapArrayFile"}) + "\n")
    foreign = tmp_path / "other.data"
    foreign.write_bytes(b"\0" * 8 + np.arange(6, dtype=np.int32).tobytes())
    (tmp_path / "other-info.jsonl").write_text(json.dumps({"key": "x", "filename": "other.data", "offset": 8, "shape": [2, 3], "dtype": "int32"}) + "\n")
    got = dict(iter_cache(tmp_path))
    assert set(got) == {"movie0.mkv", "movie1.mkv", "w:hello", "x"}
    assert np.array_equal(got["movie0.mkv"], items["movie0.mkv"] + 1) and got["movie1.mkv"].dtype == np.float64
    assert np.array_equal(got["w:hello"], items["w:hello"]) and np.array_equal(got["x"], np.arange(6).reshape(2, 3))
    assert not got["movie1.mkv"].flags.writeable
    (tmp_path / "bad-info.jsonl").write_text(json.dumps({"#key": "y", "filename": "other.data", "offset": 16, "shape": [9], "dtype": "int32"}) + "\n")
    with pytest.raises(ValueError):
        dict(iter_cache(tmp_path))
    (tmp_path / "bad-info.jsonl").unlink()

    class Store:                                                                             # records what a real HbmFeatureStore would be handed
        specs = {"video": type("S", (), {"kind": "sampled"})(), "text": type("S", (), {"kind": "words"})()}

        def __init__(self):
            self.calls = []

        def put(self, name, ev, arr):
            self.calls.append((name, ev, arr.shape))

        def put_words(self, name, evs, arr):
            self.calls.append((name, tuple(evs), arr.shape))

    st = Store()
    assert load_into_store(st, "video", tmp_path, {"movie0.mkv": "E0", "movie1.mkv": "E1"}) == 2
    assert load_into_store(st, "text", tmp_path, {"w:hello": "W"}) == 1
    assert st.calls == [("video", "E0", (5, 6, 17)), ("video", "E1", (5, 6, 3)), ("text", ("W",), (1, 7, 4))]
