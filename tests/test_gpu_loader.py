"""GPU segment loader (csrc/features.hip through the C ABI) against the reference's tensors (g11 golden vectors) and
against the oracle restatement at realistic sizes.  Bar: bit-exact -- f32 for layer aggregation and fMRI targets,
bf16(reference f32) for the packed projector operands (byte-moving work, no tolerance)."""

from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import timeline_ref as tl
from oracle.tribe_ref import aggregate_layers

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def g11():
    return np.load(GOLD / "g11_segment_assembly.npz")


def layer_cfgs(g11):
    return [([float(v) for v in row if v >= 0], "group_mean" if gm else None)
            for row, gm in zip(g11["layer_cfg_layers"], g11["layer_cfg_group_mean"])]


def bf16_round(a: np.ndarray) -> np.ndarray:
    return torch.from_numpy(np.ascontiguousarray(a)).bfloat16().float().numpy()


@pytest.mark.parametrize("shape", [(25, 1024, 37), (29, 96, 1), (41, 10, 7), (5, 3, 33)])
def test_group_mean_bit_exact_vs_numpy(shape):
    from data_utils.gpu_loader import layer_groups
    from tribe_hip import ops

    rng = np.random.default_rng(1)
    x = rng.standard_normal((2,) + shape).astype(np.float32)
    for layers, la in [([0.5, 0.75, 1.0], "group_mean"), ([0.0, 0.2, 0.4, 0.6, 0.8, 1.0], "group_mean"), ([0.5, 1.0], None), ([1.0], None)]:
        lo, hi = layer_groups(shape[0], layers, la)
        got = ops.group_mean(torch.from_numpy(x).cuda(), torch.tensor(lo, dtype=torch.int32).cuda(), torch.tensor(hi, dtype=torch.int32).cuda())
        want = np.stack([aggregate_layers(x[b], layers, la).reshape(len(lo), *shape[1:]) for b in range(2)])
        assert np.array_equal(got.cpu().numpy(), want), (layers, la)


@pytest.mark.parametrize("flavour", ["audio", "video"])
def test_sampled_feature_matches_reference_tensors(g11, flavour):
    from data_utils.events import Segment, Sound, Video
    from data_utils.gpu_loader import FeatureSpec, GpuSegmentLoader, HbmFeatureStore

    Ev = Sound if flavour == "audio" else Video
    for ci, (layers, la) in enumerate(layer_cfgs(g11)):
        spec = FeatureSpec(flavour, "sampled", Ev.__name__, layers=layers, layer_aggregation=la, pass_event_duration=flavour == "video")
        store = HbmFeatureStore([spec])
        events = [Ev(start=float(g11["dense_ev_start"][e]), duration=float(g11["dense_ev_dur"][e]), filepath=f"movie{e}.mkv") for e in range(2)]
        for e, ev in enumerate(events):
            store.put(flavour, ev, g11[f"dense_states{e}"])
        loader = GpuSegmentLoader(store)
        segs = [Segment(start=float(s0), duration=float(sd), ns_events=list(events)) for s0, sd in zip(g11["dense_seg_start"], g11["dense_seg_dur"])]
        for si, seg in enumerate(segs):
            want = g11[f"dense_{flavour}_cfg{ci}_seg{si}"]
            got = loader.feature(spec, [seg])
            L, D = store.channels[flavour]
            assert got.packed.shape == (want.shape[-1], 64) and got.shape == (1, L, D, want.shape[-1])
            assert np.array_equal(got.unpack()[0].cpu().numpy().reshape(want.shape), bf16_round(want)), (flavour, ci, si)
            assert not got.packed[:, L * D:].any()                                       # zero padding columns
        # one batch over all segments with a common padded / cropped length (dataloader.py:69-98)
        loader_pad = GpuSegmentLoader(store, pad_duration=12.0)
        got = loader_pad.feature(spec, segs).unpack().cpu().numpy()
        for si in range(len(segs)):
            want = g11[f"dense_{flavour}_cfg{ci}_seg{si}"].reshape(got.shape[1] * got.shape[2], -1)
            ref = np.zeros((want.shape[0], 24), np.float32)
            n = min(24, want.shape[1])
            ref[:, :n] = want[:, :n]
            assert np.array_equal(got[si].reshape(ref.shape), bf16_round(ref)), (flavour, ci, si)


def test_word_and_fmri_features_match_reference_tensors(g11):
    from data_utils.events import Fmri, Segment, Word
    from data_utils.gpu_loader import FeatureSpec, GpuSegmentLoader, HbmFeatureStore

    words = [Word(start=float(s), duration=float(d), text=f"w{i}", timeline="tl") for i, (s, d) in
             enumerate(zip(g11["word_start"], g11["word_dur"]))]
    for ci, (layers, la) in enumerate(layer_cfgs(g11)):
        spec = FeatureSpec("text", "words", "Word", layers=layers, layer_aggregation=la)
        store = HbmFeatureStore([spec])
        store.put_words("text", words[:25], g11["word_states"][:25])
        store.put_words("text", words[20:], g11["word_states"][20:])                    # overlapping second chunk: rows are not duplicated
        assert store.word_table("text").shape[0] == len(words)
        loader = GpuSegmentLoader(store)
        for si, (s0, sd) in enumerate(zip(g11["word_seg_start"], g11["word_seg_dur"])):
            want = g11[f"word_cfg{ci}_seg{si}"]
            got = loader.feature(spec, [Segment(start=float(s0), duration=float(sd), ns_events=list(words))])
            assert np.array_equal(got.unpack()[0].cpu().numpy().reshape(want.shape), bf16_round(want)), (ci, si)
    spec = [s for s in FeatureSpec.defaults() if s.name == "fmri"][0]
    store = HbmFeatureStore([spec])
    rec = Fmri(start=float(g11["fmri_start"]), duration=g11["fmri_data"].shape[1] * 1.49, filepath="sub-01.h5", frequency=1 / 1.49, subject="sub-01")
    store.put("fmri", rec, g11["fmri_data"])
    loader = GpuSegmentLoader(store, subject_index={"sub-01": 0})
    for si, (s0, sd) in enumerate(zip(g11["fmri_seg_start"], g11["fmri_seg_dur"])):
        batch = loader.batch([Segment(start=float(s0), duration=float(sd), ns_events=[rec])])
        assert np.array_equal(batch.data["fmri"][0].cpu().numpy(), g11[f"fmri_seg{si}"]), si
        assert batch.data["subject_id"].tolist() == [[0]]


def test_full_size_batch_vs_oracle_and_model_equivalence():
    """Reference dims: Llama word latents [29, 3072], w2v-bert [25, 1024, T], V-JEPA2 [41, 1408, T]; 8 segments of 100 s."""
    from algonauts2025.model import FmriEncoderConfig
    from data_utils.dataloader import SegmentData
    from data_utils.events import Fmri, Segment, Sound, Video, Word
    from data_utils.gpu_loader import FeatureSpec, GpuSegmentLoader, HbmFeatureStore

    rng = np.random.default_rng(7)
    T_ev = 700                                                                          # 350 s movie chunk at 2 Hz
    audio = rng.standard_normal((25, 1024, T_ev)).astype(np.float32)
    video = rng.standard_normal((41, 1408, T_ev)).astype(np.float32)
    n_words = 900
    w_start = np.sort(rng.uniform(0, 350, n_words)).round(3)
    w_dur = rng.uniform(0.05, 0.9, n_words).round(3)
    w_lat = rng.standard_normal((n_words, 29, 384)).astype(np.float32)                  # narrower D keeps the oracle loop quick
    fm = rng.standard_normal((1000, 236)).astype(np.float32)
    specs = FeatureSpec.defaults()
    store = HbmFeatureStore(specs)
    snd, vid = Sound(start=1.0, duration=350.0, filepath="a.wav"), Video(start=1.0, duration=350.0, filepath="v.mkv")
    rec = Fmri(start=1.0, duration=236 * 1.49, filepath="f.h5", frequency=1 / 1.49, subject="sub-03")
    words = [Word(start=float(s) + 1.0, duration=float(d), text=f"w{i}", timeline="t") for i, (s, d) in enumerate(zip(w_start, w_dur))]
    store.put("audio", snd, audio)
    store.put("video", vid, video)
    store.put("fmri", rec, fm)
    store.put_words("text", words, w_lat)
    loader = GpuSegmentLoader(store, subject_index={"sub-01": 0, "sub-03": 1})
    starts = [1.0, 37.5, 100.25, 180.0, 249.0, 251.5, 12.0, 60.0]
    segs = [Segment(start=s, duration=100.0, ns_events=[rec, snd, vid] + [w for w in words if w.start < s + 100.0 and w.stop > s]) for s in starts]
    batch = loader.batch(segs)
    by = {s.name: s for s in specs}
    for b, seg in enumerate(segs):
        a = tl.assemble_dense([(snd.start, audio, None)], seg.start, seg.duration, by["audio"].layers, by["audio"].layer_aggregation)
        v = tl.assemble_dense([(vid.start, video, vid.duration)], seg.start, seg.duration, by["video"].layers, by["video"].layer_aggregation)
        ws = [w for w in seg.ns_events if w.type == "Word"]
        idx = [words.index(w) for w in ws]
        t = tl.assemble_words(np.asarray([w.start for w in ws]), np.asarray([w.duration for w in ws]), w_lat[idx], seg.start, seg.duration,
                              by["text"].layers, by["text"].layer_aggregation)
        f = tl.assemble_fmri(fm, rec.start, seg.start, seg.duration)
        assert np.array_equal(batch.data["audio"].unpack()[b].cpu().numpy(), bf16_round(a)), b
        assert np.array_equal(batch.data["video"].unpack()[b].cpu().numpy(), bf16_round(v)), b
        assert np.array_equal(batch.data["text"].unpack()[b].cpu().numpy(), bf16_round(t)), b
        assert np.array_equal(batch.data["fmri"][b].cpu().numpy(), f), b
    assert batch.data["subject_id"].flatten().tolist() == [1] * 8
    # the model consumes the packed batch directly and gives the same predictions as from reference-layout tensors
    T = 200
    fdims = {m: tuple(store.channels[m]) for m in ("text", "audio", "video")}
    torch.manual_seed(0)
    model = FmriEncoderConfig(n_subjects=2, hidden=768, depth=2, heads=4, max_timesteps=256).build(fdims, 1000, 67).cuda().eval()
    y_packed = model(batch)
    plain = {m: batch.data[m].unpack() for m in fdims}
    plain["subject_id"] = batch.data["subject_id"]
    y_plain = model(SegmentData(data=plain, segments=segs))
    assert y_packed.shape == (8, 1000, 67) and torch.equal(y_packed, y_plain)
    assert batch.data["fmri"].shape == (8, 1000, 67) and T == batch.data["audio"].T
