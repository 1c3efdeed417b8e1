"""The reference's feature-plugin surface (`LLAMA3p2`, `Wav2VecBert`, `VJEPA2`: prepare / __call__ / _get_data) on the GPU.

(1) `__call__(events, start, duration, trigger) -> Tensor[L, D, T]` against the G11 golden tensors, which were produced by
    EXECUTING the reference's TimedArray + `_aggregate_layers` code on the same states (tests/golden/make_golden_timeline.py).
    The hidden states enter through the plugin's item cache (a warm exca cache in the reference).  Bar: bit-exact f32.
(2) `_get_data` / `prepare` end to end with tiny random-weight architectures attached: events expose `read()`, the HIP
    extractor runs, the result is compared with the reference route on the CPU (transformers' own model classes + the
    reference's post-processing restated in oracle/extractors_ref.py).  Bar: 3 % relative L2 (bf16 extractors)."""

from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def g11():
    return np.load(GOLD / "g11_segment_assembly.npz")


def layer_cfgs(g11):
    return [([float(v) for v in row if v >= 0], "group_mean" if gm else None)
            for row, gm in zip(g11["layer_cfg_layers"], g11["layer_cfg_group_mean"])]


@pytest.mark.parametrize("flavour", ["audio", "video"])
def test_sampled_plugins_call_matches_reference_tensors(g11, flavour):
    from data_utils.events import Sound, Video, Word
    from data_utils.features.audio import Wav2VecBert
    from data_utils.features.video import VJEPA2

    Plugin, Ev = (Wav2VecBert, Sound) if flavour == "audio" else (VJEPA2, Video)
    for ci, (layers, la) in enumerate(layer_cfgs(g11)):
        feat = Plugin(layers=layers, layer_aggregation=la, device="cuda")
        assert feat.name == Plugin.__name__ and feat.infra.folder is None
        events = [Ev(start=float(g11["dense_ev_start"][e]), duration=float(g11["dense_ev_dur"][e]), filepath=f"movie{e}.mkv", timeline="t")
                  for e in range(2)]
        for e, ev in enumerate(events):
            feat._ram[feat._item_uid(ev)] = g11[f"dense_states{e}"]              # warm item cache: no model needed
        distract = [Word(start=4.0, duration=0.2, text="x", timeline="t")]        # other event types in the segment are ignored
        feat.prepare(events + distract)
        for si, (s0, sd) in enumerate(zip(g11["dense_seg_start"], g11["dense_seg_dur"])):
            want = g11[f"dense_{flavour}_cfg{ci}_seg{si}"]
            got = feat(events + distract, start=float(s0), duration=float(sd), trigger=None)
            assert got.is_cuda and got.dtype == torch.float32 and tuple(got.shape) == want.shape, (ci, si, got.shape, want.shape)
            assert np.array_equal(got.cpu().numpy(), want), (flavour, ci, si)
        # no event of this type in the segment: the missing-feature default repeated over the window (audio.py:97-105)
        empty = feat(distract, start=0.0, duration=7.0)
        assert tuple(empty.shape) == want.shape[:-1] + (14,) and not empty.any()


def test_text_plugin_call_matches_reference_tensors(g11):
    from data_utils.events import Word
    from data_utils.features.text import LLAMA3p2

    words = [Word(start=float(s), duration=float(d), text=f"w{i}", context=f"ctx {i}", timeline="tl")
             for i, (s, d) in enumerate(zip(g11["word_start"], g11["word_dur"]))]
    for ci, (layers, la) in enumerate(layer_cfgs(g11)):
        feat = LLAMA3p2(layers=layers, layer_aggregation=la, device="cuda")
        for i, w in enumerate(words):
            feat._ram[feat._item_uid(w)] = g11["word_states"][i]
        feat.prepare(words)
        for si, (s0, sd) in enumerate(zip(g11["word_seg_start"], g11["word_seg_dur"])):
            want = g11[f"word_cfg{ci}_seg{si}"]
            got = feat(words, start=float(s0), duration=float(sd))
            assert tuple(got.shape) == want.shape and np.array_equal(got.cpu().numpy(), want), (ci, si)


def test_plugins_refuse_to_compute_without_a_model_or_gpu_path():
    from data_utils.events import Sound
    from data_utils.features.audio import Wav2VecBert

    feat = Wav2VecBert(device="cpu")
    with pytest.raises(RuntimeError, match="no CPU path"):
        feat.prepare([Sound(start=0.0, duration=2.0, filepath="a.wav")])
    ro = Wav2VecBert(device="cuda", infra={"mode": "read-only"})
    with pytest.raises(RuntimeError, match="read-only"):
        list(ro._get_data([Sound(start=0.0, duration=2.0, filepath="a.wav")]))


class _Tok:
    """Whitespace 'tokenizer' with the call signature the plugin uses (ids from a fixed table, right padding)."""

    eos_token_id = 7
    pad_token = "<eos>"

    def __init__(self, vocab):
        self.vocab = vocab

    def __call__(self, texts, add_special_tokens=False, return_tensors="pt", padding=True, truncation=True):
        rows = [[8 + (sum(map(ord, w)) % (self.vocab - 8)) for w in t.split()] for t in texts]
        n = max(len(r) for r in rows)
        ids = torch.full((len(rows), n), self.eos_token_id, dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r)
        return {"input_ids": ids, "attention_mask": (ids != self.eos_token_id).long()}


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def test_get_data_runs_the_hip_extractors_and_matches_the_cpu_route(tmp_path):
    from tests.test_gpu_extractors import _tiny_llama, _tiny_vjepa2, _tiny_w2vbert

    from data_utils.events import Sound, Video, Word
    from data_utils.features.audio import HipWav2Vec2Bert, Wav2VecBert
    from data_utils.features.text import LLAMA3p2, HipLlamaModel
    from data_utils.features.video import VJEPA2, HipVJEPA2Encoder, default_video_processor
    from oracle import extractors_ref

    rng = np.random.default_rng(5)

    # ---- text -----------------------------------------------------------------------------------------------------
    lcfg, lhf = _tiny_llama()
    tok = _Tok(lcfg.vocab_size)
    sentence = "the quick brown fox jumps over the lazy dog again and again".split()
    words = [Word(start=1.0 + 0.6 * i, duration=0.4, text=w, context=" ".join(sentence[: i + 1]), timeline="t") for i, w in enumerate(sentence)]
    text = LLAMA3p2(device="cuda", infra={"folder": str(tmp_path / "cache")}).attach(HipLlamaModel(lcfg, lhf.state_dict()), tok)
    got = list(text._get_data(words))
    enc = tok([w.context for w in words])
    want = extractors_ref.llama_word_states(lhf, enc["input_ids"], enc["attention_mask"], [w.text for w in words], tok.eos_token_id)
    assert len(got) == len(words) and got[0].shape == (lcfg.num_hidden_layers + 1, lcfg.hidden_size)
    assert max(_rel(g, w) for g, w in zip(got, want)) < 3e-2
    # second request: served from the item cache (RAM, and the folder for a fresh plugin), the extractor is not called again
    text._model = None
    again = list(text._get_data(words))
    assert all(np.array_equal(a, b) for a, b in zip(got, again))
    fresh = LLAMA3p2(device="cuda", infra={"folder": str(tmp_path / "cache")})
    assert all(np.array_equal(a, b) for a, b in zip(got, fresh._get_data(words)))
    tensor = fresh(words, start=0.0, duration=8.0)
    assert tuple(tensor.shape) == (2, lcfg.hidden_size, 16)

    # ---- audio ----------------------------------------------------------------------------------------------------
    wcfg, whf = _tiny_w2vbert()
    wav = rng.standard_normal((16000 * 6, 2)).astype(np.float32)

    class _Snd(Sound):
        def read(self):
            return torch.from_numpy(wav)

    snd = _Snd(start=0.0, duration=6.0, filepath="a.wav", frequency=16000.0, timeline="t")
    audio = Wav2VecBert(device="cuda").attach(HipWav2Vec2Bert(wcfg, whf.state_dict()))
    (lat,) = list(audio._get_data([snd]))
    mono = audio._preprocess_wav(torch.from_numpy(wav))
    feats = audio._get_features(mono)
    with torch.no_grad():
        hs = torch.stack(whf(feats, output_hidden_states=True).hidden_states).squeeze(1).transpose(-1, -2)   # audio.py:253-263
    want_a = torch.nn.functional.interpolate(hs, 12).numpy()                                                  # audio.py:163-171
    assert lat.shape == want_a.shape == (wcfg.num_hidden_layers + 1, wcfg.hidden_size, 12)
    assert _rel(lat, want_a) < 3e-2
    audio.prepare([snd])
    assert tuple(audio([snd], start=1.0, duration=3.0).shape) == (1, wcfg.hidden_size, 6)   # 3 states: layers [1, 2] -> one group [1:3]

    # ---- video ----------------------------------------------------------------------------------------------------
    vcfg, vhf = _tiny_vjepa2()

    class _Clip:
        duration = 1.0

        def get_frame(self, t):
            base = (np.arange(80 * 96 * 3).reshape(80, 96, 3) * 7 + int(t * 1000)) % 251
            return base.astype(np.uint8)

    class _Vid(Video):
        def read(self):
            return _Clip()

    vid = _Vid(start=0.0, duration=1.0, filepath="v.mkv", timeline="t")
    video = VJEPA2(device="cuda").attach(HipVJEPA2Encoder(vcfg, vhf.state_dict()))
    (lat_v,) = list(video._get_data([vid]))
    assert lat_v.shape == (vcfg.num_hidden_layers + 1, vcfg.hidden_size, 2) and lat_v.dtype == np.float64
    clip = _Clip()
    times = np.linspace(0, 1.0, 3)[1:]
    subtimes = [k / vcfg.frames_per_clip * 4.0 for k in reversed(range(vcfg.frames_per_clip))]
    for k, t in enumerate(times):
        frames = np.array([clip.get_frame(max(0, t - t2)) for t2 in subtimes])
        pix = default_video_processor(frames, vcfg.crop_size)
        with torch.no_grad():
            states = vhf(pixel_values_videos=pix, output_hidden_states=True, skip_predictor=True).hidden_states
        want_v = torch.stack([s[0].mean(0) for s in states]).numpy()                                          # video.py:225-228
        assert _rel(lat_v[:, :, k], want_v) < 3e-2, k


def test_plugins_share_one_store_with_the_batch_loader(g11):
    """the three plugins bound to ONE HbmFeatureStore under the SegmentData keys fill it in `prepare`;
    `GpuSegmentLoader.batch` then writes the projector operands directly -- equal to bf16(plugin __call__ tensor) per segment."""
    from data_utils.events import Sound, Video, Word
    from data_utils.features.audio import Wav2VecBert
    from data_utils.features.text import LLAMA3p2
    from data_utils.features.video import VJEPA2
    from data_utils.gpu_loader import GpuSegmentLoader, HbmFeatureStore
    from data_utils.segments import Segment

    snd = [Sound(start=float(g11["dense_ev_start"][e]), duration=float(g11["dense_ev_dur"][e]), filepath=f"a{e}.wav", timeline="t", extra={"subject": "sub-02"}) for e in range(2)]
    vid = [Video(start=float(g11["dense_ev_start"][e]), duration=float(g11["dense_ev_dur"][e]), filepath=f"v{e}.mkv", timeline="t", extra={"subject": "sub-02"}) for e in range(2)]
    words = [Word(start=float(s), duration=float(d), text=f"w{i}", context=f"c{i}", timeline="t", extra={"subject": "sub-02"})
             for i, (s, d) in enumerate(zip(g11["word_start"], g11["word_dur"]))]
    text, audio, video = LLAMA3p2(device="cuda"), Wav2VecBert(device="cuda"), VJEPA2(device="cuda")
    for e in range(2):
        audio._ram[audio._item_uid(snd[e])] = g11[f"dense_states{e}"]
        video._ram[video._item_uid(vid[e])] = g11[f"dense_states{e}"]
    for i, w in enumerate(words):
        text._ram[text._item_uid(w)] = g11["word_states"][i]
    store = HbmFeatureStore([])
    events = words + snd + vid
    for key, f in (("text", text), ("audio", audio), ("video", video)):
        f.bind(store, name=key)
        f.prepare(events)
    loader = GpuSegmentLoader(store)
    segs = [Segment(start=s0, duration=12.0, ns_events=list(events)) for s0 in (3.0, 10.25, 20.0)]
    batch = loader.batch(segs)
    assert set(batch.data) == {"text", "audio", "video"}
    for key, f in (("text", text), ("audio", audio), ("video", video)):
        got = batch.data[key].unpack()
        for b, seg in enumerate(segs):
            want = f(seg.ns_events, start=seg.start, duration=seg.duration)
            assert torch.equal(got[b], want.bfloat16().float().reshape(got[b].shape)), (key, b)


def test_segment_dataset_with_all_five_features(g11):
    """The reference's SegmentDataset (dataloader.py:111-187) over this build's plugins: text / audio / video / Fmri / SubjectEncoder
    items (`__getitem__` = feature(events, start, duration, trigger) + pad + batch axis), collated, against the G11 tensors; and the
    batched fast path (`gpu_batches`) against the collated items."""
    from data_utils.dataloader import SegmentDataset
    from data_utils.events import Fmri as FmriEvent
    from data_utils.events import Sound, Video, Word
    from data_utils.features.audio import Wav2VecBert
    from data_utils.features.neuro import Fmri
    from data_utils.features.subject import SubjectEncoder
    from data_utils.features.text import LLAMA3p2
    from data_utils.features.video import VJEPA2
    from data_utils.gpu_loader import HbmFeatureStore
    from data_utils.segments import Segment

    snd = [Sound(start=float(g11["dense_ev_start"][e]), duration=float(g11["dense_ev_dur"][e]), filepath=f"a{e}.wav", timeline="t", extra={"subject": "sub-02"}) for e in range(2)]
    vid = [Video(start=float(g11["dense_ev_start"][e]), duration=float(g11["dense_ev_dur"][e]), filepath=f"v{e}.mkv", timeline="t", extra={"subject": "sub-02"}) for e in range(2)]
    words = [Word(start=float(s), duration=float(d), text=f"w{i}", context=f"c{i}", timeline="t", extra={"subject": "sub-02"})
             for i, (s, d) in enumerate(zip(g11["word_start"], g11["word_dur"]))]
    rec = FmriEvent(start=float(g11["fmri_start"]), duration=g11["fmri_data"].shape[1] * 1.49, filepath="sub-02.h5", frequency=1 / 1.49, subject="sub-02", timeline="t")
    other = FmriEvent(start=0.0, duration=10.0, filepath="sub-01.h5", frequency=1 / 1.49, subject="sub-01", timeline="u")
    feats = {"text": LLAMA3p2(device="cuda"), "audio": Wav2VecBert(device="cuda"), "video": VJEPA2(device="cuda"), "fmri": Fmri(device="cuda"),
             "subject_id": SubjectEncoder()}
    for e in range(2):
        feats["audio"]._ram[feats["audio"]._item_uid(snd[e])] = g11[f"dense_states{e}"]
        feats["video"]._ram[feats["video"]._item_uid(vid[e])] = g11[f"dense_states{e}"]
    for i, w in enumerate(words):
        feats["text"]._ram[feats["text"]._item_uid(w)] = g11["word_states"][i]
    feats["fmri"]._ram[feats["fmri"]._item_uid(rec)] = g11["fmri_data"]
    store = HbmFeatureStore([])
    events = [rec] + words + snd + vid
    for key, f in feats.items():
        if key != "subject_id":
            f.bind(store, name=key)
        f.prepare(events + [other] if key == "subject_id" else events)
    assert feats["subject_id"].subject_index == {"sub-01": 0, "sub-02": 1}

    # fMRI items reproduce the reference tensors of G11 (one window each; the recording sits 4.47 s before its event)
    for si, (s0, sd) in enumerate(zip(g11["fmri_seg_start"], g11["fmri_seg_dur"])):
        got = feats["fmri"]([rec], start=float(s0), duration=float(sd))
        assert np.array_equal(got.cpu().numpy(), g11[f"fmri_seg{si}"]), si

    segs = [Segment(start=s0, duration=14.9, ns_events=list(events), _trigger=s0) for s0 in (3.0, 10.25, 20.0, 29.8)]
    ds = SegmentDataset(feats, segs, pad_duration=14.9)
    item = ds[1]
    assert item.data["text"].shape[0] == 1 and item.data["text"].shape[-1] == 30 and item.data["fmri"].shape == (1, g11["fmri_data"].shape[0], 10)
    assert item.data["subject_id"].tolist() == [[1]]
    batch = ds.as_one_batch()
    assert batch.data["audio"].shape[0] == 4 and len(batch.segments) == 4 and batch.data["subject_id"].tolist() == [[1]] * 4
    fast = list(ds.gpu_batches(batch_size=4))[0]
    for key in ("text", "audio", "video"):
        assert torch.equal(fast.data[key].unpack(), batch.data[key].bfloat16().float().reshape(fast.data[key].shape)), key
    assert torch.equal(fast.data["fmri"], batch.data["fmri"]) and torch.equal(fast.data["subject_id"].cpu(), batch.data["subject_id"].cpu())
    with pytest.warns(UserWarning, match="cropping"):
        short = SegmentDataset({"audio": feats["audio"]}, segs, pad_duration=5.0)[0]
    assert short.data["audio"].shape[-1] == 10
