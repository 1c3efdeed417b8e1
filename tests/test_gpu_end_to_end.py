"""BASELINE config 3 at small scale: trimodal encode WITH on-the-fly extraction, everything on the GPU --
   stimulus -> HIP extractors (Llama / Wav2Vec-BERT / V-JEPA2 architectures, tiny random-weight configs)
            -> HBM feature store (layer group-mean on the GPU) -> segment loader (packed bf16 projector operands)
            -> FmriEncoder -> predictions [B, V, T'],
against the reference route restated on the CPU: transformers' own models (fp32) + the reference's post-processing
(oracle/extractors_ref.py), the reference's TimedArray assembly (oracle/timeline_ref.py, pinned by g10/g11) and the fp32
encoder oracle (oracle/tribe_ref.py) with the same parameters.  Tolerance: 3 % relative L2 on the predictions (the
extractors run in bf16: 1-2 % on their hidden states; the encoder adds ~0.3 %)."""

import numpy as np
import pytest
import torch

from oracle import extractors_ref, tribe_ref
from oracle import timeline_ref as tl

pytestmark = pytest.mark.gpu


def test_extract_store_load_encode_matches_cpu_route():
    from tests.test_gpu_extractors import _tiny_llama, _tiny_vjepa2, _tiny_w2vbert

    from algonauts2025.model import FmriEncoderConfig
    from data_utils.events import Fmri, Sound, Video, Word
    from data_utils.features.audio import HipWav2Vec2Bert
    from data_utils.features.text import HipLlamaModel, word_pool_windows
    from data_utils.features.video import HipVJEPA2Encoder
    from data_utils.gpu_loader import FeatureSpec, GpuSegmentLoader, HbmFeatureStore
    from data_utils.segments import Segment

    g = torch.Generator().manual_seed(11)
    dur, freq = 24.0, 2.0
    n_steps = int(dur * freq)                                                     # 48 feature steps at 2 Hz
    snd, vid = Sound(start=0.0, duration=dur, filepath="chunk.wav", timeline="t"), Video(start=0.0, duration=dur, filepath="chunk.mkv", timeline="t")
    rec = Fmri(start=0.0, duration=20 * 1.49, filepath="sub-01.h5", frequency=1 / 1.49, subject="sub-01", timeline="t")
    fmri = torch.randn(50, 20, generator=g).numpy()

    # ---- text: one context per word, right padded (text.py:204-256) ------------------------------------------------
    lcfg, lhf = _tiny_llama()
    pad_id, n_words, ctx = 7, 12, 20
    ids = torch.randint(8, lcfg.vocab_size, (n_words, ctx), generator=g)
    mask = torch.ones(n_words, ctx, dtype=torch.long)
    for i in range(n_words):
        n_real = 5 + i
        ids[i, n_real:] = pad_id
        mask[i, n_real:] = 0
    texts = ["w" * (1 + i % 4) for i in range(n_words)]
    words = [Word(start=1.0 + 1.7 * i, duration=0.3 + 0.05 * i, text=texts[i], timeline="t") for i in range(n_words)]
    llama = HipLlamaModel(lcfg, lhf.state_dict())
    start, length = word_pool_windows(ids, texts, pad_id)
    word_states = llama.forward_pooled(ids, start, length).permute(1, 0, 2).contiguous()           # [n_words, n_states, D]
    word_states_ref = np.stack(extractors_ref.llama_word_states(lhf, ids, mask, texts, pad_id))    # same, fp32 CPU

    # ---- audio: one chunk, hidden states resampled to 2 Hz (audio.py:253-263, 163-171) ----------------------------------
    wcfg, whf = _tiny_w2vbert()
    feats = torch.randn(1, 300, 160, generator=g)
    w2v = HipWav2Vec2Bert(wcfg, whf.state_dict())
    audio_states = w2v.hidden_states_resampled(feats, n_steps)[0]                                   # [n_states, D, T]
    with torch.no_grad():
        out = whf(feats, output_hidden_states=True)
    audio_states_ref = torch.nn.functional.interpolate(torch.stack(out.hidden_states).squeeze(1).transpose(-1, -2), n_steps).numpy()

    # ---- video: one clip per 0.5 s, token mean of every hidden state (video.py:191-236) ---------------------------------
    vcfg, vhf = _tiny_vjepa2()
    clips = torch.randn(n_steps, vcfg.frames_per_clip, 3, vcfg.crop_size, vcfg.crop_size, generator=g)
    vj = HipVJEPA2Encoder(vcfg, vhf.state_dict())
    video_states = torch.cat([vj.hidden_state_means(clips[i:i + 16]) for i in range(0, n_steps, 16)]).permute(1, 2, 0).contiguous()
    with torch.no_grad():
        outs = [vhf(pixel_values_videos=clips[i:i + 16], output_hidden_states=True, skip_predictor=True) for i in range(0, n_steps, 16)]
    video_states_ref = torch.cat([torch.cat([x.unsqueeze(1) for x in o.hidden_states], dim=1).mean(dim=2) for o in outs]).permute(1, 2, 0).numpy()

    # ---- store + loader --------------------------------------------------------------------------------------------
    specs = FeatureSpec.defaults()
    store = HbmFeatureStore(specs)
    store.put_words("text", words, word_states)
    store.put("audio", snd, audio_states)
    store.put("video", vid, video_states)
    store.put("fmri", rec, fmri)
    loader = GpuSegmentLoader(store, subject_index={"sub-01": 0})
    seg_starts = [0.0, 4.25, 12.0]
    segs = [Segment(start=s, duration=12.0, ns_events=[rec, snd, vid] + [w for w in words if w.start < s + 12.0 and w.stop > s]) for s in seg_starts]
    batch = loader.batch(segs)
    T = batch.data["audio"].T
    fdims = {m: tuple(store.channels[m]) for m in ("text", "audio", "video")}
    V, Tout = 50, batch.data["fmri"].shape[-1]

    # ---- the same batch by the reference route on the CPU --------------------------------------------------------------
    by = {s.name: s for s in specs}
    ref_data = {"text": [], "audio": [], "video": []}
    for seg in segs:
        ws = [w for w in seg.ns_events if getattr(w, "type", "") == "Word"]
        idx = [words.index(w) for w in ws]
        ref_data["text"].append(tl.assemble_words(np.asarray([w.start for w in ws]), np.asarray([w.duration for w in ws]), word_states_ref[idx],
                                                  seg.start, seg.duration, by["text"].layers, by["text"].layer_aggregation))
        ref_data["audio"].append(tl.assemble_dense([(snd.start, audio_states_ref, None)], seg.start, seg.duration, by["audio"].layers,
                                                   by["audio"].layer_aggregation))
        ref_data["video"].append(tl.assemble_dense([(vid.start, video_states_ref, vid.duration)], seg.start, seg.duration, by["video"].layers,
                                                   by["video"].layer_aggregation))
        assert np.array_equal(batch.data["fmri"][len(ref_data["text"]) - 1].cpu().numpy(), tl.assemble_fmri(fmri, rec.start, seg.start, seg.duration))
    data = {m: torch.from_numpy(np.stack(v)) for m, v in ref_data.items()}
    data["subject_id"] = torch.zeros(len(segs), 1, dtype=torch.long)
    for m in fdims:                                                                # the features themselves: extractor tolerance
        got = batch.data[m].unpack().cpu()
        assert got.shape == data[m].shape == (len(segs), fdims[m][0], fdims[m][1], T)
        assert (got - data[m]).norm() / data[m].norm() < 2e-2, m

    dims = tribe_ref.EncoderDims(hidden=768, depth=2, heads=4)
    ref = tribe_ref.FmriEncoderRef(fdims, V, Tout, 1, dims=dims).eval()
    with torch.no_grad():
        tribe_ref.fill_params_(ref, seed=5)
        want = ref(data)
    model = FmriEncoderConfig(n_subjects=1, hidden=768, depth=2, heads=4).build(fdims, V, Tout).eval()
    model.load_state_dict(ref.state_dict())
    got = model.cuda()(batch).cpu()
    assert got.shape == want.shape == (len(segs), V, Tout)
    err = (got - want).norm() / want.norm()
    assert err < 3e-2, f"end-to-end relative L2 error {err:.2e}"
