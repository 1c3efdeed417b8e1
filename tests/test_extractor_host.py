"""CPU: host-side logic of the extractor mirrors (rope tables, resampling index, pooling windows) against the installed
`transformers` / torch implementations the reference relies on."""

import numpy as np
import torch

from oracle import extractors_ref


def test_rope_inv_freq_matches_transformers():
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding

    from data_utils.features.text import LLAMA_3P2_3B, rope_inv_freq

    cfg = LlamaConfig(hidden_size=256, num_attention_heads=2, head_dim=128, max_position_embeddings=131072,
                      rope_parameters=dict(LLAMA_3P2_3B["rope_parameters"]))
    hf = LlamaRotaryEmbedding(cfg)
    torch.testing.assert_close(rope_inv_freq(128, dict(cfg.rope_parameters)), hf.inv_freq.float(), rtol=1e-6, atol=0)
    assert float(hf.attention_scaling) == 1.0
    torch.testing.assert_close(rope_inv_freq(64, {"rope_type": "default", "rope_theta": 10000.0}),
                               1.0 / (10000.0 ** (torch.arange(0, 64, 2).float() / 64)))


def test_rope3d_tables_match_transformers():
    from transformers import VJEPA2Config
    from transformers.models.vjepa2.modeling_vjepa2 import VJEPA2RopeAttention

    from data_utils.features.video import rope3d_tables

    cfg = VJEPA2Config(patch_size=16, crop_size=64, frames_per_clip=8, tubelet_size=2, hidden_size=128, num_attention_heads=2,
                       num_hidden_layers=1, pred_hidden_size=64, pred_num_attention_heads=2, pred_num_hidden_layers=1)
    attn = VJEPA2RopeAttention(cfg, hidden_size=128, num_attention_heads=2)
    q = torch.randn(1, 2, 64, 64, generator=torch.Generator().manual_seed(4))
    want = attn.apply_rotary_embeddings(q, attn.get_position_ids(torch.zeros(1, 64, 128)))
    cos, sin = rope3d_tables(4, 4, 64)
    rot = torch.stack((-q[..., 1::2], q[..., 0::2]), dim=-1).flatten(-2)
    torch.testing.assert_close(q * cos + rot * sin, want, rtol=1e-5, atol=1e-6)
    assert torch.equal(cos[:, 60:], torch.ones(64, 4)) and torch.equal(sin[:, 60:], torch.zeros(64, 4))  # 64 = 3*20 + 4 pass-through


def test_nearest_index_matches_interpolate():
    from data_utils.features.audio import nearest_index

    for t_in, t_out in ((3000, 120), (333, 13), (1499, 60), (7, 20), (120, 120)):
        x = torch.arange(t_in, dtype=torch.float32)[None, None]
        want = torch.nn.functional.interpolate(x, t_out)[0, 0].to(torch.int64)  # audio.py:171
        assert torch.equal(nearest_index(t_in, t_out), want), (t_in, t_out)


def test_word_pool_windows_match_reference_slicing():
    """text.py:245-254 restated in the oracle (python slicing on the stacked states) vs the (start, length) windows."""
    from data_utils.features.text import word_pool_windows

    class _Fake(torch.nn.Module):  # hidden state l = position index + 100 * l, so means identify the window exactly
        def forward(self, input_ids, attention_mask, output_hidden_states):
            B, T = input_ids.shape
            base = torch.arange(T, dtype=torch.float32)[None, :, None].expand(B, T, 2)
            return type("O", (), {"hidden_states": tuple(base + 100.0 * l for l in range(3))})()

    pad = 7
    ids = torch.full((4, 12), 9)
    n_real = [12, 8, 3, 5]
    for i, n in enumerate(n_real):
        ids[i, n:] = pad
    words = ["hello", "", "toolongword", "ab"]
    want = extractors_ref.llama_word_states(_Fake(), ids, (ids != pad).long(), words, pad)
    start, length = word_pool_windows(ids, words, pad)
    for i in range(4):
        s, n = int(start[i]), int(length[i])
        got = np.stack([np.full(2, np.arange(s, s + n).mean() + 100.0 * l) for l in range(3)])
        np.testing.assert_allclose(got, want[i], rtol=1e-6)


def test_feature_plugin_surface_matches_the_reference_fields():
    """text.py:42-62, audio.py:27-41, video.py:56-69: pydantic models with `name` literal, `layers`, `layer_aggregation`, `device`,
    `infra`, extra='forbid'; prepare / __call__ / _get_data / _aggregate_layers; and no CPU compute path."""
    import numpy as np
    import pydantic
    import pytest

    from data_utils.events import Sound, Video, Word
    from data_utils.features.audio import Wav2VecBert
    from data_utils.features.text import LLAMA3p2
    from data_utils.features.video import VJEPA2
    from data_utils.helpers import EventTypesHelper, extract_events
    from data_utils.segments import Segment

    for cls, ev_type in ((LLAMA3p2, "Word"), (Wav2VecBert, "Sound"), (VJEPA2, "Video")):
        f = cls()
        assert f.name == cls.__name__ and f.layers == [0.5, 0.75, 1.0] and f.layer_aggregation == "group_mean"
        assert f.device in ("cpu", "cuda") and f.infra.folder is None                   # "auto" resolved as the reference does
        assert f._event_types_helper.names == [ev_type]
        assert f._exclude_from_cache_uid() == ["device", "layers", "layer_aggregation"]
        for m in ("prepare", "__call__", "_get_data", "_aggregate_layers"):
            assert callable(getattr(f, m))
        with pytest.raises(pydantic.ValidationError):
            cls(not_a_field=1)
        with pytest.raises(pydantic.ValidationError):
            cls(name="Other")
        cfg = cls(layers=[0.0, 0.5, 1.0], layer_aggregation=None, infra={"folder": "/tmp/x", "cluster": "slurm", "gpus_per_node": 1})
        lat = np.arange(9 * 4, dtype=np.float32).reshape(9, 4)
        assert np.array_equal(cfg._aggregate_layers(lat), lat[[0, 4, 8]])
        assert cfg.infra.folder == "/tmp/x"
    w, s, v = Word(start=0.0, duration=0.2, text="a"), Sound(start=0.0, duration=3.0, filepath="a.wav"), Video(start=0.0, duration=3.0, filepath="v.mkv")
    segs = [Segment(start=0.0, duration=2.0, ns_events=[w, s, v]), Segment(start=1.0, duration=2.0, ns_events=[s, v])]
    assert extract_events(segs, types="Sound") == [s] and extract_events(segs) == [w, s, v]
    assert extract_events(w, types=EventTypesHelper("Word")) == [w] and extract_events([], types="Word") == []
    assert extract_events({"type": "Word", "start": 1.0, "duration": 0.5, "text": "hi", "speaker": "x"})[0].extra == {"speaker": "x"}
    with pytest.raises(ValueError):
        EventTypesHelper("Nope")
    if Wav2VecBert().device == "cpu":
        with pytest.raises(RuntimeError, match="no CPU path"):
            Wav2VecBert().prepare([s])
    assert Wav2VecBert()._item_uid(s) == "a.wav_0.00_3.00" and LLAMA3p2()._item_uid(Word(start=0, text="a", context="b a")) == "a_b a"
