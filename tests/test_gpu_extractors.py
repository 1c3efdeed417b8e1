"""GPU parity of the frozen-extractor forwards against the installed `transformers` implementation (the classes the
reference instantiates), random weights from a local config -- real checkpoints are remote-only (parity unpinned for
them).  Tolerances: bf16 GEMM operands with f32 accumulation and f32 residual stream vs an fp32 CPU model."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import extractors_ref  # noqa: E402


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


def test_extractor_building_blocks():
    from tribe_hip import ops

    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 512, generator=g) * 2
    w, b = torch.rand(512, generator=g) + 0.5, torch.randn(512, generator=g)
    want = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * w  # modeling_llama.py:61-66
    torch.testing.assert_close(ops.rmsnorm(x.cuda(), w.cuda(), 1e-5, torch.float32).cpu(), want, rtol=1e-5, atol=1e-5)
    want = torch.nn.functional.layer_norm(x, (512,), w, b, 1e-6)
    torch.testing.assert_close(ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-6, torch.float32).cpu(), want, rtol=1e-4, atol=1e-5)
    table = torch.randn(50, 64, generator=g)
    ids = torch.randint(0, 50, (3, 5), generator=g)
    torch.testing.assert_close(ops.embedding(table.cuda(), ids.cuda()).cpu(), table[ids.flatten()], rtol=0, atol=0)
    seq = torch.randn(3 * 7, 64, generator=g)
    start, length = torch.tensor([0, 2, 5]), torch.tensor([7, 3, 2])
    got = ops.segment_mean(seq.cuda(), 3, 7, start.cuda(), length.cuda()).cpu()
    want = torch.stack([seq.view(3, 7, 64)[i, s:s + n].mean(0) for i, (s, n) in enumerate(zip(start.tolist(), length.tolist()))])
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    # whole-sequence means of a ViT-sized state (the 8-rows-in-flight loop, several row slices) and a width that is not a multiple of 4
    for T_, dim_ in ((2000, 1408), (333, 1408), (50, 30)):
        big = torch.randn(2 * T_, dim_, generator=g)
        torch.testing.assert_close(ops.segment_mean(big.cuda(), 2, T_, None, None).cpu(), big.view(2, T_, dim_).mean(1), rtol=1e-4, atol=1e-5)
    # SwiGLU epilogue: out[:, j] = silu(v[:, 2j]) * v[:, 2j+1]
    a, wgu = bf(torch.randn(70, 128, generator=g)), bf(torch.randn(2 * 96, 128, generator=g) / 11)
    v = a @ wgu.t()
    want = torch.nn.functional.silu(v[:, 0::2]) * v[:, 1::2]
    got = ops.gemm_nt(a.cuda().bfloat16(), wgu.cuda().bfloat16(), act="swiglu", out_dtype=torch.float32).cpu()
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,T,hq,hkv,d,causal", [(2, 70, 4, 2, 64, True), (1, 300, 6, 2, 128, True), (2, 129, 3, 1, 128, False),
                                                   (1, 1024, 24, 8, 128, True)])
def test_attention_gqa_causal(B, T, hq, hkv, d, causal):
    from tribe_hip import ops

    g = torch.Generator().manual_seed(1)
    width = (hq + 2 * hkv) * d
    qkv = bf(torch.randn(B * T, width, generator=g))
    out = ops.attention_gqa(qkv.cuda().bfloat16(), B, T, hq, hkv, d, d**-0.5, causal).float().cpu()
    q = qkv[:, : hq * d].view(B, T, hq, d).transpose(1, 2)
    k = qkv[:, hq * d:(hq + hkv) * d].view(B, T, hkv, d).transpose(1, 2).repeat_interleave(hq // hkv, dim=1)  # repeat_kv
    v = qkv[:, (hq + hkv) * d:].view(B, T, hkv, d).transpose(1, 2).repeat_interleave(hq // hkv, dim=1)
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * d**-0.5
    if causal:
        sim = sim.masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool), 1), float("-inf"))
    want = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v).transpose(1, 2).reshape(B * T, hq * d)
    torch.testing.assert_close(out, want, rtol=2**-6, atol=4e-3)


def _tiny_llama(layers=3, hidden=256, heads=4, kv=2, head_dim=64, inter=512, vocab=300):
    from transformers import LlamaConfig, LlamaModel

    cfg = LlamaConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                      num_attention_heads=heads, num_key_value_heads=kv, head_dim=head_dim, max_position_embeddings=16384,
                      rms_norm_eps=1e-5, tie_word_embeddings=True,
                      rope_parameters={"rope_type": "llama3", "rope_theta": 500000.0, "factor": 32.0, "low_freq_factor": 1.0,
                                       "high_freq_factor": 4.0, "original_max_position_embeddings": 8192})
    torch.manual_seed(0)
    return cfg, LlamaModel(cfg).eval()


def test_rope_inv_freq_matches_hf():
    from data_utils.features.text import rope_inv_freq

    cfg, hf = _tiny_llama()
    torch.testing.assert_close(rope_inv_freq(64, dict(cfg.rope_parameters)), hf.rotary_emb.inv_freq.float().cpu(), rtol=1e-6, atol=0)
    torch.testing.assert_close(rope_inv_freq(128, {"rope_type": "default", "rope_theta": 10000.0}),
                               1.0 / (10000.0 ** (torch.arange(0, 128, 2).float() / 128)))


@pytest.mark.parametrize("shape", ["tiny", "wide"])
def test_llama_word_states_vs_transformers(shape):
    """text.py:204-256 semantics end to end: HF LlamaModel (fp32, CPU) + the reference's pad-strip / last-len(word) mean
    vs tribe_llama_fwd.  'wide' uses the real Llama-3.2-3B widths (3072, 24/8 heads x 128, MLP 8192) on 2 layers."""
    from data_utils.features.text import HipLlamaModel, word_pool_windows

    if shape == "tiny":
        cfg, hf = _tiny_llama()
    else:
        cfg, hf = _tiny_llama(layers=2, hidden=3072, heads=24, kv=8, head_dim=128, inter=8192, vocab=512)
    pad_id = 7
    g = torch.Generator().manual_seed(3)
    B, T = 4, 45
    ids = torch.randint(8, cfg.vocab_size, (B, T), generator=g)
    n_real = [45, 30, 12, 3]
    mask = torch.ones(B, T, dtype=torch.long)
    for i, n in enumerate(n_real):
        ids[i, n:] = pad_id  # right padding with the eos/pad id (text.py:182-183)
        mask[i, n:] = 0
    words = ["hello", "a", "extraordinarily", "toolongword"]  # the last window is longer than its 3-token context
    want = extractors_ref.llama_word_states(hf, ids, mask, words, pad_id)
    model = HipLlamaModel(cfg, hf.state_dict())
    start, length = word_pool_windows(ids, words, pad_id)
    got = model.forward_pooled(ids, start, length).cpu().numpy()  # [n_states, B, dim]
    assert got.shape == (cfg.num_hidden_layers + 1, B, cfg.hidden_size)
    for j in range(B):
        assert want[j].shape == got[:, j].shape
        np.testing.assert_allclose(got[0, j], want[j][0], rtol=0, atol=4e-3 * np.abs(want[j][0]).max() + 1e-6)  # bf16 embedding table
        err = _rel(got[:, j], want[j])
        assert err < 1.5e-2, f"word {j}: relative L2 error {err:.2e}"


def _fake_quant(t: torch.Tensor, scale: float) -> torch.Tensor:
    """e4m3 round trip with a per-tensor scale, as tribe_quantize_fp8_fwd does it (clamp to +-448, round to nearest even)."""
    return (t.float() * np.float32(1.0 / scale)).clamp(-448, 448).to(torch.float8_e4m3fn).float() * np.float32(scale)


def test_llama_fp8_gemms_vs_quantisation_aware_reference():
    """BASELINE config 5: the four Linear GEMMs of every decoder layer in e4m3 with per-tensor scales (weights: amax / 448;
    inputs: static, from one bf16 calibration pass).  Oracle: the fp32 transformers model with the SAME quantisation
    emulated on the CPU (inputs and weights of the seven Linear modules round-tripped through torch.float8_e4m3fn with this
    build's scales).  The kernels themselves are pinned exactly elsewhere (tests/test_gpu_fp8.py: quantiser bit-exact vs
    torch, GEMM 1e-4 vs the dequantised product); end to end a quantiser is discontinuous, so the ~0.5 % difference between
    this build's bf16 intermediates and the fp32 reference flips the rounding of ~8 % of the values by one e4m3 step at every
    quantised input (RMS ~ sqrt(0.005 * 0.06) = 1.7 % each, eight of them in two layers).  Bar: pooled hidden states within
    10 % relative L2 of the quantisation-aware reference (measured 7 %), and closer to it than to the unquantised model.
    Against the UNquantised fp32 model the error is the format's own: about 4.5 % per GEMM on zero-mean random operands
    (3 mantissa bits on both operands), ~12 % after one layer here -- which is why fp8 is opt-in (`enable_fp8`) and the
    bf16 path stays the default."""
    from data_utils.features.text import HipLlamaModel, word_pool_windows

    cfg, hf = _tiny_llama(layers=2, hidden=3072, heads=24, kv=8, head_dim=128, inter=8192, vocab=512)
    pad_id = 7
    g = torch.Generator().manual_seed(5)
    B, T = 4, 64
    ids = torch.randint(8, cfg.vocab_size, (B, T), generator=g)
    mask = torch.ones(B, T, dtype=torch.long)
    for i, n in enumerate([64, 40, 17, 5]):
        ids[i, n:] = pad_id
        mask[i, n:] = 0
    words = ["hello", "a", "extraordinarily", "word"]
    exact = extractors_ref.llama_word_states(hf, ids, mask, words, pad_id)
    model = HipLlamaModel(cfg, hf.state_dict())
    start, length = word_pool_windows(ids, words, pad_id)
    bf16 = model.forward_pooled(ids, start, length).cpu().numpy()
    with pytest.raises(ValueError):
        model.forward_pooled(ids, start, length, fp8=True)                              # before calibration
    table = model.enable_fp8(ids)
    assert table.shape == (2, 4) and bool((table > 0).all())
    got = model.forward_pooled(ids, start, length).cpu().numpy()                        # fp8 by default once enabled
    again = model.forward_pooled(ids, start, length, fp8=False).cpu().numpy()
    assert np.array_equal(again, bf16)                                                  # the bf16 path is untouched
    # quantisation-aware reference: same scales, emulated on the fp32 CPU model
    hooks = []
    for i, layer in enumerate(hf.layers):
        F = model.fp8_layers[i]
        groups = [(0, [layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj]), (1, [layer.self_attn.o_proj]),
                  (2, [layer.mlp.gate_proj, layer.mlp.up_proj]), (3, [layer.mlp.down_proj])]
        for j, mods in groups:
            for m in mods:
                m.weight.data = _fake_quant(m.weight.data.bfloat16(), float(F.w_scale[j]))
                hooks.append(m.register_forward_pre_hook(lambda mod, args, s=float(F.in_scale[j]): (_fake_quant(args[0], s),)))
    want = extractors_ref.llama_word_states(hf, ids, mask, words, pad_id)
    for h in hooks:
        h.remove()
    for j in range(B):
        err = _rel(got[:, j], want[j])
        assert err < 1e-1, f"word {j}: relative L2 error vs the quantisation-aware reference {err:.2e}"
        assert err < _rel(got[:, j], exact[j]) < 0.25                                   # the format's own noise, see docstring
    tiny_cfg, tiny = _tiny_llama()
    if tiny_cfg.hidden_size % 128:
        with pytest.raises(ValueError):
            HipLlamaModel(tiny_cfg, tiny.state_dict()).enable_fp8(ids[:1] % tiny_cfg.vocab_size)


def _tiny_vjepa2(hidden=128, heads=2, layers=3, mlp_ratio=4.0, crop=64, frames=8):
    from transformers import VJEPA2Config, VJEPA2Model

    cfg = VJEPA2Config(patch_size=16, crop_size=crop, frames_per_clip=frames, tubelet_size=2, hidden_size=hidden, in_chans=3,
                       num_attention_heads=heads, num_hidden_layers=layers, mlp_ratio=mlp_ratio, pred_hidden_size=64,
                       pred_num_attention_heads=2, pred_num_hidden_layers=1, pred_num_mask_tokens=2)
    torch.manual_seed(0)
    return cfg, VJEPA2Model(cfg).eval()


@pytest.mark.parametrize("shape", ["tiny", "vitg_width"])
def test_vjepa2_hidden_state_means_vs_transformers(shape):
    """video.py:262-268 (stack of output_hidden_states) + :228 (mean over tokens) vs tribe_vjepa2_fwd.
    'vitg_width' uses the ViT-g widths (1408, 22 heads x 64, MLP 6144) on 2 layers and a small clip."""
    from data_utils.features.video import HipVJEPA2Encoder

    cfg, hf = _tiny_vjepa2() if shape == "tiny" else _tiny_vjepa2(hidden=1408, heads=22, layers=2, mlp_ratio=48 / 11, crop=96, frames=4)
    g = torch.Generator().manual_seed(2)
    clips = torch.randn(2, cfg.frames_per_clip, 3, cfg.crop_size, cfg.crop_size, generator=g)
    with torch.no_grad():
        out = hf(pixel_values_videos=clips, output_hidden_states=True, skip_predictor=True)
    want = torch.cat([x.unsqueeze(1) for x in out.hidden_states], dim=1).mean(dim=2)  # [B, n_states, dim]
    enc = HipVJEPA2Encoder(cfg, hf.state_dict())
    got = enc.hidden_state_means(clips).cpu()
    assert got.shape == want.shape == (2, cfg.num_hidden_layers + 1, cfg.hidden_size)
    for s in range(want.shape[1]):
        err = _rel(got[:, s], want[:, s])
        assert err < 2e-2, f"state {s}: relative L2 error {err:.2e}"


def test_vjepa2_fp8_gemms_vs_quantisation_aware_reference():
    """e4m3 Linear GEMMs in the ViT-g-width encoder vs the fp32 transformers model with the same quantisation emulated on the
    CPU (see test_llama_fp8_gemms_vs_quantisation_aware_reference for the bar and its reasoning)."""
    from data_utils.features.video import HipVJEPA2Encoder

    cfg, hf = _tiny_vjepa2(hidden=1408, heads=22, layers=2, mlp_ratio=48 / 11, crop=96, frames=4)
    g = torch.Generator().manual_seed(4)
    clips = torch.randn(2, cfg.frames_per_clip, 3, cfg.crop_size, cfg.crop_size, generator=g)
    enc = HipVJEPA2Encoder(cfg, hf.state_dict())
    bf16 = enc.hidden_state_means(clips).cpu()
    table = enc.enable_fp8(clips)
    assert table.shape == (2, 4) and bool((table > 0).all())
    got = enc.hidden_state_means(clips).cpu()
    assert torch.equal(enc.hidden_state_means(clips, fp8=False).cpu(), bf16) and not torch.equal(got, bf16)
    hooks = []
    for i, layer in enumerate(hf.encoder.layer):
        F = enc.fp8_layers[i]
        groups = [(0, [layer.attention.query, layer.attention.key, layer.attention.value]), (1, [layer.attention.proj]),
                  (2, [layer.mlp.fc1]), (3, [layer.mlp.fc2])]
        for j, mods in groups:
            for m in mods:
                m.weight.data = _fake_quant(m.weight.data.bfloat16(), float(F.w_scale[j]))
                hooks.append(m.register_forward_pre_hook(lambda mod, args, s=float(F.in_scale[j]): (_fake_quant(args[0], s),)))
    with torch.no_grad():
        out = hf(pixel_values_videos=clips, output_hidden_states=True, skip_predictor=True)
    for h in hooks:
        h.remove()
    want = torch.cat([x.unsqueeze(1) for x in out.hidden_states], dim=1).mean(dim=2)
    for s_ in range(want.shape[1]):
        err = _rel(got[:, s_], want[:, s_])
        assert err < 1e-1, f"state {s_}: relative L2 error vs the quantisation-aware reference {err:.2e}"


def test_rope3d_tables_match_transformers():
    """The per-element tables reproduce VJEPA2RopeAttention.apply_rotary_embeddings (modeling_vjepa2.py:279-294)."""
    from data_utils.features.video import rope3d_tables

    cfg, hf = _tiny_vjepa2()
    attn = hf.encoder.layer[0].attention
    g = torch.Generator().manual_seed(4)
    q = torch.randn(1, 2, 64, 64, generator=g)  # [B, heads, tokens, dim_head]
    want = attn.apply_rotary_embeddings(q, attn.get_position_ids(torch.zeros(1, 64, 128)))
    cos, sin = rope3d_tables(4, 4, 64)
    rot = torch.stack((-q[..., 1::2], q[..., 0::2]), dim=-1).flatten(-2)
    torch.testing.assert_close(q * cos + rot * sin, want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,T,Cc,K", [(2, 300, 1024, 31), (1, 7, 128, 31), (3, 50, 256, 5), (1, 20, 2048, 31)])
def test_depthwise_conv_layernorm_swish(B, T, Cc, K):
    """tribe_dwconv_ln_swish_fwd vs torch (causal depthwise Conv1d, LayerNorm over channels, SiLU): the 8-steps-per-workgroup kernel
    (C <= 1024, K <= 31; ragged last tile at T = 300 / 7 / 50, K = 5 leaves 26 zero tap slots) and the one-row kernel (C = 2048)."""
    from tribe_hip import ops

    g = torch.Generator().manual_seed(16)
    x = bf(torch.randn(B, T, Cc, generator=g))
    w = torch.randn(K, Cc, generator=g) / K**0.5
    ln_w, ln_b = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    conv = torch.nn.functional.conv1d(torch.nn.functional.pad(x.transpose(1, 2), (K - 1, 0)), w.t().unsqueeze(1), groups=Cc).transpose(1, 2)
    want = torch.nn.functional.silu(torch.nn.functional.layer_norm(conv, (Cc,), ln_w, ln_b, 1e-5))
    got = ops.dwconv_ln_swish(x.reshape(B * T, Cc).cuda().bfloat16(), B, T, w.cuda(), ln_w.cuda(), ln_b.cuda(), 1e-5).float().cpu().view(B, T, Cc)
    torch.testing.assert_close(got, want, rtol=2**-7, atol=2e-2)


def _tiny_w2vbert(hidden=128, heads=2, layers=2, inter=256):
    from transformers import Wav2Vec2BertConfig, Wav2Vec2BertModel

    cfg = Wav2Vec2BertConfig(vocab_size=None, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                             intermediate_size=inter, feature_projection_input_dim=160, hidden_act="swish",
                             position_embeddings_type="relative_key", left_max_position_embeddings=64,
                             right_max_position_embeddings=8, conv_depthwise_kernel_size=31, add_adapter=False,
                             use_intermediate_ffn_before_adapter=False, layerdrop=0.0, apply_spec_augment=False)
    torch.manual_seed(0)
    m = Wav2Vec2BertModel(cfg).eval()
    with torch.no_grad():  # zero-initialised in HF; make the relative-position path carry signal
        for layer in m.encoder.layers:
            layer.self_attn.distance_embedding.weight.normal_(0, 0.5)
    return cfg, m


@pytest.mark.parametrize("shape", ["tiny", "w2v_width"])
def test_w2vbert_hidden_states_vs_transformers(shape):
    """audio.py:253-263 (stack + transpose of output_hidden_states) and :163-171 (nearest F.interpolate to 2 Hz) vs
    tribe_w2vbert_fwd.  'w2v_width' uses the w2v-bert-2.0 widths (1024, 16 heads x 64, FFN 4096) on 2 layers."""
    from data_utils.features.audio import HipWav2Vec2Bert

    cfg, hf = _tiny_w2vbert() if shape == "tiny" else _tiny_w2vbert(hidden=1024, heads=16, layers=2, inter=4096)
    g = torch.Generator().manual_seed(6)
    T, n_out = 333, 13
    feats = torch.randn(1, T, 160, generator=g)
    with torch.no_grad():
        out = hf(feats, output_hidden_states=True)
    stacked = torch.stack(out.hidden_states).squeeze(1).transpose(-1, -2)  # [n_states, dim, T]  (audio.py:257-261)
    want = torch.nn.functional.interpolate(stacked, n_out)  # audio.py:171
    model = HipWav2Vec2Bert(cfg, hf.state_dict())
    got = model.hidden_states_resampled(feats, n_out).cpu()[0]
    assert got.shape == want.shape == (cfg.num_hidden_layers + 1, cfg.hidden_size, n_out)
    for s in range(want.shape[0]):
        err = _rel(got[s], want[s])
        assert err < 2e-2, f"state {s}: relative L2 error {err:.2e}"


def test_nearest_index_matches_interpolate():
    from data_utils.features.audio import nearest_index

    for t_in, t_out in ((3000, 120), (333, 13), (1499, 60), (7, 20)):
        x = torch.arange(t_in, dtype=torch.float32)[None, None]
        want = torch.nn.functional.interpolate(x, t_out)[0, 0].to(torch.int64)
        assert torch.equal(nearest_index(t_in, t_out), want), (t_in, t_out)


# ---- parity at the REAL layer counts ------------------------------------------------------------------------------------
# The tests above stop at 2-3 layers; the features the encoder consumes are group means over the upper half of 28 / 24 / 40
# layers (layers [0.5, 0.75, 1.0], text.py:129-149), so the bf16 kernels' error has to stay small through the whole stack.
# Depth is the real one, random weights at the libraries' default init; widths and sequences are reduced so that the fp32
# transformers model on the CPU (the reference route) runs in seconds.  Every state is checked and the growth is printed
# (pytest -s) -- scripts/extractor_bench.py records the same curve at full width against its own f32-free bf16 baseline.
def _depth_curve(got, want):
    return [_rel(got[s], want[s]) for s in range(len(want))]


def test_w2vbert_fp8_ffn_gemms_vs_quantisation_aware_reference():
    """e4m3 feed-forward GEMMs of the Conformer layers (the four FFN Linears; attention and convolution projections stay bf16) vs the fp32
    transformers model with the same quantisation emulated on the CPU (see test_llama_fp8_gemms_vs_quantisation_aware_reference for the
    bar and its reasoning).  The half-step's 0.5 rides in the GEMM's alpha on top of the two scales."""
    from data_utils.features.audio import HipWav2Vec2Bert

    cfg, hf = _tiny_w2vbert(hidden=1024, heads=16, layers=2, inter=4096)
    g = torch.Generator().manual_seed(7)
    T, n_out = 300, 11
    feats = torch.randn(2, T, 160, generator=g)
    model = HipWav2Vec2Bert(cfg, hf.state_dict())
    bf16 = model.hidden_states_resampled(feats, n_out).cpu()
    table = model.enable_fp8(feats)
    assert table.shape == (2, 4) and bool((table > 0).all())
    got = model.hidden_states_resampled(feats, n_out).cpu()
    assert torch.equal(model.hidden_states_resampled(feats, n_out, fp8=False).cpu(), bf16) and not torch.equal(got, bf16)
    hooks = []
    for i, layer in enumerate(hf.encoder.layers):
        F = model.fp8_layers[i]
        for j, m in enumerate((layer.ffn1.intermediate_dense, layer.ffn1.output_dense, layer.ffn2.intermediate_dense, layer.ffn2.output_dense)):
            m.weight.data = _fake_quant(m.weight.data.bfloat16(), float(F.w_scale[j]))
            hooks.append(m.register_forward_pre_hook(lambda mod, args, s=float(F.in_scale[j]): (_fake_quant(args[0], s),)))
    with torch.no_grad():
        out = hf(feats, output_hidden_states=True)
    for h in hooks:
        h.remove()
    want = torch.nn.functional.interpolate(torch.stack(out.hidden_states, dim=1).flatten(0, 1).transpose(-1, -2), n_out)
    want = want.view(2, cfg.num_hidden_layers + 1, cfg.hidden_size, n_out)
    for s_ in range(want.shape[1]):
        err = _rel(got[:, s_], want[:, s_])
        assert err < 1e-1, f"state {s_}: relative L2 error vs the quantisation-aware reference {err:.2e}"
    assert _rel(got[:, -1], bf16[:, -1]) < 1e-1     # and the e4m3 states stay within 10 % of the bf16 ones at this depth


def test_llama_parity_at_28_layers():
    from data_utils.features.text import HipLlamaModel, word_pool_windows

    cfg, hf = _tiny_llama(layers=28, hidden=1024, heads=8, kv=4, head_dim=128, inter=2048, vocab=400)
    pad_id, B, T = 7, 3, 96
    g = torch.Generator().manual_seed(21)
    ids = torch.randint(8, cfg.vocab_size, (B, T), generator=g)
    mask = torch.ones(B, T, dtype=torch.long)
    for i, n in enumerate([96, 61, 17]):
        ids[i, n:] = pad_id
        mask[i, n:] = 0
    words = ["alpha", "be", "gamma-delta"]
    want = np.stack(extractors_ref.llama_word_states(hf, ids, mask, words, pad_id), axis=1)        # [n_states, B, dim]
    model = HipLlamaModel(cfg, hf.state_dict())
    start, length = word_pool_windows(ids, words, pad_id)
    got = model.forward_pooled(ids, start, length).cpu().numpy()
    curve = _depth_curve(got, want)
    print("llama 28 layers, per-state relative L2:", [f"{e:.1e}" for e in curve[::4]])
    assert got.shape == want.shape == (29, B, 1024)
    assert max(curve) < 2.5e-2, f"worst state {int(np.argmax(curve))}: {max(curve):.2e}"
    # what the model consumes: the two layer groups of layers = [0.5, 0.75, 1.0] -> means over states 14..20 and 21..28
    for lo, hi in ((14, 21), (21, 29)):
        assert _rel(got[lo:hi].mean(0), want[lo:hi].mean(0)) < 2e-2


def test_w2vbert_parity_at_24_layers():
    from data_utils.features.audio import HipWav2Vec2Bert

    cfg, hf = _tiny_w2vbert(hidden=512, heads=8, layers=24, inter=1024)
    g = torch.Generator().manual_seed(22)
    T, n_out = 250, 10
    feats = torch.randn(1, T, 160, generator=g)
    with torch.no_grad():
        out = hf(feats, output_hidden_states=True)
    want = torch.nn.functional.interpolate(torch.stack(out.hidden_states).squeeze(1).transpose(-1, -2), n_out).numpy()
    got = HipWav2Vec2Bert(cfg, hf.state_dict()).hidden_states_resampled(feats, n_out).cpu().numpy()[0]
    curve = _depth_curve(got, want)
    print("w2v-bert 24 layers, per-state relative L2:", [f"{e:.1e}" for e in curve[::4]])
    assert got.shape == want.shape == (25, 512, n_out)
    # measured: 0.24 % after the feature projection, then ~0.12 % per conformer layer (four bf16 GEMM groups, a convolution and an
    # attention each; the test's distance embeddings are drawn with std 0.5 so that the relative-position path carries signal),
    # 3.1 % at state 24 -- linear growth, no blow-up
    assert max(curve) < 4e-2, f"worst state {int(np.argmax(curve))}: {max(curve):.2e}"
    assert all(b < a + 6e-3 for a, b in zip(curve[1:], curve[:-1])), "error should grow smoothly with depth"
    for lo, hi in ((12, 18), (18, 25)):
        assert _rel(got[lo:hi].mean(0), want[lo:hi].mean(0)) < 3e-2


def test_vjepa2_parity_at_40_layers():
    from data_utils.features.video import HipVJEPA2Encoder

    cfg, hf = _tiny_vjepa2(hidden=384, heads=6, layers=40, mlp_ratio=4.0, crop=64, frames=8)
    g = torch.Generator().manual_seed(23)
    clips = torch.randn(1, cfg.frames_per_clip, 3, cfg.crop_size, cfg.crop_size, generator=g)
    with torch.no_grad():
        out = hf(pixel_values_videos=clips, output_hidden_states=True, skip_predictor=True)
    want = torch.cat([x.unsqueeze(1) for x in out.hidden_states], dim=1).mean(dim=2)[0].numpy()   # [n_states, dim]
    got = HipVJEPA2Encoder(cfg, hf.state_dict()).hidden_state_means(clips).cpu().numpy()[0]
    curve = _depth_curve(got, want)
    print("v-jepa2 40 layers, per-state relative L2:", [f"{e:.1e}" for e in curve[::5]])
    assert got.shape == want.shape == (41, 384)
    assert max(curve) < 4e-2, f"worst state {int(np.argmax(curve))}: {max(curve):.2e}"
    for lo, hi in ((20, 30), (30, 41)):
        assert _rel(got[lo:hi].mean(0), want[lo:hi].mean(0)) < 2.5e-2
