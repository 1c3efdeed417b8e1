"""Throughput of the frozen-extractor forwards on one MI355X (full-size architectures, random weights, synthetic inputs).
GPU box: python scripts/extractor_bench.py [llama|vjepa2|w2vbert ...]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402


def timed(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def build_llama():
    """Full-size Llama-3.2-3B architecture with random weights -> (HipLlamaModel, vocab size)."""
    model, vocab, _ = _llama()
    return model, vocab


def _llama():
    from data_utils.features.text import LLAMA_3P2_3B, HipLlamaModel

    cfg = dict(LLAMA_3P2_3B)
    vocab = cfg["vocab_size"]
    H, I, L, hq, hkv, dh = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"], 24, 8, 128
    g = torch.Generator().manual_seed(0)
    sd = {"embed_tokens.weight": torch.randn(vocab, H, generator=g) * 0.02, "norm.weight": torch.ones(H)}
    for i in range(L):
        p = f"layers.{i}."
        sd[p + "self_attn.q_proj.weight"] = torch.randn(hq * dh, H, generator=g) * 0.02
        sd[p + "self_attn.k_proj.weight"] = torch.randn(hkv * dh, H, generator=g) * 0.02
        sd[p + "self_attn.v_proj.weight"] = torch.randn(hkv * dh, H, generator=g) * 0.02
        sd[p + "self_attn.o_proj.weight"] = torch.randn(H, hq * dh, generator=g) * 0.02
        sd[p + "mlp.gate_proj.weight"] = torch.randn(I, H, generator=g) * 0.02
        sd[p + "mlp.up_proj.weight"] = torch.randn(I, H, generator=g) * 0.02
        sd[p + "mlp.down_proj.weight"] = torch.randn(H, I, generator=g) * 0.02
        sd[p + "input_layernorm.weight"] = torch.ones(H)
        sd[p + "post_attention_layernorm.weight"] = torch.ones(H)
    model = HipLlamaModel(cfg, sd)
    del sd
    return model, vocab, (H, I, L, hq, hkv, dh, g)


def bench_llama():
    model, vocab, (H, I, L, hq, hkv, dh, g) = _llama()
    for B, T in ((8, 1024), (32, 1024)):
        ids = torch.randint(0, vocab, (B, T), generator=g)
        start, length = torch.full((B,), T - 5), torch.full((B,), 5)
        dt = timed(lambda: model.forward_pooled(ids, start, length))
        per_tok = L * (2 * H * (hq + 2 * hkv) * dh + 2 * hq * dh * H + 2 * 3 * H * I + 4 * (T / 2) * hq * dh)  # causal: T/2 keys on average
        print(f"llama-3.2-3B fwd+pool  B={B} T={T}: {dt * 1e3:8.2f} ms  {B * T / dt:10.0f} tok/s  {B / dt:7.1f} words/s  "
              f"{per_tok * B * T / dt / 1e12:7.1f} TFLOP/s", flush=True)
    ref = model.forward_pooled(ids, start, length)
    model.enable_fp8(torch.randint(0, vocab, (4, 1024), generator=g))           # e4m3 Linear GEMMs, static per-tensor scales
    for B, T in ((8, 1024), (32, 1024)):
        ids8 = torch.randint(0, vocab, (B, T), generator=g)
        start, length = torch.full((B,), T - 5), torch.full((B,), 5)
        dt = timed(lambda: model.forward_pooled(ids8, start, length))
        print(f"llama-3.2-3B fwd+pool  B={B} T={T} fp8 GEMMs: {dt * 1e3:8.2f} ms  {B * T / dt:10.0f} tok/s  {B / dt:7.1f} words/s  "
              f"{per_tok * B * T / dt / 1e12:7.1f} TFLOP/s", flush=True)
    got = model.forward_pooled(ids, start, length)
    err = ((got - ref).norm(dim=-1) / ref.norm(dim=-1)).mean(dim=1)
    print("  fp8 vs bf16 pooled states, mean relative L2 error at layers 1 / 14 / 28:", [round(float(err[i]), 4) for i in (1, 14, 28)], flush=True)


def _rand_sd(hf_cls, cfg_cls, kwargs):
    """Random-weight state dict of a full-size HF architecture without building the module on the host twice."""
    cfg = cfg_cls(**kwargs)
    with torch.device("meta"):
        m = hf_cls(cfg)
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, v in m.state_dict().items():
        if k.startswith("predictor."):
            continue
        if v.ndim >= 2:
            sd[k] = torch.randn(v.shape, generator=g) * 0.02
        else:
            sd[k] = torch.ones(v.shape) if "weight" in k else torch.zeros(v.shape)
    return cfg, sd


def build_vjepa2():
    """Full-size V-JEPA2 ViT-g encoder architecture with random weights."""
    from transformers import VJEPA2Config, VJEPA2Model

    from data_utils.features.video import HipVJEPA2Encoder

    cfg, sd = _rand_sd(VJEPA2Model, VJEPA2Config, dict(patch_size=16, crop_size=256, frames_per_clip=64, tubelet_size=2, hidden_size=1408,
                                                        in_chans=3, num_attention_heads=22, num_hidden_layers=40, mlp_ratio=48 / 11,
                                                        pred_hidden_size=64, pred_num_attention_heads=2, pred_num_hidden_layers=1))
    return HipVJEPA2Encoder(cfg, sd)


def bench_vjepa2():
    enc = build_vjepa2()
    H, L, mlp, tok = 1408, 40, int(1408 * 48 / 11), 8192
    per_tok = L * (2 * H * 3 * H + 2 * H * H + 4 * H * mlp + 4 * tok * H)
    for B in (1, 2, 4):
        clips = torch.randn(B, 64, 3, 256, 256, device="cuda")
        dt = timed(lambda: enc.hidden_state_means(clips), n=3, warm=1)
        print(f"vjepa2-vitg fwd+mean  clips={B} (8192 tokens each): {dt * 1e3:8.2f} ms = {dt * 1e3 / B:6.2f} ms per clip  {B / dt:6.2f} clips/s  "
              f"{per_tok * B * tok / dt / 1e12:7.1f} TFLOP/s", flush=True)
    clips = clips[:2].contiguous()
    ref = enc.hidden_state_means(clips)
    enc.enable_fp8(torch.randn(1, 64, 3, 256, 256, device="cuda"))
    for B in (1, 2):
        clips8 = torch.randn(B, 64, 3, 256, 256, device="cuda")
        dt = timed(lambda: enc.hidden_state_means(clips8), n=3, warm=1)
        print(f"vjepa2-vitg fwd+mean  clips={B} fp8 GEMMs: {dt * 1e3:8.2f} ms  {B / dt:6.2f} clips/s  "
              f"{per_tok * B * tok / dt / 1e12:7.1f} TFLOP/s", flush=True)
    got = enc.hidden_state_means(clips)
    err = ((got - ref).norm(dim=-1) / ref.norm(dim=-1)).mean(dim=0)
    print("  fp8 vs bf16 token means, relative L2 error at layers 1 / 20 / 40:", [round(float(err[i]), 4) for i in (1, 20, 40)], flush=True)


def build_w2vbert():
    """Full-size Wav2Vec-BERT 2.0 architecture with random weights."""
    from transformers import Wav2Vec2BertConfig, Wav2Vec2BertModel

    from data_utils.features.audio import HipWav2Vec2Bert

    cfg, sd = _rand_sd(Wav2Vec2BertModel, Wav2Vec2BertConfig, dict(vocab_size=None, hidden_size=1024, num_hidden_layers=24,
                                                                    num_attention_heads=16, intermediate_size=4096,
                                                                    feature_projection_input_dim=160, add_adapter=False,
                                                                    position_embeddings_type="relative_key"))
    return HipWav2Vec2Bert(cfg, sd)


def bench_w2vbert():
    model = build_w2vbert()
    H, L, I = 1024, 24, 4096
    for B, T in ((1, 3000), (8, 3000)):
        per_tok = L * (2 * 2 * 2 * H * I + 2 * H * 3 * H + 2 * H * H + 2 * H * 2 * H + 2 * H * H + 4 * T * H)
        feats = torch.randn(B, T, 160, device="cuda")
        dt = timed(lambda: model.hidden_states_resampled(feats, 120), n=3, warm=1)
        print(f"w2v-bert-2.0 fwd+resample  chunks={B} x {T} frames (60 s): {dt * 1e3:8.2f} ms  {B * 60 / dt:8.1f} audio-s/s  "
              f"{per_tok * B * T / dt / 1e12:7.1f} TFLOP/s", flush=True)
    ref = model.hidden_states_resampled(feats, 120)
    model.enable_fp8(torch.randn(1, 3000, 160, device="cuda"))              # e4m3 feed-forward GEMMs, static per-tensor scales
    for B, T in ((1, 3000), (8, 3000)):
        feats8 = torch.randn(B, T, 160, device="cuda")
        dt = timed(lambda: model.hidden_states_resampled(feats8, 120), n=3, warm=1)
        print(f"w2v-bert-2.0 fwd+resample  chunks={B} x {T} frames (60 s) fp8 FFN GEMMs: {dt * 1e3:8.2f} ms  {B * 60 / dt:8.1f} audio-s/s  "
              f"{per_tok * B * T / dt / 1e12:7.1f} TFLOP/s", flush=True)
    got = model.hidden_states_resampled(feats, 120)
    err = ((got - ref).flatten(2).norm(dim=-1) / ref.flatten(2).norm(dim=-1)).mean(dim=0)
    print("  fp8 vs bf16 hidden states, relative L2 error at layers 1 / 12 / 24:", [round(float(err[i]), 4) for i in (1, 12, 24)], flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["llama", "vjepa2", "w2vbert"]
    for w in which:
        {"llama": bench_llama, "vjepa2": bench_vjepa2, "w2vbert": bench_w2vbert}[w]()
