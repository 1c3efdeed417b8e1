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


def bench_llama():
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
    for B, T in ((8, 1024), (32, 1024)):
        ids = torch.randint(0, vocab, (B, T), generator=g)
        start, length = torch.full((B,), T - 5), torch.full((B,), 5)
        dt = timed(lambda: model.forward_pooled(ids, start, length))
        per_tok = L * (2 * H * (hq + 2 * hkv) * dh + 2 * hq * dh * H + 2 * 3 * H * I + 4 * (T / 2) * hq * dh)  # causal: T/2 keys on average
        print(f"llama-3.2-3B fwd+pool  B={B} T={T}: {dt * 1e3:8.2f} ms  {B * T / dt:10.0f} tok/s  {B / dt:7.1f} words/s  "
              f"{per_tok * B * T / dt / 1e12:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["llama"]
    for w in which:
        {"llama": bench_llama}[w]()
