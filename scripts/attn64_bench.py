"""Extractor attention (dim_head 64) in isolation: the 64-row-per-wave kernel (mode 0, attention_d64.hip) against the 16-row
kernel (mode 2), on the V-JEPA2 ViT-g shape (8192 tokens, 22 heads) and the Wav2Vec-BERT shape (3000 frames, 16 heads,
relative_key bias with left 64 / right 8).  GPU box:  python scripts/attn64_bench.py [out.json]
Algorithmic work: 4 * T * heads * 64 flop per query row, HIP events on the launch stream, 20 launches."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from tribe_hip import ops  # noqa: E402

D = 64
out = {}
g = torch.Generator(device="cuda").manual_seed(0)
for name, B, T, H, rel in (("vjepa2 clips=1", 1, 8192, 22, False), ("vjepa2 clips=2", 2, 8192, 22, False),
                           ("w2vbert chunks=1", 1, 3000, 16, True), ("w2vbert chunks=8", 8, 3000, 16, True)):
    qkv = torch.randn(B * T, 3 * H * D, generator=g, device="cuda").bfloat16()
    qe = torch.randn(B * T, H, 80, generator=g, device="cuda") if rel else None
    flop = 4.0 * T * H * D * B * T
    ref = None
    for mode, label in ((2, "16-row waves"), (4, "64-row waves x 4"), (5, "64-row wave pairs"), (0, "auto")):
        ops.attention_set_mode(mode)
        run = (lambda: ops.attention_relative_key(qkv, B, T, H, D, D**-0.5, qe, 64, 8)) if rel else (lambda: ops.attention(qkv, B, T, H, D, D**-0.5))
        for _ in range(3):
            y = run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            y = run()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        if ref is None:
            ref = y.float()
        err = float((y.float() - ref).abs().max())
        out.setdefault(name, {"B": B, "T": T, "heads": H, "gflop": flop / 1e9})[label] = {
            "ms": round(ms, 4), "tflops": round(flop / ms / 1e9, 1), "frac_of_2500": round(flop / ms / 1e9 / 2500, 4), "max_abs_diff_vs_16row": err}
        print(f"{name:18s} {label:18s} {ms:8.4f} ms  {flop / ms / 1e9:7.1f} TFLOP/s  {flop / ms / 1e9 / 2500:6.1%} of peak   max|diff| {err:.2e}", flush=True)
ops.attention_set_mode(0)
if len(sys.argv) > 1:
    Path(sys.argv[1]).write_text(json.dumps(out, indent=1))
