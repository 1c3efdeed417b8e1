"""Encoder attention (dim_head 384, 8 heads, T = 1024) in isolation: the 32-row one-wave-per-SIMD kernel (mode 0) against the
16-row kernel (mode 2).  GPU box:  python scripts/attn_bench.py [B] [out.json]
Algorithmic work: 4 * T * heads * dim_head flop per query row (QK^T + PV), HIP events on the launch stream, 20 launches."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from tribe_hip import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T, H, D = 1024, 8, 384
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * T, 3 * H * D, generator=g, device="cuda").bfloat16()
flop = 4.0 * T * H * D * B * T
out = {"B": B, "T": T, "heads": H, "dim_head": D, "gflop": flop / 1e9, "modes": {}}
ref = None
for mode, name in ((2, "16-row waves (round 1)"), (0, "32-row waves, one per SIMD"), (3, "key-split pairs, two per SIMD")):
    ops.attention_set_mode(mode)
    for _ in range(3):
        y = ops.attention(qkv, B, T, H, D, D**-0.5)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        y = ops.attention(qkv, B, T, H, D, D**-0.5)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    if ref is None:
        ref = y.float()
    err = float((y.float() - ref).abs().max())
    out["modes"][name] = {"ms": round(ms, 4), "tflops": round(flop / ms / 1e9, 1), "frac_of_2500": round(flop / ms / 1e9 / 2500, 4),
                          "max_abs_diff_vs_16row": err}
    print(f"{name:32s} {ms:8.4f} ms  {flop / ms / 1e9:7.1f} TFLOP/s  {flop / ms / 1e9 / 2500:6.1%} of peak   max|diff| {err:.2e}")
ops.attention_set_mode(0)
if len(sys.argv) > 2:
    Path(sys.argv[2]).write_text(json.dumps(out, indent=1))
