"""The encoder's attention launch (B = 64 x T = 1024, 8 heads x 384) on N(0, 1) and on all-zero q | k | v: is ITS rate a power figure too
(scripts/gemm_power_probe.py says the big GEMMs' is)?  Usage: python scripts/attn_power_probe.py"""
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from tribe_hip import ops  # noqa: E402

dev = torch.device("cuda")
B, T, h, d = 64, 1024, 8, 384
data = {"N(0,1) q|k|v": torch.randn(B * T, 3 * h * d, device=dev).bfloat16(), "all zeros": torch.zeros(B * T, 3 * h * d, device=dev, dtype=torch.bfloat16)}
times = {k: [] for k in data}
for rnd in range(5):
    for k, qkv in data.items():
        for _ in range(4):
            ops.attention(qkv, B, T, h, d, d**-0.5)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            ops.attention(qkv, B, T, h, d, d**-0.5)
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            times[k].append(e0.elapsed_time(e1) / 8)
fl = 4.0 * B * h * T * T * d
for k, ts in times.items():
    m = statistics.median(ts)
    print(f"  {k:14s} {m * 1e3:8.1f} us  {fl / m / 1e9:7.1f} TFLOP/s = {fl / m / 1e9 / 2500:.3f} of peak", flush=True)
