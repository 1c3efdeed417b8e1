"""bf16 vs fp8 (e4m3) MFMA GEMM at the extractor shapes.  GPU box: python scripts/fp8_microbench.py"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from tribe_hip import ops  # noqa: E402

shapes = [("llama qkv", 8192, 5120, 3072), ("llama gate_up", 8192, 16384, 3072), ("llama down", 8192, 3072, 8192),
          ("vjepa2 fc1", 16384, 6144, 1408), ("vjepa2 fc2", 16384, 1408, 6144), ("square 8192", 8192, 8192, 8192)]
for name, M, N, K in shapes:
    Kp = (K + 127) // 128 * 128
    a = torch.randn(M, Kp, device="cuda")
    b = torch.randn(N, Kp, device="cuda") * 0.05
    a16, b16 = a.bfloat16(), b.bfloat16()
    sa, sb = float(a.abs().max()) / 448, float(b.abs().max()) / 448
    a8, b8 = ops.quantize_fp8(a, sa), ops.quantize_fp8(b, sb)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    res = []
    for fn in (lambda: ops.gemm_nt(a16, b16, out=out), lambda: ops.gemm_fp8_nt(a8, b8, sa * sb, out=out)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20)
    fl = 2.0 * M * N * Kp
    print(f"{name:14s} M={M:6d} N={N:6d} K={Kp:5d}: bf16 {res[0] * 1e3:8.1f} us {fl / res[0] / 1e9:7.0f} TF | fp8 {res[1] * 1e3:8.1f} us "
          f"{fl / res[1] / 1e9:7.0f} TF ({fl / res[1] / 1e9 / 5000 * 100:4.1f} % of 5 PF)  x{res[0] / res[1]:.2f}", flush=True)
