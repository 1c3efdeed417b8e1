"""Is the big GEMM's rate set by the kernel's structure or by the chip's power limit?  The FF1 / FF2 / QKV launches of the B = 64 step (plain
epilogue) on operands of different switching activity -- N(0, 1) values (the bench's synthetic features and random-init weights behave like
this), all zeros, one repeated constant -- interleaved in one process.  Same instruction stream, same addresses, same bytes moved: only
the toggling in the MFMA datapath / register file / LDS / memory buses differs.  Usage: python scripts/gemm_power_probe.py"""
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from tribe_hip import ops  # noqa: E402

dev = torch.device("cuda")
for name, M, N, K in [("ff1", 65536, 12288, 3072), ("ff2", 65536, 3072, 12288), ("qkv", 65536, 9216, 3072)]:
    data = {
        "N(0,1) operands": (torch.randn(M, K, device=dev).bfloat16(), torch.randn(N, K, device=dev).bfloat16()),
        "all zeros": (torch.zeros(M, K, device=dev, dtype=torch.bfloat16), torch.zeros(N, K, device=dev, dtype=torch.bfloat16)),
        "constant 1.0": (torch.ones(M, K, device=dev, dtype=torch.bfloat16), torch.ones(N, K, device=dev, dtype=torch.bfloat16)),
    }
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    times = {k: [] for k in data}
    for rnd in range(5):
        for k, (a, b) in data.items():
            for _ in range(4):   # let the clock settle on this operand set before timing
                ops.gemm_nt(a, b, out=out, out_dtype=torch.bfloat16)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ops.gemm_nt(a, b, out=out, out_dtype=torch.bfloat16)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[k].append(e0.elapsed_time(e1) / 8)
    fl = 2.0 * M * N * K
    print(f"== {name} {M} x {N} x {K} (bf16 out, plain epilogue, 8-wave 256 x 256 kernel)")
    for k, ts in times.items():
        m = statistics.median(ts)
        print(f"  {k:18s} {m * 1e3:9.1f} us  {fl / m / 1e9:7.1f} TFLOP/s = {fl / m / 1e9 / 2500:.3f} of peak", flush=True)
    del data, out
