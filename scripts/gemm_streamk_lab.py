"""Weight-gradient GEMMs (transposed-operand form, reduction over the B*T rows) with and without the stream-K schedule, interleaved in one
process: dW = dY^T X for the four Linear layers of an encoder block at B = 16 x T = 1024 (and B = 4).  Times include the zero fill of dW that
the split form needs.  Usage: python scripts/gemm_streamk_lab.py"""
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from tribe_hip import ops  # noqa: E402

dev = torch.device("cuda")
for name, n_out, k_in, rows in [("qkv", 9216, 3072, 16384), ("out_proj", 3072, 3072, 16384), ("ff1", 12288, 3072, 16384), ("ff2", 3072, 12288, 16384),
                                ("qkv", 9216, 3072, 4096), ("out_proj", 3072, 3072, 4096), ("ff1", 12288, 3072, 4096), ("projector", 1024, 4096, 16384)]:
    dy = torch.randn(rows, n_out, device=dev).bfloat16()
    x = torch.randn(rows, k_in, device=dev).bfloat16()
    times = {False: [], True: []}
    for rnd in range(6):
        for sk in (False, True):
            ops.gemm_tn(dy, x, stream_k=sk)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm_tn(dy, x, stream_k=sk)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[sk].append(e0.elapsed_time(e1) / 5)
    fl = 2.0 * n_out * k_in * rows
    a, b = statistics.median(times[False]), statistics.median(times[True])
    tiles = ((n_out + 255) // 256) * ((k_in + 255) // 256)
    print(f"dW {name:9s} [{n_out} x {k_in}] over {rows} rows ({tiles} tiles = {tiles / 256:.2f} rounds): whole tiles {a * 1e3:8.1f} us {fl / a / 1e9:7.1f} TFLOP/s   "
          f"stream-K {b * 1e3:8.1f} us {fl / b / 1e9:7.1f} TFLOP/s  ({'split' if ops.gemm_tn.last_split else 'not split'})", flush=True)
    del dy, x
