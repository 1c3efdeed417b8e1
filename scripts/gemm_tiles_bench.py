"""Tile-kernel choice per GEMM shape: times every tile kernel (tile_hint 1 = 128^2 double-buffered, 2 = 256^2, 3 = 128^2 ring, 4 = 256 x 192)
against the launcher's automatic choice (hint 0; the one-wave-per-SIMD kernel exists for the four encoder roles only and has its own A/B, scripts/gemm_4w_lab.py) on the shapes of the BASELINE config at B = 4 (M = 4096) and of the extractors,
interleaved in one process on random operands.  Usage (GPU box): python scripts/gemm_tiles_bench.py > gpurun_out/gemm_tiles.txt"""
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from tribe_hip import ops  # noqa: E402

dev = torch.device("cuda")
SHAPES = [  # name, batch, M, N, K, out dtype
    ("projector B=4", 1, 4096, 1024, 4096, torch.float32), ("voxel head B=4", 4, 1024, 1000, 3072, torch.float32),
    ("qkv B=4", 1, 4096, 9216, 3072, torch.bfloat16), ("out-proj B=4", 1, 4096, 3072, 3072, torch.float32),
    ("ff1 B=4", 1, 4096, 12288, 3072, torch.bfloat16), ("ff2 B=4", 1, 4096, 3072, 12288, torch.float32),
    ("qkv B=16", 1, 16384, 9216, 3072, torch.bfloat16), ("out-proj B=16", 1, 16384, 3072, 3072, torch.float32),
    ("vit proj 8192x1408x1408", 1, 8192, 1408, 1408, torch.bfloat16), ("vit qkv 8192x4224x1408", 1, 8192, 4224, 1408, torch.bfloat16),
    ("vit fc1 8192x6144x1408", 1, 8192, 6144, 1408, torch.bfloat16), ("vit fc2 8192x1408x6144", 1, 8192, 1408, 6144, torch.float32),
    ("w2v 3000x1024x1024", 1, 3000, 1024, 1024, torch.bfloat16), ("w2v ffn 24000x4096x1024", 1, 24000, 4096, 1024, torch.bfloat16),
    ("llama qkv 8192x5120x3072", 1, 8192, 5120, 3072, torch.bfloat16), ("llama down 8192x3072x8192", 1, 8192, 3072, 8192, torch.float32),
    ("llama 1024 tok 1024x3072x3072", 1, 1024, 3072, 3072, torch.float32),
    ("ff1 B=64", 1, 65536, 12288, 3072, torch.bfloat16), ("ff2 B=64", 1, 65536, 3072, 12288, torch.float32), ("8192^3", 1, 8192, 8192, 8192, torch.bfloat16),
]
for name, Z, M, N, K, odt in SHAPES:
    Kp = (K + 63) // 64 * 64
    a = torch.randn(Z, M, Kp, device=dev).bfloat16()
    b = torch.randn(Z, N, Kp, device=dev).bfloat16()
    if Z == 1:
        a, b = a[0], b[0]
    out = torch.empty((Z, M, N) if Z > 1 else (M, N), device=dev, dtype=odt)
    hints = [0, 1, 2, 3, 4]
    times = {h: [] for h in hints}
    reps = 10
    for rnd in range(5):
        for h in hints:
            ops.gemm_nt(a, b, out=out, tile_hint=h)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.gemm_nt(a, b, out=out, tile_hint=h)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[h].append(e0.elapsed_time(e1) / reps * 1e3)
    med = {h: statistics.median(times[h]) for h in hints}
    fl = 2.0 * Z * M * N * K
    best = min((h for h in hints if h), key=lambda h: med[h])
    print(f"{name:32s} auto {med[0]:8.1f} us ({fl / med[0] / 1e6:6.0f} TF) | dbuf128 {med[1]:8.1f}  256^2 {med[2]:8.1f}  ring128 {med[3]:8.1f}  256x192 {med[4]:8.1f} | "
          f"best = hint {best}{'' if med[0] <= 1.03 * med[best] else '   <-- auto is not the best'}", flush=True)
    del a, b, out
