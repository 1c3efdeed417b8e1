"""Extractor GEMM shapes, 256^2 tiles (hint 2) against 128^2 tiles (hint 1): where tile quantisation (tiles vs 256 CUs) decides.
GPU box: python scripts/gemm_shapes_bench.py"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from tribe_hip import ops  # noqa: E402

shapes = [("vjepa2 qkv", 8192, 4224, 1408), ("vjepa2 proj", 8192, 1408, 1408), ("vjepa2 fc1", 8192, 6144, 1408), ("vjepa2 fc2", 8192, 1408, 6144),
          ("vjepa2 qkv x2", 16384, 4224, 1408), ("vjepa2 proj x2", 16384, 1408, 1408), ("vjepa2 fc2 x2", 16384, 1408, 6144),
          ("w2v qkv x8", 24000, 3072, 1024), ("w2v ffn1 x8", 24000, 4096, 1024), ("w2v ffn2 x8", 24000, 1024, 4096), ("w2v out x8", 24000, 1024, 1024),
          ("w2v qkv x1", 3000, 3072, 1024), ("w2v ffn1 x1", 3000, 4096, 1024), ("w2v ffn2 x1", 3000, 1024, 4096),
          ("tribe B=4 qkv", 4096, 9216, 3072), ("tribe B=4 out", 4096, 3072, 3072), ("tribe B=4 ff2", 4096, 3072, 12288),
          ("tribe B=4 projector", 4096, 1024, 4096)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16()
    b = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    row = []
    for hint in (2, 1):
        for _ in range(3):
            ops.gemm_nt(a, b, out=out, out_dtype=torch.bfloat16, tile_hint=hint)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm_nt(a, b, out=out, out_dtype=torch.bfloat16, tile_hint=hint)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        row.append((ms, 2.0 * M * N * K / ms / 1e9))
    t256 = -(-M // 256) * -(-N // 256)
    t128 = -(-M // 128) * -(-N // 128)
    print(f"{name:16s} M={M:6d} N={N:5d} K={K:5d}  256^2: {t256:5d} tiles {row[0][0] * 1e3:7.1f} us {row[0][1]:7.1f} TF/s   128^2: {t128:5d} tiles "
          f"{row[1][0] * 1e3:7.1f} us {row[1][1]:7.1f} TF/s", flush=True)
