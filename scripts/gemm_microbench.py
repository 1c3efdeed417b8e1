"""Micro-benchmark of the MFMA GEMM: fixed per-tile overhead vs per-K-tile cost, per epilogue flavour.
Usage (GPU box): python scripts/gemm_microbench.py"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402
from tribe_hip import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us


def main():
    dev = torch.device("cuda")
    M = 16384
    for N in (3072, 12288):
        for K in (64, 256, 1024, 3072, 12288):
            a = torch.randn(M, K, device=dev).bfloat16()
            b = torch.randn(N, K, device=dev).bfloat16()
            res = torch.randn(M, N, device=dev)
            bias = torch.randn(N, device=dev)
            out32 = torch.empty(M, N, device=dev)
            out16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            t_f32 = timeit(lambda: ops.gemm_nt(a, b, out=out32))
            t_res = timeit(lambda: ops.gemm_nt(a, b, bias=bias, res=res, out=res))
            t_bf = timeit(lambda: ops.gemm_nt(a, b, out=out16))
            t_gelu = timeit(lambda: ops.gemm_nt(a, b, bias=bias, act="gelu", out=out16))
            fl = 2.0 * M * N * K
            print(f"M={M} N={N} K={K:6d}: f32 {t_f32:8.1f}us ({fl/t_f32/1e6:7.1f} TF)  f32+res {t_res:8.1f}us ({fl/t_res/1e6:7.1f})  "
                  f"bf16 {t_bf:8.1f}us ({fl/t_bf/1e6:7.1f})  bf16+gelu {t_gelu:8.1f}us ({fl/t_gelu/1e6:7.1f})", flush=True)
            del a, b, res, out32, out16


if __name__ == "__main__":
    main()
