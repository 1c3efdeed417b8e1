"""Weight-gradient GEMMs of the TRIBE encoder at B = 16 (16384 tokens): the transposed-operand kernel (desc.trans_ab) against two
explicit bf16 transposes + the NT kernel.  GPU box: python scripts/gemm_tn_bench.py"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from modeling_utils import autograd as ag  # noqa: E402
from tribe_hip import ops  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


M = 16384
for name, N, K in (("qkv", 9216, 3072), ("out", 3072, 3072), ("ff1", 12288, 3072), ("ff2", 3072, 12288)):
    dy = torch.randn(M, N, device="cuda").bfloat16()
    x = torch.randn(M, K, device="cuda").bfloat16()
    dw = torch.empty(N, K, device="cuda")

    def old():
        dy_t = ag.transpose_bf16(dy, 1, M, N, 0, N)[0]
        x_t = ag.transpose_bf16(x, 1, M, K, 0, K)[0]
        ag._gemm(dy_t, x_t, dw, lda=M, ldb=M, ldc=K, M=N, N=K, K=M)

    t_old = timed(old)
    ref = dw.clone()
    t_new = timed(lambda: ag._gemm(dy, x, dw, lda=N, ldb=K, ldc=K, M=N, N=K, K=M, trans_ab=True))
    err = float((dw - ref).abs().max() / ref.abs().max())
    fl = 2.0 * M * N * K
    print(f"wgrad {name:4s} [{N} x {K}] over {M} tokens: transposes + NT {t_old:7.3f} ms ({fl / t_old / 1e9:6.0f} TF/s)   TN {t_new:7.3f} ms ({fl / t_new / 1e9:6.0f} TF/s)"
          f"   max rel diff {err:.1e}", flush=True)
