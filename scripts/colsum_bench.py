"""Column sums (bias / res_scale gradients) in isolation.  GPU box: python scripts/colsum_bench.py"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from modeling_utils.autograd import colsum  # noqa: E402

for M, N, dtype, with_b in ((16384, 3072, torch.float32, False), (16384, 3072, torch.float32, True), (16384, 12288, torch.bfloat16, False),
                            (16384, 9216, torch.bfloat16, False), (16384, 1000, torch.float32, False)):
    a = torch.randn(M, N, device="cuda").to(dtype)
    b = torch.randn(M, N, device="cuda") if with_b else None
    for _ in range(3):
        colsum(a, M, N, b=b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        colsum(a, M, N, b=b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    nbytes = M * N * (a.element_size() + (4 if with_b else 0))
    print(f"colsum [{M} x {N}] {str(dtype)[6:]:9s}{' x f32 b' if with_b else '        '}: {ms * 1e3:7.1f} us  {nbytes / ms / 1e9:6.2f} TB/s", flush=True)
