"""L2-reuse experiment for the big encoder GEMMs ("FF1 reads 7.25 GB for 0.48 GB of operands").

The tile walk gives the ~32 workgroups an XCD runs at once a GM x (32 / GM) patch of output tiles (GM = band rows, gemm_common.h).
This script times the FF1 / FF2 / QKV shapes of the B = 64 bench step with the library named by TRIBE_HIP_LIB (the container
builds one per GM: ab_tmp/libtribe_gm{2,8,16}.so; the default library has GM = 4).  Run each under
`rocprofv3 --pmc FETCH_SIZE --kernel-trace` to get the fabric read bytes per launch next to the time:

    python scripts/gemm_band_experiment.py                      # prints ms and TFLOP/s per shape
"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from tribe_hip import _lib  # noqa: E402

dev = torch.device("cuda")
shapes = {"ff1": (65536, 12288, 3072, torch.bfloat16, 6), "ff2": (65536, 3072, 12288, torch.float32, 7), "qkv": (65536, 9216, 3072, torch.bfloat16, 2)}
for name, (M, N, K, odt, role) in shapes.items():
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16()
    o = torch.empty(M, N, device=dev, dtype=odt)
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.batch1, d.batch0 = M, N, K, 1, 1
    d.A, d.lda, d.B, d.ldb = a.data_ptr(), K, b.data_ptr(), K
    d.C, d.ldc, d.c_dtype, d.alpha, d.role = o.data_ptr(), N, _lib.BF16 if odt == torch.bfloat16 else _lib.F32, 1.0, role
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.check(_lib.lib().tribe_gemm_bf16(C.byref(d), s), "gemm")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _lib.check(_lib.lib().tribe_gemm_bf16(C.byref(d), s), "gemm")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: M={M} N={N} K={K}  {ms:.4f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")
    del a, b, o
