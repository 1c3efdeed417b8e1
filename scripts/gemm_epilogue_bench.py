"""What the fused epilogue operators cost on the residual-stream GEMMs (out-proj 65536 x 3072 x 3072, FF2 65536 x 3072 x 12288, B = 64 bench
shapes, and their B = 4 forms): the same product with (a) a plain f32 store, (b) + scaled f32 residual in place, (c) + the bf16 copy,
(d) + the row sums of squares (= the model's launch), interleaved in one process.  Usage: python scripts/gemm_epilogue_bench.py"""
import ctypes as C
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from tribe_hip import _lib  # noqa: E402

dev = torch.device("cuda")
lib = _lib.lib()
for name, M, N, K in [("out-proj B=64", 65536, 3072, 3072), ("ff2 B=64", 65536, 3072, 12288), ("out-proj B=4", 4096, 3072, 3072), ("ff2 B=4", 4096, 3072, 12288)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) / K**0.5).bfloat16()
    x = torch.randn(M, N, device=dev)
    xb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    rs, bias = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
    ssq = torch.empty(M, N // 32, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def desc(level):
        d = _lib.GemmDesc()
        d.M, d.N, d.K, d.batch1, d.batch0 = M, N, K, 1, 1
        d.A, d.lda, d.B, d.ldb = a.data_ptr(), K, b.data_ptr(), K
        d.C, d.ldc, d.c_dtype, d.alpha = x.data_ptr(), N, _lib.F32, 1.0
        if level >= 1:
            d.res, d.ldres, d.res_scale = x.data_ptr(), N, rs.data_ptr()
            d.bias, d.bias_mode = bias.data_ptr(), _lib.BIAS_COL
        if level >= 2:
            d.c_bf16, d.ld_c_bf16 = xb.data_ptr(), N
        if level >= 3:
            d.row_sumsq = ssq.data_ptr()
            d.ld_row_sumsq = lib.tribe_gemm_sumsq_slots(C.byref(d))
        return d

    levels = {"plain f32": desc(0), "+ bias + scaled residual": desc(1), "+ bf16 copy": desc(2), "+ row sumsq": desc(3)}
    times = {k: [] for k in levels}
    reps = 5 if M > 10000 else 20
    for rnd in range(5):
        for k, d in levels.items():
            x.normal_()   # the in-place residual would otherwise grow without bound
            lib.tribe_gemm_bf16(C.byref(d), s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                lib.tribe_gemm_bf16(C.byref(d), s)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[k].append(e0.elapsed_time(e1) / reps * 1e3)
    base = None
    for k in levels:
        med = statistics.median(times[k])
        base = base or med
        print(f"{name:14s} {k:28s} {med:9.1f} us  {2.0 * M * N * K / med / 1e6:7.1f} TF  (+{med - base:7.1f} us vs plain)", flush=True)
