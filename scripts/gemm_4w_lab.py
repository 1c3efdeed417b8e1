"""The one-wave-per-SIMD 256 x 256 GEMM kernel (tile_hint 5) against the 8-wave kernel (tile_hint 2) on the four encoder GEMMs WITH their
model epilogues (QKV: row scale -> bf16; FF1: row scale + bias + GELU -> bf16; out-proj / FF2: [bias +] scaled f32 residual in place + bf16
copy + row sums of squares), B = 64 and B = 4 bench shapes, interleaved in one process on random operands.
Usage: python scripts/gemm_4w_lab.py [name=path/to/lib.so ...]   (default: the in-tree library; stamps builds export tribe_debug_4w)"""
import ctypes as C
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from tribe_hip import _lib  # noqa: E402

libs = {}
for arg in sys.argv[1:] or ["tree=algonauts-2025_amd/tribe_hip/libtribe_hip.so"]:
    name, path = arg.split("=", 1)
    h = C.CDLL(str(ROOT / path))
    h.tribe_gemm_bf16.argtypes = [C.POINTER(_lib.GemmDesc), C.c_void_p]
    h.tribe_gemm_bf16.restype = C.c_int
    h.tribe_gemm_sumsq_slots.argtypes = [C.POINTER(_lib.GemmDesc)]
    libs[name] = h
dev = torch.device("cuda")
SHAPES = [("qkv", 65536, 9216, 3072), ("out_proj", 65536, 3072, 3072), ("ff1", 65536, 12288, 3072), ("ff2", 65536, 3072, 12288),
          ("qkv", 4096, 9216, 3072), ("out_proj", 4096, 3072, 3072), ("ff1", 4096, 12288, 3072), ("ff2", 4096, 3072, 12288)]
for role, M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) / K**0.5).bfloat16()
    bias, rs, scale = torch.randn(N, device=dev), torch.rand(N, device=dev) + 0.5, torch.rand(M, device=dev) + 0.5
    res_role = role in ("out_proj", "ff2")
    x = torch.randn(M, N, device=dev) if res_role else None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)   # bf16 result, or the bf16 copy of x
    ssq = torch.empty(M, N // 32, device=dev) if res_role else None
    s = torch.cuda.current_stream().cuda_stream

    def desc(h, hint):
        d = _lib.GemmDesc()
        d.M, d.N, d.K, d.batch1, d.batch0 = M, N, K, 1, 1
        d.A, d.lda, d.B, d.ldb = a.data_ptr(), K, b.data_ptr(), K
        d.alpha, d.tile_hint, d.role = 1.0, hint, _lib.ROLES.index(role)
        if res_role:
            d.C, d.ldc, d.c_dtype, d.res, d.ldres, d.res_scale = x.data_ptr(), N, _lib.F32, x.data_ptr(), N, rs.data_ptr()
            d.c_bf16, d.ld_c_bf16, d.row_sumsq = out.data_ptr(), N, ssq.data_ptr()
            if role == "ff2":
                d.bias, d.bias_mode = bias.data_ptr(), _lib.BIAS_COL
            d.ld_row_sumsq = h.tribe_gemm_sumsq_slots(C.byref(d))
        else:
            d.C, d.ldc, d.c_dtype, d.row_scale = out.data_ptr(), N, _lib.BF16, scale.data_ptr()
            if role == "ff1":
                d.bias, d.bias_mode, d.act = bias.data_ptr(), _lib.BIAS_COL, _lib.ACT_GELU
        return d

    import os
    arms = ([(n, h, 2) for n, h in libs.items()] if os.environ.get("LAB_HINT2_ONLY") else   # A/B of two builds of the 8-wave kernel
            [(n, h, hint) for n, h in libs.items() for hint in ((2, 5) if n == next(iter(libs)) else (5,))])
    times = {(n, hint): [] for n, _, hint in arms}
    reps = 3 if M > 10000 else 20
    for rnd in range(5):
        for n, h, hint in arms:
            d = desc(h, hint)
            if res_role:
                x.normal_()
            assert h.tribe_gemm_bf16(C.byref(d), s) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                h.tribe_gemm_bf16(C.byref(d), s)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[(n, hint)].append(e0.elapsed_time(e1) / reps)
    print(f"== {role} {M} x {N} x {K}")
    for (n, hint), ts in times.items():
        med = statistics.median(ts)
        line = f"  {n:8s} {'8 waves (hint 2)' if hint == 2 else '4 waves (hint 5)'}: {med * 1e3:9.1f} us  {2.0 * M * N * K / med / 1e9:7.1f} TFLOP/s"
        h = libs[n]
        if hint == 5 and hasattr(h, "tribe_debug_4w"):
            buf = (C.c_ulonglong * 8)()
            h.tribe_debug_4w(buf)
            if buf[1]:
                line += f"   [wg 0: {buf[0] / buf[2]:.0f} cycles per K-tile, clock {buf[0] / (buf[1] * 10.0):.3f} GHz; prologue {buf[3]}, epilogue {buf[4]} cycles]"
        print(line, flush=True)
    del a, b, x, out, ssq
