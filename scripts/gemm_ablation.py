"""What bounds the K loop of the 256 x 256 x 64 GEMM?  Times the ablation builds of scripts/build_gemm_ablation.sh (each a GEMM-only
library under ab_tmp/) interleaved in ONE process on random operands (guide rules 24 / 25): the shipped loop, the loop without
fragment ds_reads, without LDS-DMA, without both, and without MFMAs.  Run times of ablated builds are diagnostic only.
Usage (GPU box): python scripts/gemm_ablation.py > gpurun_out/gemm_ablation.txt"""
import ctypes as C
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from tribe_hip import _lib  # noqa: E402

VARIANTS = ["base", "nolds", "nostage", "bare", "nomfma", "dmaonly", "same", "dmasame"]
libs = {}
for v in VARIANTS:
    p = ROOT / "ab_tmp" / f"libgemm_abl_{v}.so"
    if p.exists():
        h = C.CDLL(str(p))
        h.tribe_gemm_bf16.argtypes = [C.POINTER(_lib.GemmDesc), C.c_void_p]
        h.tribe_gemm_bf16.restype = C.c_int
        libs[v] = h
dev = torch.device("cuda")
shapes = {"8192^3": (8192, 8192, 8192), "ff1 65536x12288x3072": (65536, 12288, 3072), "ff2 65536x3072x12288": (65536, 3072, 12288),
          "qkv 65536x9216x3072": (65536, 9216, 3072), "ff1 B=4 4096x12288x3072": (4096, 12288, 3072)}
for name, (M, N, K) in shapes.items():
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16()
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.batch1, d.batch0 = M, N, K, 1, 1
    d.A, d.lda, d.B, d.ldb = a.data_ptr(), K, b.data_ptr(), K
    d.C, d.ldc, d.c_dtype, d.alpha, d.role = o.data_ptr(), N, _lib.BF16, 1.0, 0
    s = torch.cuda.current_stream().cuda_stream
    times = {v: [] for v in libs}
    reps = 6 if M * N * K < 1e12 else 3
    for rnd in range(6):
        for v, h in libs.items():
            for _ in range(1 if rnd else 2):
                assert h.tribe_gemm_bf16(C.byref(d), s) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                h.tribe_gemm_bf16(C.byref(d), s)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[v].append(e0.elapsed_time(e1) / reps)
    tiles_per_cu = (M // 256) * (N // 256) / 256.0
    nk = K // 64
    print(f"== {name}: {tiles_per_cu:.1f} tiles per CU x {nk} K-steps")
    for v in libs:
        med, mn = statistics.median(times[v]), min(times[v])
        print(f"  {v:8s} median {med:8.4f} ms  min {mn:8.4f} ms  {2.0 * M * N * K / med / 1e9:7.1f} TFLOP/s-equivalent  "
              f"{med * 1e3 / (tiles_per_cu * nk):6.3f} us per K-step", flush=True)
    del a, b, o
