"""Diagnostic: build libtribe_hip_stamps.so with -DTRIBE_GEMM_STAMPS and print where the K loop of the 256x256 GEMM
spends its cycles (shares per slot, per wave group).  GPU box only; never a timing run."""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CS = ROOT / "algonauts-2025_amd" / "csrc"
out = ROOT / "gpurun_out" / "libtribe_hip_stamps.so"
out.parent.mkdir(exist_ok=True)
srcs = ["gemm.hip", "attention.hip", "elementwise.hip", "loss.hip", "encoder.hip", "extractors.hip", "backward.hip", "abi.cpp"]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-DTRIBE_GEMM_STAMPS", "-shared",
                *[str(CS / s) for s in srcs], "-o", str(out)], check=True)
os.environ["TRIBE_HIP_LIB"] = str(out)
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402
from tribe_hip import _lib, ops  # noqa: E402

dev = torch.device("cuda")
for (M, N, K) in [(16384, 3072, 12288), (16384, 12288, 3072)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16()
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ntiles = (M // 256) * (N // 256)
    dbg = torch.zeros(ntiles * 8 * 8, dtype=torch.int64, device=dev)
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.batch1, d.batch0 = M, N, K, 1, 1
    d.A, d.lda, d.B, d.ldb = a.data_ptr(), K, b.data_ptr(), K
    d.C, d.ldc, d.c_dtype, d.alpha = o.data_ptr(), N, _lib.BF16, 1.0
    d.gadd_index = dbg.data_ptr()
    for _ in range(3):
        _lib.check(_lib.lib().tribe_gemm_bf16(C.byref(d), torch.cuda.current_stream().cuda_stream), "gemm")
    torch.cuda.synchronize()
    t = dbg.view(ntiles, 8, 8)[:, :, :5].double().cpu()
    names = ["lds_reads", "stage+vmcnt", "barrier1", "mfma", "barrier2"]
    nk = K // 64
    for grp, sl in (("wr=0", slice(0, 4)), ("wr=1", slice(4, 8))):
        m = t[:, sl].mean(dim=(0, 1))
        tot = m.sum()
        print(f"M={M} N={N} K={K} {grp}: cycles/K-tile {tot / nk:8.1f}  " + "  ".join(f"{n} {v / nk:7.1f} ({v / tot * 100:4.1f}%)" for n, v in zip(names, m)))
