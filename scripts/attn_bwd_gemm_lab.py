"""The two epilogue GEMMs of the attention backward (P = exp2(scale log2e q.k - lse2), dS = (scale dO.v - scale D) * P), batched over
(sequence, head) = 128 at B = 16, T = 1024, d = 384 -- against the same batched product with a plain bf16 / f32 store and on the other
tile kernels (tile_hint), to see what a 6-K-step tile pays for.  Usage: python scripts/attn_bwd_gemm_lab.py [T]"""
import ctypes as C
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from tribe_hip import _lib  # noqa: E402

dev = torch.device("cuda")
nb, h, T, d = 16, 8, (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 384
Tp = (T + 63) // 64 * 64   # score buffers are padded to 64 columns as in modeling_utils/autograd.py
inner, ld = h * d, 3 * h * d
qkv = torch.randn(nb * T, ld, device=dev).bfloat16()
dout = torch.randn(nb * T, inner, device=dev).bfloat16()
bias = torch.randn(nb, h, T, device=dev)
P = torch.zeros(nb * h, T, Tp, dtype=torch.bfloat16, device=dev)
dS = torch.empty_like(P)
F = torch.empty(nb * h, T, Tp, dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
lib = _lib.lib()


def desc(kind, hint):
    g = _lib.GemmDesc()
    g.M, g.N, g.K, g.batch1, g.batch0, g.alpha, g.tile_hint = T, T, d, nb, h, 0.07, hint
    g.lda, g.sA1, g.sA0 = (inner, T * inner, d) if kind == "ds" else (ld, T * ld, d)
    g.A = dout.data_ptr() if kind == "ds" else qkv.data_ptr()
    g.B, g.ldb, g.sB1, g.sB0 = qkv.data_ptr() + 2 * (2 * inner if kind == "ds" else inner), ld, T * ld, d
    out = F if kind == "f32" else (dS if kind == "ds" else P)
    g.C, g.ldc, g.sC1, g.sC0, g.c_dtype = out.data_ptr(), Tp, h * T * Tp, T * Tp, (_lib.F32 if kind == "f32" else _lib.BF16)
    if kind in ("p", "ds"):
        g.bias, g.bias_mode, g.sBias1, g.sBias0 = bias.data_ptr(), _lib.BIAS_ROW, h * T, T
        g.act = _lib.ACT_EXP2 if kind == "p" else _lib.ACT_MUL_AUX
    if kind == "ds":
        g.aux, g.ld_aux = P.data_ptr(), Tp
    return g


arms = [("plain bf16", "bf16", 0), ("plain f32", "f32", 0), ("P  (EXP2 + row bias)", "p", 0), ("dS (MUL_AUX + row bias)", "ds", 0),
        ("plain bf16, 128x128 double-buffered", "bf16", 1), ("plain bf16, 128x128 ring", "bf16", 3), ("P, 128x128 ring", "p", 3), ("dS, 128x128 ring", "ds", 3),
        ("P, 256x192", "p", 4), ("dS, 128x128 double-buffered", "ds", 1), ("P, 128x128 double-buffered", "p", 1)]
times = {a[0]: [] for a in arms}
for rnd in range(5):
    for name, kind, hint in arms:
        g = desc(kind, hint)
        assert lib.tribe_gemm_bf16(C.byref(g), s) == 0, lib.tribe_last_error()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.tribe_gemm_bf16(C.byref(g), s)
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            times[name].append(e0.elapsed_time(e1) / 5)
fl = 2.0 * T * T * d * nb * h
print(f"batched {nb * h} x [{T} x {T} x {d}] ({fl / 1e9:.0f} GFLOP per launch)")
for name, ts in times.items():
    m = statistics.median(ts)
    print(f"  {name:38s} {m * 1e3:8.1f} us  {fl / m / 1e9:7.1f} TFLOP/s", flush=True)
