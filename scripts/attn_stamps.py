"""Diagnostic: build a copy of the library with -DTRIBE_ATTN_STAMPS and print where a wave of the DH = 384 attention kernel spends
its cycles per 32-key tile (S^T phase, softmax, wait for the LDS-DMA of tile t + 1, barrier, P V phase).  GPU box only; never a
timing run (every stamp drains the LDS queue)."""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CS = ROOT / "algonauts-2025_amd" / "csrc"
out = ROOT / "ab_tmp" / "libtribe_hip_attn_stamps.so"      # built in the container (hipcc), travels with the snapshot
if not out.exists():
    out.parent.mkdir(exist_ok=True)
    obj = out.with_name("attention_stamps.o")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-DTRIBE_ATTN_STAMPS", f"-I{CS}", "-c",
                    str(CS / "attention.hip"), "-o", str(obj)], check=True)
    objs = [str(CS / f"{n}.o") for n in ("gemm", "elementwise", "loss", "encoder", "extractors", "backward", "features", "gemm_fp8", "abi")]
    subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", str(obj), *objs, "-o", str(out)], check=True)
os.environ["TRIBE_HIP_LIB"] = str(out)
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402
from tribe_hip import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T, H, D = 1024, 8, 384
dev = torch.device("cuda")
qkv = torch.randn(B * T, 3 * H * D, device=dev).bfloat16()
o = torch.empty(B * T, H * D, device=dev, dtype=torch.bfloat16)
nblocks = (B * H + 7) // 8 * 8 * (T // 128)
dbg = torch.zeros(nblocks * 4 * 8, dtype=torch.int64, device=dev)
d = _lib.AttentionDesc()
inner = H * D
d.q, d.k, d.v = qkv.data_ptr(), qkv.data_ptr() + 2 * inner, qkv.data_ptr() + 4 * inner
d.ld_q = d.ld_k = d.ld_v = 3 * inner
d.out, d.ld_out = o.data_ptr(), inner
d.B, d.T, d.heads_q, d.heads_kv, d.dim_head, d.causal, d.scale = B, T, H, H, D, 0, D**-0.5
d.rel_qe = dbg.data_ptr()
for _ in range(2):
    _lib.check(_lib.lib().tribe_attention_fwd_ex(C.byref(d), torch.cuda.current_stream().cuda_stream), "attn")
torch.cuda.synchronize()
t = dbg.view(nblocks, 4, 8).double().cpu()
names = ["S^T phase", "softmax", "vmcnt wait", "barrier", "P V phase"]
per_tile = t[:, :, :5].mean(dim=(0, 1)) / T * 32
tot = per_tile.sum()
print(f"B={B}: s_memtime ticks per 32-key tile (100 MHz constant clock x ? -- read the shares): total {tot:.1f}")
for n, v in zip(names, per_tile):
    print(f"  {n:12s} {v:9.1f}  {v / tot:6.1%}")
