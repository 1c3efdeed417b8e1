"""Diagnostic: where a wave of the DH = 64 attention kernel (attention_d64.hip) spends its time per 64-key tile.  Needs the stamps
build ab_tmp/libtribe_hip_attn_stamps.so (built in the container: scripts/build_stamps_lib.sh).  GPU box only; never a timing run."""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ["TRIBE_HIP_LIB"] = str(ROOT / "ab_tmp" / "libtribe_hip_attn_stamps.so")
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402
from tribe_hip import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T, H, D = 8192, 22, 64
dev = torch.device("cuda")
qkv = torch.randn(B * T, 3 * H * D, device=dev).bfloat16()
o = torch.empty(B * T, H * D, device=dev, dtype=torch.bfloat16)
nblocks = (B * H + 7) // 8 * 8 * (T // 256)
dbg = torch.zeros(nblocks * 4 * 8, dtype=torch.int64, device=dev)
d = _lib.AttentionDesc()
inner = H * D
d.q, d.k, d.v = qkv.data_ptr(), qkv.data_ptr() + 2 * inner, qkv.data_ptr() + 4 * inner
d.ld_q = d.ld_k = d.ld_v = 3 * inner
d.out, d.ld_out = o.data_ptr(), inner
d.B, d.T, d.heads_q, d.heads_kv, d.dim_head, d.causal, d.scale = B, T, H, H, D, 0, D**-0.5
d.rel_qe, d.rel_left = dbg.data_ptr(), -1
for _ in range(2):
    dbg.zero_()
    _lib.check(_lib.lib().tribe_attention_fwd_ex(C.byref(d), torch.cuda.current_stream().cuda_stream), "attn")
torch.cuda.synchronize()
t = dbg.view(nblocks, 4, 8).double().cpu()
live = t[:, 0, 5] > 0
names = ["wait + barrier + staging", "S^T MFMAs", "softmax", "P V MFMAs"]
per_tile = t[live][:, :, :4].mean(dim=(0, 1)) / (T / 64)
tot = per_tile.sum()
print(f"B={B}: s_memtime ticks (100 MHz) per 64-key tile and wave: total {tot:.1f} = {tot * 10:.0f} ns")
for n, v in zip(names, per_tile):
    print(f"  {n:26s} {v:9.1f}  {v / tot:6.1%}")
