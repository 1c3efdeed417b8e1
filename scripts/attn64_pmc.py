"""Counter run for the DH = 64 attention kernels (V-JEPA2 shape, 2 clips): 3 launches each of modes 2, 4, 5.
rocprofv3 --pmc <counters> --kernel-trace -- python3 scripts/attn64_pmc.py"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from tribe_hip import ops  # noqa: E402

B, T, H, D = 2, 8192, 22, 64
qkv = torch.randn(B * T, 3 * H * D, device="cuda").bfloat16()
for mode in (2, 4, 5):
    ops.attention_set_mode(mode)
    for _ in range(3):
        ops.attention(qkv, B, T, H, D, D**-0.5)
    torch.cuda.synchronize()
ops.attention_set_mode(0)
