"""One V-JEPA2 ViT-g clip (8192 tokens, bf16) five times: the per-kernel breakdown under
rocprofv3 --kernel-trace --stats -- python3 scripts/vjepa2_profile.py [clips]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "scripts"), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from extractor_bench import build_vjepa2, timed  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
enc = build_vjepa2()
clips = torch.randn(B, 64, 3, 256, 256, device="cuda")
dt = timed(lambda: enc.hidden_state_means(clips), n=5, warm=2)
print(f"vjepa2-vitg clips={B}: {dt * 1e3:.2f} ms per forward")
