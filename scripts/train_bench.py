"""Training-step throughput (forward + backward + Adam) of the full-size TRIBE encoder on one MI355X.
GPU box: python scripts/train_bench.py [B] [torch-adam] [graph]      ("graph": forward + backward replayed from one HIP graph)"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from algonauts2025.model import FmriEncoderConfig  # noqa: E402
from algonauts2025.pl_module import BrainModule  # noqa: E402
from data_utils.dataloader import SegmentData  # noqa: E402
from modeling_utils.losses import TorchLossConfig  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T, L, D, V, S = 1024, 2, 2048, 1000, 4
dev = torch.device("cuda")
torch.manual_seed(0)
fdims = {"text": (L, D), "audio": (L, D), "video": (L, D)}
model = FmriEncoderConfig(n_subjects=S).build(fdims, V, T).to(dev).train()
bm = BrainModule(model, TorchLossConfig(name="MSELoss").build(), None, {})
from modeling_utils.optim import HipAdam  # noqa: E402

stock = "torch-adam" in sys.argv[2:]
use_graph = "graph" in sys.argv[2:]
for _a in sys.argv[2:]:
    if _a.startswith("attn-chunk="):   # sequences per chunk of the materialised attention backward (experiment: keep S / P / dP / dS cache-resident)
        from modeling_utils import autograd as _ag
        _ag.Attention.CHUNK_BYTES = int(_a.split("=")[1]) * 8 * 1024 * 1024 * 4
opt = torch.optim.Adam(model.parameters(), lr=1e-4) if stock else HipAdam(model.parameters(), lr=1e-4)   # defaults.py:126-133
g = torch.Generator().manual_seed(1)
data = {m: torch.stack([torch.randn(L, D, T, generator=g).bfloat16() for _ in range(B)]).to(dev) for m in fdims}
data["subject_id"] = (torch.arange(B) % S).view(B, 1).to(dev)
data["fmri"] = torch.randn(B, V, T, generator=g).to(dev)
batch = SegmentData(data=data, segments=[None] * B)


def step():
    opt.zero_grad(set_to_none=True)
    loss = bm.training_step(batch, 0)
    loss.backward()
    opt.step()
    return loss


if use_graph:
    # whole-network capture (torch.cuda.graphs): warm up on a side stream, then record forward + backward once; the optimizer step
    # stays outside (its learning rate and step count are kernel arguments, a recorded launch would freeze them)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        static_loss = bm.training_step(batch, 0)
        static_loss.backward()

    def step():  # noqa: F811
        graph.replay()
        opt.step()
        return static_loss

for _ in range(2):
    step()
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
fl = 3 * 1943.9e6 * B * T  # forward + ~2x backward
print(f"train step B={B} T={T}: {dt * 1e3:.1f} ms  {B * T / dt:.0f} TRs/s  ~{fl / dt / 1e12:.0f} TFLOP/s (3x fwd flops)  loss {float(loss):.4f}  "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB{'  [forward + backward from a HIP graph]' if use_graph else ''}", flush=True)
