"""Training-step throughput (forward + backward + Adam) of the full-size TRIBE encoder on one MI355X.
GPU box: python scripts/train_bench.py [B] [torch-adam] [graph] [--reference-defaults] [no-share] [no-fused-softmax] [no-stream-k] [attn-chunk=N]
  "graph": forward + backward replayed from one HIP graph.
  "--reference-defaults": the configuration the reference actually trains with (grids/defaults.py:95-141, main.py:199,337): batch 16,
      feature widths 2 x 3072 (Llama-3.2-3B) / 2 x 1024 (Wav2Vec-BERT) / 2 x 1408 (V-JEPA2 ViT-g), 298 feature steps pooled to 100 TRs,
      modality_dropout 0.3, contrastive alignment with video (MSE + 0.1 InfoNCE), Adam lr 1e-4 + OneCycleLR(max_lr 1e-4, pct_start 0.1).
  "no-share": always re-run the encoder for the contrastive pass (model.py:228), for the A/B of the shared latents."""
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

REF = "--reference-defaults" in sys.argv[1:]
_pos = [a for a in sys.argv[1:] if a.isdigit()]
B = int(_pos[0]) if _pos else (16 if REF else 4)
T, L, D, V, S = 1024, 2, 2048, 1000, 4
dev = torch.device("cuda")
torch.manual_seed(0)
fdims = {"text": (L, D), "audio": (L, D), "video": (L, D)}
T_OUT = T
cfg_kw = {}
if REF:
    T, T_OUT = 298, 100
    fdims = {"text": (2, 3072), "audio": (2, 1024), "video": (2, 1408)}
    cfg_kw = dict(modality_dropout=0.3, contrastive_enabled=True, contrastive_modalities=["video"], contrastive_weight=0.1,
                  contrastive_temperature=0.07, share_contrastive_latents="no-share" not in sys.argv[1:])
model = FmriEncoderConfig(n_subjects=S, **cfg_kw).build(fdims, V, T_OUT).to(dev).train()
bm = BrainModule(model, TorchLossConfig(name="MSELoss").build(), None, {})
from modeling_utils.optim import HipAdam  # noqa: E402

stock = "torch-adam" in sys.argv[1:]
use_graph = "graph" in sys.argv[1:]
for _a in sys.argv[1:]:
    if _a.startswith("attn-chunk="):   # sequences per chunk of the materialised attention backward (experiment: keep S / P / dP / dS cache-resident)
        from modeling_utils import autograd as _ag
        _ag.Attention.CHUNK_BYTES = _ag.Attention.CHUNK_BYTES_FUSED = int(_a.split("=")[1]) * 8 * 1024 * 1024 * 4
if "no-stream-k" in sys.argv[1:]:   # weight gradients on whole tiles only (the round-2 schedule), for the A/B of tribe_gemm_desc.stream_k
    from modeling_utils import autograd as _ag
    _ag.STREAM_K_WGRAD = False
if "no-fused-softmax" in sys.argv[1:]:   # attention backward through materialised f32 scores + softmax kernels (the round-2 path), for the A/B
    from modeling_utils import autograd as _ag
    _ag.Attention.FUSED_SOFTMAX = False
opt = torch.optim.Adam(model.parameters(), lr=1e-4) if stock else HipAdam(model.parameters(), lr=1e-4)   # defaults.py:126-133
g = torch.Generator().manual_seed(1)
data = {m: torch.stack([torch.randn(l, d, T, generator=g).bfloat16() for _ in range(B)]).to(dev) for m, (l, d) in fdims.items()}
data["subject_id"] = (torch.arange(B) % S).view(B, 1).to(dev)
data["fmri"] = torch.randn(B, V, T_OUT, generator=g).to(dev)
sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-4, pct_start=0.1, total_steps=1000) if REF else None
batch = SegmentData(data=data, segments=[None] * B)


def step():
    opt.zero_grad(set_to_none=True)
    loss = bm.training_step(batch, 0)
    loss.backward()
    opt.step()
    if sched is not None:
        sched.step()
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
n = 20 if REF else 5   # (the default config's step time depends on the dropout draws: average over more steps)
t0 = time.perf_counter()
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
if REF:
    # forward flops per feature step: projectors + contrastive head (video) + encoder (attention term scales with T) + voxel head
    proj = 2 * sum(l * d for l, d in fdims.values()) * 1024 + 2 * 2 * 1408 * 3072
    enc = 8 * (75.50e6 + 150.99e6 + 4 * T * 3072)
    fwd = proj + enc + 6.144e6
    hits = getattr(model, "shared_latent_hits", 0)
    fl = 3 * B * T * (fwd + enc * (1 - hits / (n + 2)))   # the contrastive pass re-runs the encoder unless the draws coincided
    print(f"train step, reference defaults (B={B}, {T} feature steps -> {T_OUT} TRs, widths {dict(fdims)}, modality_dropout 0.3, contrastive video, "
          f"MSE + 0.1 InfoNCE, HipAdam + OneCycleLR): {dt * 1e3:.1f} ms per step  {B * T_OUT / dt:.0f} TRs/s  ~{fl / dt / 1e12:.0f} TFLOP/s "
          f"(3x forward flops incl. the second encoder pass where it ran)  shared-latent steps {hits}/{n + 2}  loss {float(loss):.4f}  "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
else:
    fl = 3 * 1943.9e6 * B * T  # forward + ~2x backward
    print(f"train step B={B} T={T}: {dt * 1e3:.1f} ms  {B * T / dt:.0f} TRs/s  ~{fl / dt / 1e12:.0f} TFLOP/s (3x fwd flops)  loss {float(loss):.4f}  "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB{'  [forward + backward from a HIP graph]' if use_graph else ''}", flush=True)
