"""RCCL sanity on one GPU: init_process_group('nccl', world_size=1, device_id=...), the collectives bench.py uses."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from algonauts2025.distributed import allreduce_stats, gather_predictions

x = torch.randn(4, 10, 16, device=dev)
buf = torch.empty(4, 10, 16, device=dev)
w = dist.all_gather_into_tensor(buf, x, async_op=True)
w.wait()
assert torch.equal(buf, x)
t = torch.ones(3, dtype=torch.float64, device=dev)
dist.all_reduce(t)
dist.barrier()
torch.cuda.synchronize()
print("nccl ok", dist.get_backend(), gather_predictions(x)[0].shape, allreduce_stats(t).tolist())
dist.destroy_process_group()
