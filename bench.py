#!/usr/bin/env python
"""Headline benchmark: TRs/sec of the trimodal fMRI encode (SegmentData -> [B, 1000, T']) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward pass of the hot path over one batch of synthetic, HBM-resident inputs:
BASELINE.md section 3 -- per rank B = 4 subjects x R stacked passes sequences of T = 1024 feature
steps, three modalities of L x D = 2 x 2048 (= 4096) bf16 features each, 1000 parcels, 4 subjects,
n_output_timesteps = 1024, eval mode, random-init weights from seed 0.  Work per rank is fixed
(weak scaling): ranks own disjoint sequences, and for N > 1 every step's predictions are
all-gathered over RCCL (overlapped with the next step's compute).  Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for _p in (str(ROOT), str(ROOT / "algonauts-2025_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

T, L, D, V, S, HIDDEN, DEPTH, HEADS = 1024, 2, 2048, 1000, 4, 3072, 8, 8
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"


def flops_per_tr(t: int = T) -> dict[str, float]:
    """BASELINE.md section 4 / SURVEY.md section 8(d): algorithmic forward FLOPs per predicted TR row."""
    proj = 3 * 2 * (L * D) * (HIDDEN // 3)
    enc = DEPTH * (2 * 4 * HIDDEN * HIDDEN + 2 * 2 * HIDDEN * 4 * HIDDEN + 4 * t * HIDDEN)
    head = 2 * HIDDEN * V
    return {"projectors": proj, "encoder": enc, "voxel_head": head, "total": proj + enc + head}


def build_model(device: torch.device):
    from algonauts2025.model import FmriEncoderConfig

    torch.manual_seed(0)
    fdims = {"text": (L, D), "audio": (L, D), "video": (L, D)}
    with torch.device(device):   # random-init the 0.94 G parameters on the GPU: eight ranks sharing the host cores would crawl
        model = FmriEncoderConfig(n_subjects=S).build(fdims, n_outputs=V, n_output_timesteps=T).eval()
    return model.to(device), fdims


def make_batch(B: int, fdims, device: torch.device, seed: int):
    from data_utils.dataloader import SegmentData

    g = torch.Generator(device=device).manual_seed(seed)
    data = {}
    for m, (l, d) in fdims.items():
        # synthetic N(0, 1) features, drawn on the GPU per sequence (bounded f32 scratch) and kept resident in bf16
        data[m] = torch.stack([torch.randn(l, d, T, generator=g, device=device).to(torch.bfloat16) for _ in range(B)])
    data["subject_id"] = (torch.arange(B) % S).view(B, 1).to(device)
    return SegmentData(data=data, segments=[None] * B)


def cpu_baseline(threads: int) -> dict:
    """The CPU oracle (fp32 PyTorch restatement of the reference path, oracle/tribe_ref.py) timed on the host
    cores of this box on a bounded sample of the same workload: B = 1 sequence of T = 1024 TRs, full model."""
    from oracle import tribe_ref

    torch.set_num_threads(threads)
    fdims = {"text": (L, D), "audio": (L, D), "video": (L, D)}
    torch.manual_seed(0)
    ref = tribe_ref.FmriEncoderRef(fdims, V, T, S).eval()
    data = tribe_ref.synthetic_batch(1, T, fdims, S, seed=0)
    with torch.inference_mode():
        ref(data)  # warm-up
        times = []
        for _ in range(2):
            t0 = time.perf_counter()
            ref(data)
            times.append(time.perf_counter() - t0)
    med = min(times)
    return {"value": T / med, "unit": "TRs/s", "cores": threads, "kind": "port",
            "sample": f"oracle.tribe_ref.FmriEncoderRef fp32, B=1 x T={T} (1024 TRs), best of 2 passes after 1 warm-up, {med:.2f} s/pass"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=16, help="R: stacked passes of the 4-subject batch per rank (B = 4R); R = 4 / 8 / 16 measured 536 k / 548 k / 554 k TRs/s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL all-gather of predictions (N > 1)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL, default) or gloo (rehearsal "
                    "of the multi-process path when fewer GPUs than ranks are visible)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1 (one process per GPU)")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from algonauts2025.distributed import gather_predictions
    from tribe_hip import ops

    B = 4 * args.repeats
    model, fdims = build_model(device)
    batch = make_batch(B, fdims, device, seed=1000 + rank)  # disjoint sequences per rank
    gather = world > 1 and not args.no_gather
    gather_bufs = [torch.empty(world * B, V, T, dtype=torch.float32, device=device) for _ in range(2)] if gather else None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(i: int, pending: list):
        pred = model(batch)  # [B, V, T] f32
        if gather:
            if len(pending) == 2:
                pending.pop(0).wait()
            pending.append(gather_predictions(pred, out=gather_bufs[i & 1], async_op=True)[1])
        return pred

    pending: list = []
    for i in range(args.warmup):
        step(i, pending)
    for w in pending:
        w.wait()
    pending.clear()
    barrier()

    ops.prof_begin(max_records=(args.steps + 1) * 128)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    start.record()
    for i in range(args.steps):
        step(i, pending)
    for w in pending:
        w.wait()
    stop.record()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ops.prof_end()

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    trs = world * B * T * args.steps  # (subject, TR) rows predicted by all ranks in the timed region

    if rank == 0:
        fl = flops_per_tr()
        # dominant kernel = the GEMM operator with the largest summed time inside the timed region
        by_kernel = {}
        for role, r in prof.items():
            avg_ms = r["ms"] / r["launches"]
            tf = r["flops"] / r["launches"] / (avg_ms * 1e-3) / 1e12
            by_kernel[role] = {"avg_ms": round(avg_ms, 4), "launches": r["launches"],
                               "gflop_per_launch": round(r["flops"] / r["launches"] / 1e9, 3),
                               "achieved_tflops": round(tf, 1), "frac": round(tf / PEAK_BF16_TFLOPS, 4)}
        dom = max(prof, key=lambda k: prof[k]["ms"])
        gemm_ms = sum(r["ms"] for r in prof.values()) / args.steps
        roofline = {"bound": "mfma", "kernel": f"gemm_nt_256x256x64<{dom}>", "achieved": by_kernel[dom]["achieved_tflops"],
                    "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": by_kernel[dom]["frac"], "traffic": None,
                    "avg_launch_ms": by_kernel[dom]["avg_ms"], "flops_per_launch": by_kernel[dom]["gflop_per_launch"] * 1e9,
                    "by_kernel": by_kernel,
                    "whole_path": {"tflops": round(trs / world * fl["total"] / elapsed / 1e12, 1),
                                   "frac": round(trs / world * fl["total"] / elapsed / 1e12 / PEAK_BF16_TFLOPS, 4),
                                   "gemm_ms_per_step": round(gemm_ms, 3)}}
        traffic_file = ROOT / "profiles" / "roofline_traffic.json"
        if traffic_file.exists():
            try:
                measured = json.loads(traffic_file.read_text())
                if measured.get("_sequences_per_gpu") == B:   # PMC bytes per launch are only comparable at the batch they were taken at
                    roofline["traffic"] = measured.get(dom)
            except Exception:
                pass
        out = {
            "metric": "TRs/sec trimodal encode to 1000 parcels", "value": round(trs / elapsed, 1), "unit": "TRs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"trimodal encode, precomputed bf16 feats, per GPU B={B} sequences (4 subjects x R={args.repeats}) "
                                   f"x T={T}, L*D={L * D} per modality, hidden {HIDDEN}, {DEPTH} layers, V={V} parcels, "
                                   f"n_output_timesteps={T}; forward only" + ("; RCCL all-gather of predictions" if gather else ""),
                       "global_batch": world * B, "seq_len": T, "parallelism": f"dp{world}", "flops_per_tr": fl["total"]},
            "gpu_event_ms_per_step": round(start.elapsed_time(stop) / args.steps, 3),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            del model, batch
            torch.cuda.empty_cache()
            # a 1-GPU box grants 16 CPUs; more threads than that only oversubscribes the share
            out["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, 16))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
