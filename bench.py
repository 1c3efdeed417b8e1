#!/usr/bin/env python
"""Headline benchmark: TRs/sec of the trimodal fMRI encode (SegmentData -> [B, 1000, T']) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats R]

`python bench.py --gpus N` invoked plainly starts N fresh rank processes itself (one per GPU, the layout of the
reference's only multi-GPU path, main.py:388-395: one task per GPU): the parent touches no GPU API, picks a free
rendezvous port on 127.0.0.1, spawns the children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's
JSON line and exits with the first non-zero child code.  Under an external launcher
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) the process IS a rank and runs directly.

A "step" is one forward pass of the hot path over one batch of synthetic, HBM-resident inputs:
BASELINE.md section 3 -- per rank B = 4 subjects x R stacked passes sequences of T = 1024 feature
steps, three modalities of L x D = 2 x 2048 (= 4096) bf16 features each, 1000 parcels, 4 subjects,
n_output_timesteps = 1024, eval mode, random-init weights from seed 0.  Work per rank is fixed
(weak scaling): ranks own disjoint sequences, and for N > 1 every step's predictions are exchanged
(`--exchange gather`: bf16 all-gather of the [B, V, T'] shards, SURVEY 8e's sizing; `--exchange stats`: all-reduce of the
f64 per-voxel Pearson statistics only; `none`), overlapped with the next step's compute.  Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for _p in (str(ROOT), str(ROOT / "algonauts-2025_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

T, L, D, V, S, HIDDEN, DEPTH, HEADS = 1024, 2, 2048, 1000, 4, 3072, 8, 8
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
METRIC = "TRs/sec trimodal encode to 1000 parcels"


def parse_args(argv=None) -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=16, help="R: stacked passes of the 4-subject batch per rank (B = 4R); R = 4 / 8 / 16 measured 536 k / 548 k / 554 k TRs/s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-r1-point", action="store_true", help="skip the untimed-headline R = 1 (B = 4) measurement")
    ap.add_argument("--exchange", choices=("gather", "stats", "none"), default="gather",
                    help="N > 1: what the ranks exchange per step (bf16 prediction all-gather / f64 Pearson statistics all-reduce / nothing)")
    ap.add_argument("--no-gather", action="store_true", help="same as --exchange none")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL, default) or gloo (rehearsal "
                    "of the multi-process path when fewer GPUs than ranks are visible)")
    ap.add_argument("--launcher-rehearsal", action="store_true",
                    help="exercise ONLY the process launch / rendezvous / exchange / reporting plumbing with a stand-in "
                         "step on host tensors (CPU test of the N > 1 path; no kernel runs and the line carries value null)")
    args = ap.parse_args(argv)
    if args.no_gather:
        args.exchange = "none"
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    return args


# ---------------------------------------------------------------------------------------------------------------------
# parent: one fresh process per GPU (never an exec of, or a fork from, a process that touched the GPU)
# ---------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv: list[str]) -> int:
    """Start n children of this same script, wait for all, relay their output, return the first non-zero exit code.
    No torch import and no HIP call happens in this process."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TRIBE_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + argv, env=env))
    rc = 0
    deadline = None   # set at the first failure: survivors get SIGTERM, then 30 s, then SIGKILL (a rank stuck in a collective ignores SIGTERM)
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    deadline = time.monotonic() + 30.0
                    print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    for q in pending:
                        procs[q].terminate()
            if deadline is not None and pending and time.monotonic() > deadline:
                print(f"bench.py: ranks {sorted(pending)} ignored SIGTERM for 30 s; killing them", file=sys.stderr, flush=True)
                break   # the finally clause kills what is left; the recorded non-zero code is returned
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
    return rc


# ---------------------------------------------------------------------------------------------------------------------
# rank process
# ---------------------------------------------------------------------------------------------------------------------
def flops_per_tr(t: int = T) -> dict[str, float]:
    """BASELINE.md section 4 / SURVEY.md section 8(d): algorithmic forward FLOPs per predicted TR row."""
    proj = 3 * 2 * (L * D) * (HIDDEN // 3)
    enc = DEPTH * (2 * 4 * HIDDEN * HIDDEN + 2 * 2 * HIDDEN * 4 * HIDDEN + 4 * t * HIDDEN)
    head = 2 * HIDDEN * V
    return {"projectors": proj, "encoder": enc, "voxel_head": head, "total": proj + enc + head}


def build_model(device):
    import torch

    from algonauts2025.model import FmriEncoderConfig

    torch.manual_seed(0)
    fdims = {"text": (L, D), "audio": (L, D), "video": (L, D)}
    with torch.device(device):   # random-init the 0.94 G parameters on the GPU: eight ranks sharing the host cores would crawl
        model = FmriEncoderConfig(n_subjects=S).build(fdims, n_outputs=V, n_output_timesteps=T).eval()
    return model.to(device), fdims


def make_batch(B: int, fdims, device, seed: int):
    import torch

    from data_utils.dataloader import SegmentData

    g = torch.Generator(device=device).manual_seed(seed)
    data = {}
    for m, (l, d) in fdims.items():
        # synthetic N(0, 1) features, drawn on the GPU per sequence (bounded f32 scratch) and kept resident in bf16
        data[m] = torch.stack([torch.randn(l, d, T, generator=g, device=device).to(torch.bfloat16) for _ in range(B)])
    data["subject_id"] = (torch.arange(B) % S).view(B, 1).to(device)
    return SegmentData(data=data, segments=[None] * B)


def _time_oracle(ref, data, passes: int) -> float:
    import torch

    with torch.inference_mode():
        ref(data)  # warm-up
        times = []
        for _ in range(passes):
            t0 = time.perf_counter()
            ref(data)
            times.append(time.perf_counter() - t0)
    return min(times)


def cpu_baseline(threads: int) -> dict:
    """The CPU oracle (fp32 PyTorch restatement of the reference path, oracle/tribe_ref.py) timed on the host cores of
    this box on bounded samples of the workload (SURVEY 8d / BASELINE.md 5).  Headline `value`: B = 1 sequence of
    T = 1024 TRs through the full model on all granted cores; `legs` adds the full B = 4 pass, BASELINE config 1
    (text only, B = 1, T = 128: the reference's own CPU-runnable case) and the single-thread rates."""
    import torch

    from oracle import tribe_ref

    fdims = {"text": (L, D), "audio": (L, D), "video": (L, D)}
    torch.manual_seed(0)
    ref = tribe_ref.FmriEncoderRef(fdims, V, T, S).eval()
    one = tribe_ref.synthetic_batch(1, T, fdims, S, seed=0)
    four = tribe_ref.synthetic_batch(4, T, fdims, S, seed=0)

    legs = {}
    torch.set_num_threads(threads)
    t_one = _time_oracle(ref, one, 2)
    legs[f"synthetic_B1_T{T}_{threads}thr"] = {"TRs_per_s": round(T / t_one, 1), "s_per_pass": round(t_one, 3)}
    t_four = _time_oracle(ref, four, 1)
    legs[f"synthetic_B4_T{T}_{threads}thr"] = {"TRs_per_s": round(4 * T / t_four, 1), "s_per_pass": round(t_four, 3)}
    torch.set_num_threads(1)
    quarter = tribe_ref.synthetic_batch(1, 256, fdims, S, seed=0)   # a quarter-length sequence keeps the 1-thread leg to seconds
    t_q = _time_oracle(ref, quarter, 1)
    legs["synthetic_B1_T256_1thr"] = {"TRs_per_s": round(256 / t_q, 1), "s_per_pass": round(t_q, 3)}
    del ref, four
    # BASELINE config 1: text-only features (audio / video absent -> zero thirds, model.py:143-144), 1 subject, 128 TRs
    ref1 = tribe_ref.FmriEncoderRef({"text": (L, D), "audio": None, "video": None}, V, 128, 1).eval()
    c1 = tribe_ref.synthetic_batch(1, 128, {"text": (L, D)}, 1, seed=0)
    t_c1_1 = _time_oracle(ref1, c1, 2)
    legs["config1_text_only_B1_T128_1thr"] = {"TRs_per_s": round(128 / t_c1_1, 1), "s_per_pass": round(t_c1_1, 4)}
    torch.set_num_threads(threads)
    t_c1 = _time_oracle(ref1, c1, 3)
    legs[f"config1_text_only_B1_T128_{threads}thr"] = {"TRs_per_s": round(128 / t_c1, 1), "s_per_pass": round(t_c1, 4)}
    return {"value": T / t_one, "unit": "TRs/s", "cores": threads, "kind": "port",
            "sample": f"oracle.tribe_ref.FmriEncoderRef fp32, B=1 x T={T} (1024 TRs), best of 2 passes after 1 warm-up, {t_one:.2f} s/pass",
            "legs": legs}


def _role_table(prof: dict) -> dict:
    table = {}
    for role, r in prof.items():
        avg_ms = r["ms"] / r["launches"]
        unit_per_s = r["flops"] / r["launches"] / (avg_ms * 1e-3) / 1e12
        table[role] = {"avg_ms": round(avg_ms, 4), "launches": r["launches"],
                       "gflop_per_launch": round(r["flops"] / r["launches"] / 1e9, 3),
                       "achieved_tflops": round(unit_per_s, 1), "frac": round(unit_per_s / PEAK_BF16_TFLOPS, 4)}
    return table


def rehearse_launcher(args, rank: int, world: int) -> None:
    """Stand-in rank body for the CPU test of the N > 1 plumbing: rendezvous, the bf16 exchange, barrier / MAX-over-ranks
    timing and the one JSON line, with host tensors filled with the rank id instead of predictions.  Nothing here is the
    product path and nothing is measured: `value` is null."""
    import torch
    import torch.distributed as dist

    from algonauts2025.distributed import gather_predictions

    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    b, v, t = 2, 8, 16
    pred = torch.full((b, v, t), float(rank), dtype=torch.float32)
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        out, _ = gather_predictions(pred.to(torch.bfloat16))
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    ok = all(float(out[r * b:(r + 1) * b].float().mean()) == float(r) for r in range(world))
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "TRs/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "bf16", "data": "launcher rehearsal: no kernel ran, nothing measured",
                          "config": {"workload": f"launcher rehearsal, {world} ranks over gloo", "parallelism": f"dp{world}"},
                          "exchange_ok": ok}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def run_rank(args) -> None:
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.launcher_rehearsal:
        return rehearse_launcher(args, rank, world)

    import torch
    import torch.distributed as dist

    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and world > n_dev:
        raise SystemExit(f"--gpus {world} over RCCL needs {world} visible GPUs, found {n_dev} (use --backend gloo to rehearse)")
    device = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(device)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from algonauts2025.distributed import allreduce_stats, gather_predictions
    from tribe_hip import ops

    B = 4 * args.repeats
    model, fdims = build_model(device)
    batch = make_batch(B, fdims, device, seed=1000 + rank)  # disjoint sequences per rank
    exchange = args.exchange if world > 1 else "none"
    gather_bufs = pred_bf16 = stats = fmri = None
    if exchange == "gather":
        gather_bufs = [torch.empty(world * B, V, T, dtype=torch.bfloat16, device=device) for _ in range(2)]
        pred_bf16 = [torch.empty(B, V, T, dtype=torch.bfloat16, device=device) for _ in range(2)]
    elif exchange == "stats":
        fmri = torch.randn(B, V, T, device=device)
        stats = [torch.zeros(1, V, 6, dtype=torch.float64, device=device) for _ in range(2)]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(i: int, pending: list):
        pred = model(batch)  # [B, V, T] f32
        if exchange == "none":
            return pred
        if len(pending) == 2:
            pending.pop(0).wait()
        if exchange == "gather":
            pred_bf16[i & 1].copy_(pred)       # SURVEY 8e sizes the exchange in bf16: 131 MB per rank at B = 64
            pending.append(gather_predictions(pred_bf16[i & 1], out=gather_bufs[i & 1], async_op=True)[1])
        else:
            st = stats[i & 1].zero_()
            ops.pearson_stats_update(st, pred, fmri)
            if args.backend == "gloo":          # gloo reduces host tensors
                host = st.cpu()
                dist.all_reduce(host)
                st.copy_(host)
            else:
                pending.append(dist.all_reduce(st, async_op=True))
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
        if args.backend == "gloo":
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX)
            t = host
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    trs = world * B * T * args.steps  # (subject, TR) rows predicted by all ranks in the timed region

    if rank == 0:
        fl = flops_per_tr()
        # dominant kernel = the operator (GEMM role or the fused attention) with the largest summed time inside the timed region
        by_kernel = _role_table(prof)
        dom = max(prof, key=lambda k: prof[k]["ms"])
        gemm_ms = sum(r["ms"] for r in prof.values()) / args.steps
        # kernel symbol behind the dominant role at this batch: every GEMM role runs the 8-wave 256 x 256 kernel by default (csrc/gemm.hip:
        # gemm_plan; the one-wave-per-SIMD gemm_nt_4w256 is tile_hint 5) -- rocprof shows it as gemm_nt_256x256x64<out_bf16, role, 0, 4>
        dom_name = "attn_fwd_wide384_kernel" if dom == "attention" else f"gemm_nt_256x256x64<{dom}>"
        roofline = {"bound": "mfma", "kernel": dom_name, "achieved": by_kernel[dom]["achieved_tflops"],
                    "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": by_kernel[dom]["frac"], "traffic": None,
                    "traffic_source": None,
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
                    roofline["traffic_source"] = ("stored: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, gfx950-corrected, "
                                                  f"committed as profiles/roofline_traffic.json ({measured.get('_source', 'see profiles/')}); "
                                                  "NOT re-measured in this run")
            except Exception:
                pass
        xch = {"gather": f"; bf16 all-gather of predictions over {'RCCL' if args.backend == 'nccl' else args.backend}",
               "stats": f"; f64 Pearson-statistics all-reduce over {'RCCL' if args.backend == 'nccl' else args.backend}",
               "none": ""}[exchange]
        out = {
            "metric": METRIC, "value": round(trs / elapsed, 1), "unit": "TRs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"trimodal encode, precomputed bf16 feats, per GPU B={B} sequences (4 subjects x R={args.repeats}) "
                                   f"x T={T}, L*D={L * D} per modality, hidden {HIDDEN}, {DEPTH} layers, V={V} parcels, "
                                   f"n_output_timesteps={T}; forward only" + xch,
                       "global_batch": world * B, "seq_len": T, "parallelism": f"dp{world}", "backend": args.backend if world > 1 else None,
                       "flops_per_tr": fl["total"]},
            "gpu_event_ms_per_step": round(start.elapsed_time(stop) / args.steps, 3),
            "roofline": roofline,
        }
        if world == 1 and not args.no_r1_point:
            out["r1_point"] = r1_point(model, fdims, device, ops)
            out["config1_point"] = config1_point(device, ops)
        if world == 1 and not args.no_cpu_baseline:
            del model, batch
            torch.cuda.empty_cache()
            # a 1-GPU box grants 16 CPUs; more threads than that only oversubscribes the share
            out["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, 16))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _side_point(model, batch, ops, steps: int, warmup: int):
    """(seconds per step WITHOUT the per-launch HIP events, role profile of a second, instrumented pass).  The event pair around every
    GEMM / attention launch costs a step of a few milliseconds about 4 % (B = 4: 7.26 ms with, 6.98 ms without, profiles/r03_z8_r1_graph.txt;
    replaying the step from a HIP graph gains nothing further), so the side points time clean steps and profile separately."""
    import torch

    with torch.no_grad():
        for _ in range(warmup):
            model(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model(batch)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        ops.prof_begin(max_records=(steps + 1) * 128)
        for _ in range(steps):
            model(batch)
        torch.cuda.synchronize()
        prof = ops.prof_end()
    return elapsed / steps, prof


def r1_point(model, fdims, device, ops, steps: int = 30, warmup: int = 5) -> dict:
    """The BASELINE configuration un-stacked: B = 4 (one 1024-TR sequence per subject), same model, outside the timed
    headline.  Reported so the per-operator numbers (voxel head above all) exist at the literal config, where a launch has
    16x fewer tiles to fill 256 CUs with."""
    import torch

    batch = make_batch(4, fdims, device, seed=7)
    per_step, prof = _side_point(model, batch, ops, steps, warmup)
    fl = flops_per_tr()
    return {"workload": f"B=4 sequences (4 subjects x R=1) x T={T}", "value": round(4 * T / per_step, 1), "unit": "TRs/s",
            "steps": steps, "ms_per_step": round(per_step * 1e3, 3),
            "whole_path_tflops": round(4 * T * fl["total"] / per_step / 1e12, 1), "by_kernel": _role_table(prof),
            "by_kernel_source": "a second pass of the same steps with the per-launch HIP events on (they cost ~4 % at this step size; value is the clean pass)"}


def config1_point(device, ops, steps: int = 50, warmup: int = 10) -> dict:
    """BASELINE config 1 on the GPU, outside the timed headline: text-only features (audio = video = None -> zero thirds of the
    fusion, /root/reference/algonauts2025/model.py:143-144), one subject, B = 1 sequence of T = 128 TRs, the full-size encoder
    (hidden 3072 x 8 layers, V = 1000) -- the GPU figure beside `cpu_baseline.legs.config1_text_only_B1_T128_*`.  128 rows give
    every GEMM a single row of tiles: a launch-latency-bound point by construction, reported as measured."""
    import torch

    from algonauts2025.model import FmriEncoderConfig
    from data_utils.dataloader import SegmentData

    fd = {"text": (L, D), "audio": None, "video": None}
    torch.manual_seed(0)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):   # the constructor prints the reference's "no feature dimensions" warnings; stdout carries ONE JSON line
        model = FmriEncoderConfig(n_subjects=1, hidden=HIDDEN, depth=DEPTH, heads=HEADS).build(fd, V, 128).eval().to(device)
    g = torch.Generator(device=device).manual_seed(3)
    data = {"text": torch.randn(1, L, D, 128, generator=g, device=device).to(torch.bfloat16),
            "subject_id": torch.zeros(1, 1, dtype=torch.long, device=device)}
    batch = SegmentData(data=data, segments=[None])
    per_step, prof = _side_point(model, batch, ops, steps, warmup)
    del model
    return {"workload": "BASELINE config 1: text-only, 1 subject, B=1 x T=128, hidden 3072 x 8 layers, V=1000 (audio / video absent)",
            "value": round(128 / per_step, 1), "unit": "TRs/s", "steps": steps, "ms_per_step": round(per_step * 1e3, 3),
            "by_kernel": _role_table(prof), "by_kernel_source": "second, instrumented pass (see r1_point)"}


def main() -> None:
    args = parse_args()
    external = "RANK" in os.environ and "WORLD_SIZE" in os.environ    # torch.distributed.run, or our own parent
    if args.gpus > 1 and not external:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    run_rank(args)


if __name__ == "__main__":
    main()
