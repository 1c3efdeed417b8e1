"""Usage: python scripts/r1_graph_ab.py [B] [steps].  The literal BASELINE config (B = 4 x T = 1024; or B stacked sequences) three ways, same process: eager with the per-launch HIP-event profile on (what r1_point
timed until round 3), eager without it, and replayed from one HIP graph (torch.cuda.CUDAGraph around model(batch))."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from tribe_hip import ops  # noqa: E402

dev = torch.device("cuda")
model, fdims = bench.build_model(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = bench.make_batch(B, fdims, dev, seed=7)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50


def timed(fn, prof=False):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    if prof:
        ops.prof_begin(max_records=(steps + 1) * 128)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if prof:
        ops.prof_end()
    return dt


with torch.no_grad():
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            model(batch)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_out = model(batch)
    ref = model(batch)
    graph.replay()
    torch.cuda.synchronize()
    print("graph output == eager output:", bool(torch.equal(ref, static_out)))
    for rnd in range(3):
        a = timed(lambda: model(batch), prof=True)
        b = timed(lambda: model(batch))
        c = timed(graph.replay)
        print(f"round {rnd}: eager + per-launch events {a * 1e3:.3f} ms ({B * 1024 / a:.0f} TRs/s)   eager {b * 1e3:.3f} ms ({B * 1024 / b:.0f} TRs/s)   "
              f"HIP graph {c * 1e3:.3f} ms ({B * 1024 / c:.0f} TRs/s)", flush=True)
