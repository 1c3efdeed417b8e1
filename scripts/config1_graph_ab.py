"""BASELINE config 1 on the GPU (text only, one subject, B = 1 x T = 128, 3072 x 8 layers): eager steps against the same step replayed from one HIP
graph (torch.cuda.CUDAGraph around model(batch)) -- ~90 launches per 1.2-ms step, so launch gaps are a visible share here."""
import contextlib
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from algonauts2025.model import FmriEncoderConfig  # noqa: E402
from data_utils.dataloader import SegmentData  # noqa: E402

dev = torch.device("cuda")
fd = {"text": (bench.L, bench.D), "audio": None, "video": None}
torch.manual_seed(0)
with contextlib.redirect_stdout(sys.stderr):
    model = FmriEncoderConfig(n_subjects=1, hidden=bench.HIDDEN, depth=bench.DEPTH, heads=bench.HEADS).build(fd, bench.V, 128).eval().to(dev)
g = torch.Generator(device=dev).manual_seed(3)
data = {"text": torch.randn(1, bench.L, bench.D, 128, generator=g, device=dev).to(torch.bfloat16), "subject_id": torch.zeros(1, 1, dtype=torch.long, device=dev)}
batch = SegmentData(data=data, segments=[None])
steps = 200


def timed(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


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
        a, b = timed(lambda: model(batch)), timed(graph.replay)
        print(f"round {rnd}: eager {a * 1e3:.3f} ms ({128 / a:.0f} TRs/s)   HIP graph {b * 1e3:.3f} ms ({128 / b:.0f} TRs/s)", flush=True)
