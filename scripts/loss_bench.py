"""Loss / metric kernels against the HBM roofline (north star: "wavefront reductions for the
per-voxel loss with rocprof-reported HBM GB/s").

GPU box:  python scripts/loss_bench.py [B] [out.json]

Inputs: predictions and targets f32 [B, 1000, 1024] (B = 64: the bench batch, 262 MB per tensor), written just before
the timed launches by a fill kernel so the first read is not served from a warm cache beyond what the pipeline itself
would leave there.  Per kernel: average HIP-event time over 20 launches on the launch stream, ALGORITHMIC bytes (each
input element read once, each output element written once), GB/s and the fraction of the 8 TB/s HBM3E peak
(MI355X_MICROARCH.md; ~6.3 TB/s is what a streaming kernel reaches in practice).  Run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel durations committed under profiles/."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import torch  # noqa: E402

from tribe_hip import _lib, ops  # noqa: E402
from tribe_hip._lib import check, lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
V, T, HBM_PEAK = 1000, 1024, 8.0e12
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
pred = torch.randn(B, V, T, generator=g, device=dev)
true = 0.3 * pred + torch.randn(B, V, T, generator=g, device=dev)
subj = (torch.arange(B, device=dev) % 4).to(torch.int64)
n = pred.numel()
stream = torch.cuda.current_stream().cuda_stream
one = torch.ones((), device=dev)
dpred = torch.empty_like(pred)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


stats1 = torch.zeros(1, V, 6, dtype=torch.float64, device=dev)
stats4 = torch.zeros(4, V, 6, dtype=torch.float64, device=dev)
loss_stats = torch.zeros(V, 6, dtype=torch.float64, device=dev)
ops.pearson_stats_update(loss_stats.view(1, V, 6), pred, true)


def mse_bwd():
    check(lib().tribe_mse_bwd(pred.data_ptr(), true.data_ptr(), n, one.data_ptr(), dpred.data_ptr(), stream), "tribe_mse_bwd")


def pearson_bwd():
    check(lib().tribe_pearson_loss_bwd(pred.data_ptr(), true.data_ptr(), B, V, T, V * T, T, 1, loss_stats.data_ptr(), 0, one.data_ptr(),
                                       dpred.data_ptr(), stream), "tribe_pearson_loss_bwd")


cases = {
    "mse_fwd (mse_partial_kernel)": (lambda: ops.mse(pred, true), 8 * n),
    "mse_bwd (mse_bwd_kernel)": (mse_bwd, 12 * n),
    "pearson_stats_update, one group (pearson_stats_rows_kernel)": (lambda: ops.pearson_stats_update(stats1, pred, true), 8 * n),
    "pearson_stats_update, grouped by subject": (lambda: ops.pearson_stats_update(stats4, pred, true, subj), 8 * n),
    "pearson_loss_fwd (stats + final)": (lambda: ops.pearson_loss(pred, true), 8 * n),
    "pearson_loss_bwd (pearson_loss_bwd_kernel)": (pearson_bwd, 12 * n),
}
out = {"shape": [B, V, T], "dtype": "f32", "hbm_peak_TBps": HBM_PEAK / 1e12, "kernels": {}}
for name, (fn, nbytes) in cases.items():
    s = timed(fn)
    out["kernels"][name] = {"avg_us": round(s * 1e6, 1), "algorithmic_MB": round(nbytes / 1e6, 1), "GBps": round(nbytes / s / 1e9, 1),
                            "frac_of_8TBps": round(nbytes / s / HBM_PEAK, 3)}
    print(f"{name:62s} {s * 1e6:8.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / s / 1e12:5.2f} TB/s  {nbytes / s / HBM_PEAK:5.1%} of peak")
# correctness spot check against torch in f64
x, y = pred.permute(0, 2, 1).reshape(-1, V).double(), true.permute(0, 2, 1).reshape(-1, V).double()
r_ref = ((x - x.mean(0)) * (y - y.mean(0))).sum(0) / ((x - x.mean(0)).norm(dim=0) * (y - y.mean(0)).norm(dim=0))
st = torch.zeros(1, V, 6, dtype=torch.float64, device=dev)
ops.pearson_stats_update(st, pred, true)
out["max_abs_r_error_vs_f64"] = float((ops.pearson_from_stats(st)[0].double() - r_ref).abs().max())
out["mse_rel_error_vs_f64"] = float(abs(ops.mse(pred, true).double() - ((x - y) ** 2).mean()) / ((x - y) ** 2).mean())
print("max |r - r_f64| =", out["max_abs_r_error_vs_f64"], " mse rel err =", out["mse_rel_error_vs_f64"])
if len(sys.argv) > 2:
    Path(sys.argv[2]).write_text(json.dumps(out, indent=1))
