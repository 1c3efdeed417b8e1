#!/bin/bash
# Collects the evidence the bench line is judged against, on the GPU box:  bash scripts/profile_round.sh <tag>   (e.g. r02_f)
#   <tag>_bench.json                  the default bench.py line (roofline + cpu_baseline + r1_point)
#   <tag>_bench_kernel_stats.{csv,txt} rocprofv3 --kernel-trace --stats of the same command (fewer steps)
#   <tag>_pmc_traffic.txt              FETCH_SIZE / WRITE_SIZE per GEMM role (separate --pmc passes, gfx950-corrected) -> profiles/roofline_traffic.json
#   <tag>_mfma_util.json               SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles) per kernel
# Counter passes use --kernel-trace only (gpurun refuses --pmc together with the sys / hip / hsa trace domains).
set -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT" profiles
python3 bench.py --steps 10 --warmup 3 > profiles/${TAG}_bench.json 2> "$OUT/bench.err" || { echo "bench failed"; tail -5 "$OUT/bench.err"; exit 1; }
SHORT="bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-r1-point"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 $SHORT > "$OUT/stats.log" 2>&1 || { echo "kernel-trace failed"; tail -5 "$OUT/stats.log"; exit 1; }
STATS=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
cp "$STATS" profiles/${TAG}_bench_kernel_stats.csv
python3 - "$STATS" > profiles/${TAG}_bench_kernel_stats.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
print(f"{'kernel':78s} {'calls':>6s} {'total ms':>10s} {'avg us':>10s} {'%':>6s}")
for r in rows[:40]:
    print(f"{r['Name'][:78]:78s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.1f} {float(r['Percentage']):6.2f}")
PY
# the literal BASELINE config (B = 4, R = 1): per-kernel stats of its own
R1="bench.py --repeats 1 --steps 20 --warmup 5 --no-cpu-baseline --no-r1-point"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_r1" -o s -- python3 $R1 > "$OUT/stats_r1.log" 2>&1 || { echo "R=1 kernel-trace failed"; tail -5 "$OUT/stats_r1.log"; exit 1; }
STATS_R1=$(find "$OUT/stats_r1" -name '*kernel_stats.csv' | head -1)
python3 - "$STATS_R1" > profiles/${TAG}_r1_kernel_stats.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("rocprofv3 --kernel-trace --stats -- python3 bench.py --repeats 1 --steps 20 --warmup 5 --no-cpu-baseline --no-r1-point   (B = 4: the literal BASELINE config; 25 steps incl. warm-up)")
print(f"{'kernel':100s} {'calls':>6s} {'total ms':>10s} {'avg us':>10s} {'%':>6s}")
for r in rows[:24]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.1f} {float(r['Percentage']):6.2f}")
print(f"all kernels: {tot / 1e6:.3f} ms = {tot / 1e6 / 25:.3f} ms per step")
PY
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq_r1" -o p -- python3 $R1 > "$OUT/pmc_sq_r1.log" 2>&1 || { echo "R=1 pmc sq failed"; tail -5 "$OUT/pmc_sq_r1.log"; exit 1; }
Q1=$(find "$OUT/pmc_sq_r1" -name '*counter_collection.csv' | head -1)
python3 profiles/pmc_sq.py "$Q1" "$STATS_R1" > profiles/${TAG}_r1_mfma_util.json
cp profiles/${TAG}_r1_kernel_stats.txt profiles/${TAG}_r1_mfma_util.json gpurun_out/ 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o p -- python3 $SHORT > "$OUT/pmc_$C.log" 2>&1 || { echo "pmc $C failed"; tail -5 "$OUT/pmc_$C.log"; exit 1; }
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o p -- python3 $SHORT > "$OUT/pmc_sq.log" 2>&1 || { echo "pmc sq failed"; tail -5 "$OUT/pmc_sq.log"; exit 1; }
F=$(find "$OUT/pmc_FETCH_SIZE" -name '*counter_collection.csv' | head -1); W=$(find "$OUT/pmc_WRITE_SIZE" -name '*counter_collection.csv' | head -1); Q=$(find "$OUT/pmc_sq" -name '*counter_collection.csv' | head -1)
python3 profiles/pmc_traffic.py "$F" "$W" 64 "profiles/${TAG}_pmc_traffic.txt" > profiles/${TAG}_pmc_traffic.txt
python3 profiles/pmc_sq.py "$Q" "$STATS" > profiles/${TAG}_mfma_util.json
cp profiles/${TAG}_bench.json profiles/${TAG}_pmc_traffic.txt profiles/${TAG}_mfma_util.json profiles/${TAG}_bench_kernel_stats.txt profiles/${TAG}_bench_kernel_stats.csv profiles/roofline_traffic.json gpurun_out/ 2>/dev/null
head -12 profiles/${TAG}_bench_kernel_stats.txt; head -12 profiles/${TAG}_pmc_traffic.txt
