"""MFMA utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass.

  utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * kernel cycles),  kernel cycles = GRBM_GUI_ACTIVE / 8
(rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs and the MFMA busy cycles summed over all SIMDs; the busy
count of a bf16 GEMM equals 16 cycles per v_mfma_f32_16x16x32_bf16, i.e. its FLOPs / 1024 per SIMD-cycle.)
The effective clock is kernel cycles / kernel time from a kernel-stats file of the same workload.

usage: python profiles/pmc_sq.py <sq counter_collection.csv> <kernel_stats.csv> > profiles/<tag>_mfma_util.json"""
import collections
import csv
import json
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
avg_ns = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(sys.argv[2]))}
out = []
for k, v in agg.items():
    n = len(disp[k])
    busy, gui = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n, v.get("GRBM_GUI_ACTIVE", 0.0) / n
    if busy <= 0 or gui <= 0:
        continue
    cycles = gui / 8
    rec = {"kernel": k[:90], "dispatches": n, "mfma_busy_cycles": busy, "kernel_cycles": cycles, "mfma_util": round(busy / (1024 * cycles), 4)}
    if k in avg_ns:
        rec["avg_us_unprofiled"] = round(avg_ns[k] / 1e3, 1)
        rec["effective_clock_ghz"] = round(cycles / avg_ns[k], 3)   # profiled cycles over un-profiled time: indicative only
    out.append(rec)
json.dump(sorted(out, key=lambda r: -r["mfma_busy_cycles"] * r["dispatches"]), sys.stdout, indent=1)
