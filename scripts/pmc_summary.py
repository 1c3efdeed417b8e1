"""Per-kernel mean of every counter in a rocprofv3 --pmc run: python scripts/pmc_summary.py <dir> [name filter]"""
import csv
import sys
from collections import defaultdict
from pathlib import Path

acc = defaultdict(lambda: defaultdict(list))
for f in Path(sys.argv[1]).rglob("*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in acc.items():
    if flt in k:
        print(k)
        for c, v in sorted(cs.items()):
            print(f"    {c:32s} {sum(v) / len(v):16.0f}  (n={len(v)})")
