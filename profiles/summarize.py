"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short table (stdout)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{r['Name'][:88]:88s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} "
          f"tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} {float(r['Percentage']):5.1f}%")
