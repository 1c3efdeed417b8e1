"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files (separate passes) into per-launch HBM
traffic per GEMM role, applying the gfx950 corrections of MI355X_MICROARCH.md section HBM:
FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams -> x2; WRITE_SIZE is exact for 16-B/lane stores.
Both counters are in KiB.  Writes profiles/roofline_traffic.json (read by bench.py)."""
import collections
import csv
import json
import re
import sys

ROLES = ["generic", "projector", "qkv", "attn_scores", "attn_pv", "out_proj", "ff1", "ff2", "voxel_head", "attention"]


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            n[k] += 1
    return {k: tot[k] / n[k] for k in tot}, n


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    m = re.search(r"gemm_nt_\w+<(\d), (\d+)(?:, \d+)*>", k)   # every tile kernel: 256x256x64<bf, role, tn, nb>, 4w256 / ring128 / 128x128x64<bf, role>
    if m:
        role = int(m.group(2))
        name = ROLES[role] if role < len(ROLES) else "ext_epilogue"
        if int(m.group(1)) == 1 and role == 0:
            name = "generic_bf16_out"
    else:
        name = "attention" if "attn_fwd" in k else k.split("(")[0][-40:]
    rd = fetch.get(k, 0.0) * 1024 * 2  # KiB -> bytes, x2 gfx950 under-count of wide reads
    wr = write.get(k, 0.0) * 1024
    out[name] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr, "launches_profiled": int(nf.get(k, nw.get(k, 0)))}
    print(f"{name:14s} read {rd / 1e6:9.1f} MB  write {wr / 1e6:9.1f} MB  per launch   ({k[:70]})")
# the per-launch figures only mean something for the batch they were collected at: usage
#   python profiles/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <B per GPU of that run>
batch = int(sys.argv[3]) if len(sys.argv) > 3 else None
source = sys.argv[4] if len(sys.argv) > 4 else "profiles/*_pmc_traffic.txt"
json.dump({k: v["hbm_bytes"] for k, v in out.items()} | {"_sequences_per_gpu": batch, "_source": source, "_detail": out},
          open("profiles/roofline_traffic.json", "w"), indent=1)
