#!/usr/bin/env python3
"""Collapse rocprofv3 --pmc counter_collection CSVs into per-kernel averages (one line per kernel/counter)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: [0.0, 0])
dur = defaultdict(lambda: [0.0, 0])   # kernel -> [sum of (End - Start) ns, dispatches] of the counter passes (one row per dispatch and counter)
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if "pygim" not in k:
                continue
            name = k.split("(")[0].replace("void pygim::", "")
            key = (name, row["Counter_Name"])
            agg[key][0] += float(row["Counter_Value"])
            agg[key][1] += 1
            if row["Counter_Name"] == "FETCH_SIZE" and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                dur[name][0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                dur[name][1] += 1
lines = []
for (k, c), (tot, n) in sorted(agg.items()):
    lines.append(f"{k:40s} {c:34s} avg/dispatch {tot / n:18.1f}  dispatches {n}")
txt = "\n".join(lines)
print(txt)
open(os.path.join(root, "summary.txt"), "w").write(txt + "\n")

# traffic of the dominant kernel, corrected as MI355X_MICROARCH.md (HBM section) prescribes:
# FETCH_SIZE (KiB) reads 1/2 of the bytes of wide reads on gfx950 -> doubled; WRITE_SIZE (KiB) exact.
dom = None
for (k, c), (tot, n) in agg.items():
    if c == "FETCH_SIZE" and ("k_lds_spmm" in k or "k_lds_code" in k or "k_csr_panel" in k or "k_csr_wide" in k or "k_coo_wide" in k):
        if dom is None or tot > agg[(dom, "FETCH_SIZE")][0]:
            dom = k
if dom and (dom, "WRITE_SIZE") in agg:
    import json
    f_tot, f_n = agg[(dom, "FETCH_SIZE")]
    w_tot, w_n = agg[(dom, "WRITE_SIZE")]
    per_launch = (2.0 * f_tot / f_n + w_tot / w_n) * 1024.0
    # launches per product: the sweep's dispatches over those of k_slice_pack (one per product), else the env override
    packs = [n for (k, c), (tot, n) in agg.items() if c == "FETCH_SIZE" and k.startswith("k_slice_pack")]
    launches_per_product = int(os.environ.get("LAUNCHES_PER_PRODUCT", "0")) or (max(1, round(f_n / packs[0])) if packs else 1)
    hit = agg.get((dom, "TCC_HIT_sum"), [0, 1])
    miss = agg.get((dom, "TCC_MISS_sum"), [0, 1])
    import datetime
    import socket

    rec = {"collected": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), "box": socket.gethostname(),
           "command": "rocprofv3 --pmc <one counter group per pass> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "
                      "(scripts/profile_pmc.sh)",
           "kernel": dom, "kernel_ms": (dur[dom][0] / dur[dom][1] * 1e-6 if dur[dom][1] else None),   # of the FETCH_SIZE pass: bench.py replays the
           # traffic only when its own kernel time is within 10 % of this
           "hbm_bytes_per_launch": per_launch, "launches_per_product": launches_per_product,
           "hbm_bytes_per_product": per_launch * launches_per_product,
           "fetch_size_kib_avg": f_tot / f_n, "write_size_kib_avg": w_tot / w_n,
           "l2_hit_rate": hit[0] / max(hit[0] + miss[0], 1),
           "correction": "2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes), per MI355X_MICROARCH.md HBM section"}
    json.dump(rec, open(os.path.join(root, "traffic.json"), "w"), indent=1)
    print(json.dumps(rec))
