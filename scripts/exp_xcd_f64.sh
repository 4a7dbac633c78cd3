#!/bin/bash
# Round 5: the 8-byte code-stream product (DBL64, Reddit-shaped, h = 256) under lds_xcd_slices -- time, fabric traffic, L2 hit rate.
# Question: is the 8-byte stream bound by the per-CU fill path or by what the L2s have to fetch from the fabric?  Writes gpurun_out/xcd64/summary.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/xcd64
rm -rf $out; mkdir -p $out
cd /tmp
for sx in 1 2 4; do
    tag="f64_sx${sx}"
    python3 $R/scripts/exp_code_geo.py --dtype f64 --reps 7 --tune lds_xcd_slices=$sx 0:0:0:0:0 2>&1 | grep -v amdgpu.ids > $out/time_$tag.txt
    i=0
    for grp in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
      i=$((i+1))
      timeout 300 rocprofv3 --pmc $grp --output-format csv -d $out/$tag/pass$i -- python3 $R/scripts/exp_code_geo.py --dtype f64 --reps 2 --tune lds_xcd_slices=$sx 0:0:0:0:0 > $out/$tag.pass$i.log 2>&1
    done
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
with open(out + "/summary.txt", "w") as o:
    for tag in sorted(os.listdir(out)):
        if not os.path.isdir(os.path.join(out, tag)):
            continue
        agg = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob(os.path.join(out, tag) + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "k_lds_code8" in row["Kernel_Name"]:
                    a = agg[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
        m = {k: v[0] / max(v[1], 1) for k, v in agg.items()}
        fetch = 2 * m.get("FETCH_SIZE", 0) * 1024 / 1e9
        write = m.get("WRITE_SIZE", 0) * 1024 / 1e9
        hit, miss = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
        t = open(os.path.join(out, f"time_{tag}.txt")).read().strip().splitlines()[-1]
        print(f"{tag:16s} fetch {fetch:6.2f} GB  write {write:5.2f} GB  traffic {fetch + write:6.2f} GB  EA_RDREQ {m.get('TCC_EA0_RDREQ_sum', 0) / 1e6:7.1f} M  "
              f"L2 hit {100 * hit / max(hit + miss, 1):5.1f} %", file=o)
        print("    " + t[:150], file=o)
print(open(out + "/summary.txt").read())
PY
rm -rf $out/*/pass*   # (the raw counter files are large)
