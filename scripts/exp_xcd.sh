#!/bin/bash
# Round 5: slices of X per XCD for the code-stream product -- time and fabric traffic.
# lds_xcd_slices = 1: an XCD streams ONE slice of X (round 4); 2 / 4: the workgroups an XCD runs side by side are slices of the same
# tile and share its code stream in that XCD's L2.  Writes gpurun_out/xcd/summary.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/xcd
rm -rf $out; mkdir -p $out
cd /tmp
for sx in 1 2 4; do
  for extra in "" "--clustered"; do
    tag="sx${sx}${extra:+_clustered}"
    python3 $R/scripts/exp_code_geo.py --reps 9 --tune lds_xcd_slices=$sx $extra 0:0:0:0:0 2>&1 | grep -v amdgpu.ids > $out/time_$tag.txt
    i=0
    for grp in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
      i=$((i+1))
      timeout 300 rocprofv3 --pmc $grp --output-format csv -d $out/$tag/pass$i -- python3 $R/scripts/exp_code_geo.py --reps 2 --tune lds_xcd_slices=$sx $extra 0:0:0:0:0 > $out/$tag.pass$i.log 2>&1
    done
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
        # gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide read (MI355X_MICROARCH.md) -> doubled; units of KB
        fetch = 2 * m.get("FETCH_SIZE", 0) * 1024 / 1e9
        write = m.get("WRITE_SIZE", 0) * 1024 / 1e9
        hit, miss = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
        t = open(os.path.join(out, f"time_{tag}.txt")).read().strip().splitlines()[-1]
        print(f"{tag:16s} fetch {fetch:6.2f} GB  write {write:5.2f} GB  traffic {fetch + write:6.2f} GB  EA_RDREQ {m.get('TCC_EA0_RDREQ_sum', 0) / 1e6:7.1f} M  "
              f"L2 hit {100 * hit / max(hit + miss, 1):5.1f} %", file=o)
        print("    " + t[:150], file=o)
print(open(out + "/summary.txt").read())
PY
