#!/bin/bash
# L2 residency probe: single-panel sweeps over restricted column ranges, TCC hit/miss per launch.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for nc in "$@"; do
  out=$R/gpurun_out/pmc_probe/nc$nc
  mkdir -p $out
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $out -- python3 $R/scripts/exp_panel.py --budgets 64 --ncols $nc > $out.log 2>&1
  echo "ncols $nc: $(grep slice-major $out.log | sed 's/  */ /g' | cut -d' ' -f1-9)"
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_csr_panel" in row["Kernel_Name"]:
            a = agg[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
print("   ", {k: round(v[0] / v[1] / 1e6, 2) for k, v in agg.items()}, "M per launch, launches", {k: v[1] for k, v in agg.items()})
PY
done
