#!/bin/bash
# PMC passes over one sweep configuration: scripts/exp_sweep_pmc.sh <tag> [exp_sweep_one.py args...]
tag=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/sweeppmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "GRBM_GUI_ACTIVE TCC_BUSY_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $out/pass$i -- python3 $R/scripts/exp_sweep_one.py --reps 1 "$@" > $out/pass$i.log 2>&1
  echo "pass$i [$grp] rc=$?"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "k_csr_panel" in k:
            a = agg["k_csr_panel"][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
with open(sys.argv[1] + "/summary.txt", "w") as o:
    for k, d in agg.items():
        print(k, file=o)
        for c, v in sorted(d.items()):
            print(f"    {c:34s} {v[0] / v[1]:18.1f} per launch ({v[1]} launches)", file=o)
print(open(sys.argv[1] + "/summary.txt").read())
PY
