#!/bin/bash
# PMC passes over one LDS-staged product configuration: scripts/lds_pmc.sh <tag> [exp_lds_one.py args...]
tag=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/hypmc_$tag
mkdir -p $out
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE TCC_BUSY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $out/pass$i -- python3 $R/scripts/exp_hybrid.py --kinds sbm --reps 2 "$@" > $out/pass$i.log 2>&1
  echo "pass$i [$grp] rc=$?"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "k_lds" in k or "k_csr_panel" in k or "k_slice_pack" in k:
            a = agg[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
with open(sys.argv[1] + "/summary.txt", "w") as o:
    for k, d in agg.items():
        print(k, file=o)
        for c, v in sorted(d.items()):
            print(f"    {c:34s} {v[0] / v[1]:18.1f} per launch ({v[1]} launches)", file=o)
print(open(sys.argv[1] + "/summary.txt").read())
PY
