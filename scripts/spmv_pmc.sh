#!/bin/bash
# counters of the SpMV kernels (one group per pass; rocprofv3 --pmc only).  usage: scripts/spmv_pmc.sh [probe args]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/spmv_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 $R/scripts/spmv_probe.py "$@" > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_spmv' not in k and 'k_csr_vec' not in k: continue
        a = acc[(k.split('<')[0], r['Counter_Name'])]; a[0] += float(r['Counter_Value']); a[1] += 1
for (kn, c), (v, n) in sorted(acc.items()):
    print(f"{kn:24s} {c:28s} avg/dispatch {v / n:16.1f} dispatches {n}")
PY
