#!/bin/bash
# LDS / issue counters of the prototype kernel (separate passes, rocprofv3 --pmc only)
cd "$(dirname "$0")"
export TMPDIR=/tmp
OUT=../../gpurun_out/lds_pmc
mkdir -p $OUT
i=0
for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 run_proto.py --check 0 --iters 2 "$@" > $OUT/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('../../gpurun_out/lds_pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_lds' not in r['Kernel_Name']: continue
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print(f"k_lds {k:28s} avg/dispatch {v / n:16.1f} dispatches {n}")
PY
