#!/bin/bash
# kernel-time breakdown of one inference.py configuration (run on the GPU box): args = inference.py flags
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_inf
rm -rf $out; mkdir -p $out
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o inf -- python3 $R/inference.py "$@" > $out.log 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$out/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:22]:
    print(f'{r["Name"][:120]:120s} calls {r["Calls"]:>5s} total_ms {float(r["TotalDurationNs"])/1e6:9.3f} avg_us {float(r["AverageNs"])/1e3:9.1f} {r["Percentage"]}%')
PY
grep "infer_time" $out.log | tail -3
