#!/bin/bash
# kernel-time breakdown of any python command (run on the GPU box): scripts/profile_cmd.sh <script.py> [args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_cmd
rm -rf $out; mkdir -p $out
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o cmd -- python3 "$@" > $out.log 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$out/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(f'{r["Name"][:110]:110s} calls {r["Calls"]:>5s} total_ms {float(r["TotalDurationNs"])/1e6:9.3f} avg_us {float(r["AverageNs"])/1e3:9.1f}')
PY
tail -3 $out.log
