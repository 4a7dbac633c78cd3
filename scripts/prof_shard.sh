export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr8 -o r8 -- python3 $GRAFT_REPO_ROOT/scripts/exp_shard.py "" FLT32 r8 > /tmp/pr8.log 2>&1
f=$(find /tmp/pr8 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "pygim" in r["Name"] and int(r["Calls"])>=5: print(r["Name"][:60].ljust(62), r["Calls"].rjust(5), "avg us", round(float(r["AverageNs"])/1e3,1))
PY
