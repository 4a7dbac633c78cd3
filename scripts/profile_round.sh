#!/bin/bash
# Round profile refresh on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC passes.
# Results land in gpurun_out/round/ ; copy what should be judged into profiles/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/round
rm -rf $out; mkdir -p $out
cd $R
bash scripts/profile_pmc.sh round > $out/pmc.log 2>&1
cp gpurun_out/pmc_round/summary.txt $out/pmc_summary.txt 2>/dev/null
cp gpurun_out/pmc_round/traffic.json $out/traffic.json 2>/dev/null
cp $out/traffic.json profiles/traffic_latest.json 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-extra > $out/bench_under_rocprof.log 2>&1
cd $R
python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.err
tail -1 $out/bench_n1.json | cut -c1-400
find $out -name "*kernel_stats.csv" | head -2
