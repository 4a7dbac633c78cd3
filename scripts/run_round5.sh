#!/bin/bash
# round-5 evidence run on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC passes (traffic), config / inference tables,
# creation phases; results under gpurun_out/r05/ (copied into profiles/ by hand)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
out=$R/gpurun_out/r05
rm -rf $out; mkdir -p $out
bash scripts/profile_pmc.sh round > $out/pmc.log 2>&1
cp gpurun_out/pmc_round/summary.txt $out/pmc_summary.txt 2>/dev/null
cp gpurun_out/pmc_round/traffic.json $out/traffic.json 2>/dev/null
cp $out/traffic.json profiles/traffic_latest.json 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-extra > $out/bench_under_rocprof.log 2>&1
cd $R
PYGIM_PLAN_TIMING=1 python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.err
tail -1 $out/bench_n1.json | cut -c1-300
timeout 900 python scripts/exp_configs.py --cases reddit:CSR:f32:256,reddit:COO:i32:256,reddit:CSR:i32:256,reddit:CSR:i16:256,reddit:CSR:i8:256,reddit:CSR:f64:256,reddit:CSR:i64:256,reddit:CSR:f32:128,reddit:CSR:f32:64,reddit:CSR:f32:100,reddit:CSR:i8:100,reddit:CSR:f64:100,ogbn-products:COO:i32:256,ogbn-products:CSR:f32:256 2>&1 | grep -v amdgpu.ids > $out/config_table.txt
bash scripts/inference_table.sh 2>&1 | grep -v amdgpu.ids > $out/inference_table.txt
timeout 300 python scripts/exp_weighted.py 2>&1 | grep -v amdgpu.ids > $out/exp_weighted.txt
find $out -name "*kernel_stats.csv" | head -2
