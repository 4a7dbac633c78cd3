#!/bin/bash
# round-6 evidence run on the GPU box: headline PMC passes (traffic), rocprofv3 kernel stats of the driver's bench command, the bench line itself,
# PMC passes for BASELINE configs[2] and configs[4] (the sweep), the MFMA prototype's counters, shard / stamp tables; results under gpurun_out/r06/
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
out=$R/gpurun_out/r06
rm -rf $out; mkdir -p $out
bash scripts/profile_pmc.sh round > $out/pmc.log 2>&1
cp gpurun_out/pmc_round/summary.txt $out/pmc_summary.txt 2>/dev/null
cp gpurun_out/pmc_round/traffic.json $out/traffic.json 2>/dev/null
cp $out/traffic.json profiles/traffic_latest.json 2>/dev/null
for cfg in c3 c5a c5b; do bash scripts/cfg_pmc.sh $cfg > $out/cfgpmc_$cfg.log 2>&1; cp gpurun_out/cfgpmc_$cfg/summary.txt $out/cfgpmc_${cfg}_summary.txt; done
python3 - <<'PY'
import json, os
out = {}
for cfg in ("c3", "c5a", "c5b"):
    f = f"gpurun_out/cfgpmc_{cfg}/traffic.json"
    if os.path.exists(f):
        out[cfg] = json.load(open(f))
json.dump(out, open("gpurun_out/r06/cfg_traffic.json", "w"), indent=1)
json.dump(out, open("profiles/cfg_traffic_latest.json", "w"), indent=1)
PY
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_under_rocprof.log 2>&1
cd $R
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/stats
PYGIM_PLAN_TIMING=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err
tail -1 $out/bench_n1.json | cut -c1-300
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_cfg -o cfg -- python3 $R/scripts/exp_cfg_one.py c3 > $out/cfg_c3_under_rocprof.log 2>&1
find $out/stats_cfg -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_c3.csv; rm -rf $out/stats_cfg
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_cfg -o cfg -- python3 $R/scripts/exp_cfg_one.py c5 > $out/cfg_c5_under_rocprof.log 2>&1
find $out/stats_cfg -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_c5.csv; rm -rf $out/stats_cfg
# the MFMA prototype: timings, then its matrix-core counters
cd $R
python3 scripts/exp_mfma_cells.py clustered,sbm 26,52,103 4 2>&1 | grep -v amdgpu.ids > $out/mfma_cells.txt
cd /tmp
for grp in "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $out/mfma_pmc -- python3 $R/scripts/exp_mfma_cells.py sbm 52 4 > $out/mfma_pmc.log 2>&1
  python3 - "$out/mfma_pmc" >> $out/mfma_counters.txt <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "k_mfma_cells" in k:
            a = agg[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:34s} {v[0] / v[1]:18.1f} per launch ({v[1]} launches)")
PY
  rm -rf $out/mfma_pmc
done
cd $R
python3 scripts/exp_shard.py "" FLT32 full,r2,r4,r8,h128,h64,h32,g24,g42 2>&1 | grep -v amdgpu.ids | cut -c1-260 > $out/exp_shard.txt
python3 scripts/exp_shard.py "lds_col_split_f32=0,lds_min_width=33" FLT32 r8,h64,h32,g24,g42 2>&1 | grep -v amdgpu.ids | cut -c1-260 >> $out/exp_shard.txt
python3 scripts/exp_shard.py "" INT8 full,r8,h64,g24 2>&1 | grep -v amdgpu.ids | cut -c1-260 >> $out/exp_shard.txt
python3 scripts/exp_stamps.py "" full,r8,h64,g24,g42 2>&1 | grep -v amdgpu.ids > $out/stamps.txt
for e in 1 2 3; do python3 scripts/exp_shard.py "lds_code_exp=$e" FLT32 full 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $out/exp_ablate.txt; done
timeout 900 python scripts/exp_configs.py --cases reddit:CSR:f32:256,reddit:COO:i32:256,reddit:CSR:i32:256,reddit:CSR:i16:256,reddit:CSR:i8:256,reddit:CSR:f64:256,reddit:CSR:i64:256,reddit:CSR:f32:128,reddit:CSR:f32:64,reddit:CSR:f32:100,reddit:CSR:f32:32 2>&1 | grep -v amdgpu.ids > $out/config_table.txt
bash scripts/inference_table.sh 2>&1 | grep -v amdgpu.ids > $out/inference_table.txt
ls $out
