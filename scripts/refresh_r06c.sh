#!/bin/bash
# round 6, after the device generator learnt half-split plans (lds_half_split = 1 by default): GPU suite, shard table, 8 ranks on one GPU, bench line
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
out=$R/gpurun_out/r06c
rm -rf $out; mkdir -p $out
( time timeout 2400 python -m pytest tests -q -m gpu ) > $out/gpu_tests.txt 2>&1; echo "rc=$?" >> $out/gpu_tests.txt
tail -5 $out/gpu_tests.txt
python3 scripts/exp_shard.py "" FLT32 full,r2,r4,r8,h128,h64,h32,g24,g42 2>&1 | grep -v amdgpu.ids | cut -c1-260 > $out/exp_shard.txt
python3 scripts/exp_shard.py "lds_half_split=0" FLT32 h32 2>&1 | grep -v amdgpu.ids | cut -c1-260 >> $out/exp_shard.txt
python3 scripts/exp_shard.py "" INT32 h32 2>&1 | grep -v amdgpu.ids | cut -c1-260 >> $out/exp_shard.txt
cat $out/exp_shard.txt | cut -c1-150
bash scripts/check_multirank.sh 8 reddit 240 > $out/multirank.txt 2>&1
grep "^ranks=\|candidate" $out/multirank.txt | cut -c1-200
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err
tail -1 $out/bench_n1.json | cut -c1-200
