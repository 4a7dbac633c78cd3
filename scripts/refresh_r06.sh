#!/bin/bash
# round 6, after the host-operand pipeline: the GPU suite, the bench line of the driver's command (timed), and rocprofv3 kernel stats of the same command
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
out=$R/gpurun_out/r06b
rm -rf $out; mkdir -p $out
( time timeout 2400 python -m pytest tests -q -m gpu ) > $out/gpu_tests.txt 2>&1; echo "rc=$?" >> $out/gpu_tests.txt
tail -5 $out/gpu_tests.txt
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err ) 2> $out/bench_time.txt
tail -1 $out/bench_n1.json | cut -c1-300; cat $out/bench_time.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extra > $out/bench_under_rocprof.log 2>&1
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/stats
head -5 $out/kernel_stats.csv | cut -c1-200
