export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
rm -rf /tmp/ks
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o bench -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extra > $R/gpurun_out/r06/bench_noextra_under_rocprof.log 2>&1
find /tmp/ks -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/r06/kernel_stats_noextra.csv
grep -E "k_lds_code8_f32|k_slice_pack" $R/gpurun_out/r06/kernel_stats_noextra.csv | cut -c1-140
tail -c 400 $R/gpurun_out/r06/bench_noextra_under_rocprof.log
