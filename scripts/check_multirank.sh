cd $GRAFT_REPO_ROOT
PYGIM_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/bench_n2_gloo.json 2> gpurun_out/bench_n2_gloo.err; echo rc=$?
tail -c 1500 gpurun_out/bench_n2_gloo.json; tail -5 gpurun_out/bench_n2_gloo.err
# (four ranks on ONE GPU: the small shape -- four processes each building the 115 M-entry graph with torch on one device did not finish
# the graph construction in 5 minutes on the round-3 boxes, before any library call; a rank's watchdog, PYGIM_RANK_TIMEOUT, now says where it waits)
PYGIM_BENCH_BACKEND=gloo PYGIM_RANK_TIMEOUT=600 timeout 900 python bench.py --gpus 4 --steps 3 --warmup 1 --shape products-mini --partition pipelined-feature > gpurun_out/bench_n4_gloo.json 2> gpurun_out/bench_n4_gloo.err; echo rc=$?
tail -c 800 gpurun_out/bench_n4_gloo.json
timeout 600 python bench.py --clustered --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_clustered.json 2>/dev/null; tail -c 600 gpurun_out/bench_clustered.json
