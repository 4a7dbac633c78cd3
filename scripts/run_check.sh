cd $GRAFT_REPO_ROOT
L=gpurun_out/check.log
: > $L
timeout 300 python scripts/exp_lds.py quick 2>&1 | grep -v "^small n" | grep -v amdgpu >> $L
timeout 200 python scripts/exp_lds_one.py --waves 16 2>&1 | tail -1 >> $L
timeout 200 python scripts/exp_lds_one.py --waves 16 --clustered 2>&1 | tail -1 >> $L
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> $L
timeout 900 python -m pytest tests/test_bench_gpu.py -q -m gpu -k "own_ranks or one_json" 2>&1 | tail -3 >> $L
timeout 900 python -m pytest tests/test_lds_gpu.py -q -m gpu 2>&1 | tail -2 >> $L
cat $L | cut -c1-160
