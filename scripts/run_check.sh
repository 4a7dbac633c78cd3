cd $GRAFT_REPO_ROOT
L=gpurun_out/check.log
: > $L
timeout 900 python -m pytest tests/test_lds_gpu.py -q -m gpu 2>&1 | tail -3 >> $L
timeout 200 python scripts/exp_lds_one.py --waves 16 2>&1 | tail -1 >> $L
timeout 200 python scripts/exp_lds_one.py --waves 16 --clustered 2>&1 | tail -1 >> $L
timeout 200 python scripts/exp_lds_one.py --waves 16 --clustered --tune lds_long_slots=0 2>&1 | tail -1 >> $L
timeout 200 python scripts/exp_lds_one.py --waves 16 --tune lds_long_slots=1 2>&1 | tail -1 >> $L
timeout 900 python -m pytest tests/test_gnn_gpu.py -q -m gpu 2>&1 | tail -2 >> $L
cat $L | cut -c1-170
