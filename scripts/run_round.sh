cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/test_gpu_all.log 2>&1; echo "all rc=$?" >> gpurun_out/test_gpu_all.log
tail -6 gpurun_out/test_gpu_all.log
timeout 300 python scripts/exp_lds_one.py --waves 16 | tail -1
timeout 300 python scripts/exp_lds_one.py --waves 8 | tail -1
timeout 300 python scripts/exp_lds_one.py --waves 16 --clustered | tail -1
bash scripts/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -3 gpurun_out/profile_round.log | cut -c1-300
