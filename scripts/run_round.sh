cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lds_gpu.py tests/test_autotune_gpu.py -q -m gpu > gpurun_out/test_lds_gpu.log 2>&1; echo "lds rc=$?" >> gpurun_out/test_lds_gpu.log
grep -a "autotune\]" gpurun_out/test_lds_gpu.log | head -3; tail -5 gpurun_out/test_lds_gpu.log
for vm in 1 2; do timeout 300 python scripts/exp_lds_one.py --waves 16 --mode $vm --tune lds_weighted_probe=0 2>/dev/null | tail -1; done
timeout 600 python scripts/exp_weighted.py > gpurun_out/exp_weighted.log 2>&1; grep -v amdgpu gpurun_out/exp_weighted.log
bash scripts/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -5 gpurun_out/profile_round.log
