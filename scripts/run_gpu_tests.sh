cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/test_gpu_all.log 2>&1; echo "all rc=$?" >> gpurun_out/test_gpu_all.log
grep -a "autotune\]" gpurun_out/test_gpu_all.log | head -3
tail -25 gpurun_out/test_gpu_all.log
