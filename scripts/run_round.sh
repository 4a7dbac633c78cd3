#!/bin/bash
# Round check on the GPU box: the whole GPU test suite, then the profile refresh (bench line, rocprofv3 kernel stats, PMC passes).
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/test_gpu_all.log 2>&1; echo "all rc=$?" >> gpurun_out/test_gpu_all.log
tail -6 gpurun_out/test_gpu_all.log
bash scripts/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -3 gpurun_out/profile_round.log | cut -c1-300
