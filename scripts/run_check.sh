cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_lds_gpu.py -q -m gpu 2>&1 | tail -8 | cut -c1-300
timeout 900 python scripts/exp_configs.py --cases "reddit:CSR:i16:256,reddit:CSR:i16:128,reddit:COO:i16:256" 2>&1 | grep -v amdgpu
