cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_stress_gpu.py tests/test_lds_gpu.py -q -m gpu -x 2>&1 | tail -8 | cut -c1-300
