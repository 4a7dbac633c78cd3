cd $GRAFT_REPO_ROOT
L=gpurun_out/exp_lds_ablate.log
: > $L
timeout 300 python scripts/exp_lds.py quick >> $L 2>&1
for a in 0 13 6 7 10; do timeout 200 python scripts/exp_lds_one.py --waves 16 --ablate $a >> $L 2>&1; done
for a in 0 13; do timeout 200 python scripts/exp_lds_one.py --waves 16 --clustered --ablate $a >> $L 2>&1; done
timeout 200 python scripts/exp_lds_one.py --waves 8 >> $L 2>&1
timeout 200 python scripts/exp_lds_one.py --waves 8 --clustered >> $L 2>&1
timeout 600 python -m pytest tests/test_lds_gpu.py -q -m gpu -x 2>&1 | tail -3 >> $L
grep -v amdgpu.ids $L | grep -v "^small n" | cut -c1-120
