cd $GRAFT_REPO_ROOT
L=gpurun_out/exp_lds_ablate.log
: > $L
for a in 0 1 2 3 4; do timeout 200 python scripts/exp_lds_one.py --clustered --ablate $a >> $L 2>&1; done
for a in 0 1 2 3 4; do timeout 200 python scripts/exp_lds_one.py --ablate $a >> $L 2>&1; done
timeout 200 python scripts/exp_lds_one.py --waves 16 --clustered >> $L 2>&1
timeout 200 python scripts/exp_lds_one.py --waves 16 >> $L 2>&1
timeout 200 python scripts/exp_lds_one.py --waves 16 --dtype i32 >> $L 2>&1
grep -v amdgpu.ids $L
bash scripts/lds_pmc.sh w8c --clustered > gpurun_out/lds_pmc_w8c.log 2>&1
bash scripts/lds_pmc.sh w8u > gpurun_out/lds_pmc_w8u.log 2>&1
tail -50 gpurun_out/lds_pmc_w8c.log
