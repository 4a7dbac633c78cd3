cd $GRAFT_REPO_ROOT
L=gpurun_out/exp_lds_variants.log
: > $L
timeout 300 python scripts/exp_lds.py quick >> $L 2>&1
for w in 8 16; do
  _LW=$w timeout 200 python scripts/exp_lds_one.py --waves $w --clustered >> $L 2>&1
  timeout 200 python scripts/exp_lds_one.py --waves $w >> $L 2>&1
done
timeout 200 python scripts/exp_lds_one.py --waves 8 --ablate 4 >> $L 2>&1
timeout 200 python scripts/exp_lds_one.py --waves 8 --ablate 3 >> $L 2>&1
timeout 200 python scripts/exp_lds_one.py --waves 8 --ablate 1 >> $L 2>&1
grep -v amdgpu.ids $L
bash scripts/lds_pmc.sh w16u --waves 16 > gpurun_out/lds_pmc_w16u.log 2>&1
