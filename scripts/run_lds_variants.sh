cd $GRAFT_REPO_ROOT
L=gpurun_out/exp_lds_variants.log
: > $L
timeout 300 python scripts/exp_lds.py quick >> $L 2>&1
for w in 8 16; do
  timeout 200 python scripts/exp_lds_one.py --waves $w --clustered >> $L 2>&1
  timeout 200 python scripts/exp_lds_one.py --waves $w >> $L 2>&1
done
timeout 200 python scripts/exp_lds_one.py --waves 16 --tune lds_round_tiles=0 >> $L 2>&1
timeout 200 python scripts/exp_lds_one.py --waves 16 --dtype i32 >> $L 2>&1
timeout 200 python scripts/exp_lds_one.py --waves 16 --h 128 >> $L 2>&1
timeout 200 python scripts/exp_lds_one.py --waves 16 --h 64 >> $L 2>&1
timeout 200 python scripts/exp_lds_one.py --waves 16 --h 100 >> $L 2>&1
timeout 200 python scripts/exp_lds_one.py --mode 2 --h 128 >> $L 2>&1
timeout 200 python scripts/exp_lds_one.py --mode 2 --h 64 >> $L 2>&1
timeout 200 python scripts/exp_lds_one.py --mode 2 --h 100 >> $L 2>&1
grep -v amdgpu.ids $L
bash scripts/lds_pmc.sh w16u --waves 16 > gpurun_out/lds_pmc_w16u.log 2>&1
