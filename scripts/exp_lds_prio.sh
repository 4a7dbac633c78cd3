cd ${GRAFT_REPO_ROOT:-/root/repo}
L=gpurun_out/exp_lds_prio.log
: > $L
for a in 0 15 16 0 15 16; do timeout 200 python scripts/exp_lds_one.py --waves 16 --ablate $a >> $L 2>&1; done
for a in 0 15 16; do timeout 200 python scripts/exp_lds_one.py --waves 16 --clustered --ablate $a >> $L 2>&1; done
grep -v amdgpu.ids $L | cut -c1-160
