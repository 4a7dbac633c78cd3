cd ${GRAFT_REPO_ROOT:-/root/repo}
for a in 0 17 18 19 0 17 18 19; do timeout 200 python scripts/exp_lds_one.py --waves 16 --ablate $a 2>&1 | grep -v amdgpu.ids | cut -c1-120; done
for a in 0 17 18 19; do timeout 200 python scripts/exp_lds_one.py --waves 16 --h 64 --ablate $a 2>&1 | grep -v amdgpu.ids | cut -c1-120; done
for a in 0 19; do timeout 200 python scripts/exp_lds_one.py --waves 16 --clustered --ablate $a 2>&1 | grep -v amdgpu.ids | cut -c1-120; done
