#!/bin/bash
# Ablation builds of the 16-wave LDS-staged kernel on the bench workload (timing only, wrong results): see scripts/gen_lds_kernel.py
# body(ablate=...) for the codes.  Results: profiles/r03_lds_kernel.md.
cd ${GRAFT_REPO_ROOT:-/root/repo}
make -C pygim_amd/csrc ablate > /dev/null   # (rebuild with `make -C pygim_amd/csrc clean all` afterwards)
L=gpurun_out/exp_lds_ablate.log
: > $L
for a in 0 6 7 10 11 12; do timeout 200 python scripts/exp_lds_one.py --waves 16 --ablate $a >> $L 2>&1; done
for a in 0 6; do timeout 200 python scripts/exp_lds_one.py --waves 16 --clustered --ablate $a >> $L 2>&1; done
grep -v amdgpu.ids $L | cut -c1-120
