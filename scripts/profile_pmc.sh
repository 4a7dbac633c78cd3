#!/bin/bash
# PMC passes for the bench step (one counter group per pass; --pmc never combined with tracing).
# usage: scripts/profile_pmc.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/pass*/...
tag=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH" \
           "GRBM_GUI_ACTIVE TCC_BUSY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $out/pass$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra "$@" > $out/pass$i.log 2>&1
  echo "pass$i [$grp] rc=$?"
done
python3 $R/scripts/summarize_pmc.py $out
