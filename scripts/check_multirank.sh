#!/bin/bash
# N > 1 logic checks of bench.py on ONE GPU (gloo between processes that share the device): never a measurement.
# usage: scripts/check_multirank.sh [ranks] [shape]     (default: 8 ranks, the full Reddit shape, every --partition choice)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=${1:-8}
SHAPE=${2:-reddit}
LIMIT=${3:-1700}     # seconds per launch (8 ranks that share ONE GPU can starve one another for minutes: bound every launch)
mkdir -p gpurun_out/multirank
for part in auto row feature pipelined pipelined-feature; do
  t0=$(date +%s)
  PYGIM_BENCH_BACKEND=gloo PYGIM_RANK_TIMEOUT=$((LIMIT - 60)) PYGIM_COLLECTIVE_TIMEOUT=$((LIMIT - 120)) PYGIM_LAUNCH_TIMEOUT=$LIMIT timeout $((LIMIT + 30)) python bench.py --gpus $R --steps 3 --warmup 1 --shape $SHAPE --partition $part \
      > gpurun_out/multirank/n${R}_${SHAPE}_${part}.json 2> gpurun_out/multirank/n${R}_${SHAPE}_${part}.err
  rc=$?
  echo "ranks=$R shape=$SHAPE partition=$part rc=$rc seconds=$(( $(date +%s) - t0 ))"
  python3 - "$R" "$part" "$SHAPE" <<'PY'
import json, sys
r, part, shape = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    d = json.loads([l for l in open(f"gpurun_out/multirank/n{r}_{shape}_{part}.json") if l.startswith("{")][0])
    fam = sorted({g["kernel"] + (f" S={g['col_splits']}" if g["col_splits"] > 1 else "") for pr in d["config"]["per_rank"] for g in pr["groups"]})
    print(f"   n_gpus {d['n_gpus']}  candidate {d['config']['candidate']}  ms/step {d['ms_per_step']}  check: {d['check'][:90]}")
    print(f"   kernels {fam}  create ms per rank {[pr['group_create_ms'] for pr in d['config']['per_rank']]}  threads {d['config']['per_rank'][0]['plan_threads']}")
except Exception as e:
    print("   no JSON line:", e)
PY
  grep -h "\[bench\] rank" gpurun_out/multirank/n${R}_${SHAPE}_${part}.err | cut -c1-260 | head -8
done
