#!/bin/bash
# end-to-end inference.py table (profiles/r03_inference_table.txt): best of 5 repeats per (model, adjacency type, --fuse_post)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in gcn sage gin; do for dt in FLT32 INT32 INT8; do for fp in 0 1; do
  [ $m != gcn ] && [ $fp = 1 ] && continue   # (the fused epilogue is a GCN-layer feature)
  timeout 300 python inference.py --dataset Reddit --num_layers 3 --hidden_size 256 --version spmm --model $m --data_type $dt --repeat 5 --fuse_post $fp 2>&1 \
    | grep infer_time | sort -t: -k2 -n | head -1 | sed "s/^/$m $dt fuse_post=$fp /"
done; done; done
