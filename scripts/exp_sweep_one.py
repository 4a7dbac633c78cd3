#!/usr/bin/env python3
"""One products-shaped SBM sweep configuration for rocprofv3: exp_sweep_one.py [--p_in 1.0] [--locality 1] [--shuffle 0] [--reps 2]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pygim_amd import _lib, synth
ap = argparse.ArgumentParser()
ap.add_argument("--p_in", type=float, default=1.0)
ap.add_argument("--locality", type=int, default=1)
ap.add_argument("--shuffle", type=int, default=0)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--shape", default="ogbn-products")
ap.add_argument("--tune", default="")
args = ap.parse_args()
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES[args.shape]
x = synth.features(n, 256, torch.int32, seed=0, device=dev)
out = torch.empty((n, 256), dtype=torch.int32, device=dev)
rowptr, col = synth.make_sbm(n, nnz, dmax, synth.SBM_BLOCKS.get(args.shape, 50), p_in=args.p_in, seed=0, device=dev, shuffle=bool(args.shuffle))
_lib.set_tunable("panel_locality", args.locality)
_lib.set_tunable("lds_mode", 2)
for kv in filter(None, args.tune.split(",")):
    k, v = kv.split("=")
    _lib.set_tunable(k, int(v))
hd = _lib.group_create(_lib.CSR, _lib.INT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [256], 256)
for _ in range(2): _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
torch.cuda.synchronize()
ts = []
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(args.reps):
    a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
print(f"{args.shape} p_in {args.p_in} locality {args.locality} shuffle {args.shuffle} {args.tune}: {min(ts):.3f} ms  {_lib.group_lds_tiles(hd)} plan {_lib.group_plan(hd)}", flush=True)
