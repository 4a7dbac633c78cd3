#!/usr/bin/env python3
"""One configuration of the LDS-staged product on the Reddit-shaped graph, for timing / rocprofv3:
exp_lds_one.py [--waves 8|16] [--ablate N] [--clustered] [--mode 1|2] [--reps R] [--dtype f32|i32] [--h H]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--waves", type=int, default=8)
ap.add_argument("--ablate", type=int, default=0)
ap.add_argument("--clustered", action="store_true")
ap.add_argument("--mode", type=int, default=1)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--h", type=int, default=256)
ap.add_argument("--shape", default="reddit")
ap.add_argument("--tune", default="", help="name=value,... extra tunables")
args = ap.parse_args()
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES[args.shape]
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev, clustered=args.clustered)
dt = {"f32": torch.float32, "i32": torch.int32, "i16": torch.int16}[args.dtype]
x = synth.features(n, args.h, dt, seed=0, device=dev)
_lib.set_tunable("lds_mode", args.mode)
_lib.set_tunable("lds_waves", args.waves)
_lib.set_tunable("lds_ablate", args.ablate)
for kv in filter(None, args.tune.split(",")):
    k, v = kv.split("=")
    _lib.set_tunable(k, int(v))
hd = _lib.group_create(_lib.CSR, {torch.float32: _lib.FLT32, torch.int32: _lib.INT32, torch.int16: _lib.INT16}[dt], [rowptr.data_ptr()], [col.data_ptr()], None,
                       [n], [n], [nnz], [1], [args.h], args.h)
out = torch.empty((n, args.h), dtype=dt, device=dev)
for _ in range(2):
    _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
torch.cuda.synchronize()
ts = []
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(args.reps):
    a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0); b.record(); b.synchronize()
    ts.append(a.elapsed_time(b))
lp = _lib.group_lds_plan(hd)
ok = ""
if args.ablate in (0, 13):
    colcount = torch.bincount(col.long(), minlength=n).double()
    ok = " checksum " + ("OK" if torch.equal(out.double().sum(0), colcount @ x.double()) else "MISMATCH")
print(f"waves={args.waves} ablate={args.ablate} clustered={args.clustered} mode={args.mode} {args.dtype} h={args.h} {args.tune}: "
      f"{min(ts):7.3f} ms (median {sorted(ts)[len(ts)//2]:7.3f}){ok}  plan={lp}", flush=True)
_lib.group_free(hd)
