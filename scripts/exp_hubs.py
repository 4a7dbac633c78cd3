#!/usr/bin/env python3
"""Round 5: hub rows (R-MAT) -- what splitting rows longer than a wave's mean load into pieces would buy.
The LDS-staged product gives every row to ONE wave; a row of 800 K entries is then a serial chain.  Here the same graph is fed with its long rows cut into
pieces as separate rows of a taller matrix (row pointers with extra split points, column ids untouched): the product of that matrix is what a plan with
split rows would run before its final add of the pieces."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="reddit")
ap.add_argument("--kind", default="rmat")
ap.add_argument("--h", type=int, default=256)
ap.add_argument("--limits", default="0,200000,100000,50000,25000")
args = ap.parse_args()
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES[args.shape]
rowptr, col = synth.make_shape(args.shape, seed=0, device=dev, kind=args.kind)
deg = (rowptr[1:] - rowptr[:-1]).long()
top = torch.sort(deg, descending=True).values[:12].tolist()
print(f"# {args.shape}-shaped {args.kind}: longest rows {top}; mean entries per wave of a tile (1 824 rows, 8 waves): {nnz / ((n + 1823) // 1824) / 8:.0f}")
x = synth.features(n, args.h, torch.int32, seed=0, device=dev)
_lib.set_tunable("lds_mode", 1)
_lib.set_tunable("lds_tile_order", 0)
for lim in [int(v) for v in args.limits.split(",")]:
    if lim:
        hub = torch.nonzero(deg > lim).flatten()
        pts = []
        for r in hub.tolist():
            l = int(deg[r]); k = (l + lim // 2 - 1) // (lim // 2); step = (l + k - 1) // k
            pts.append(int(rowptr[r]) + torch.arange(1, k, device=dev, dtype=torch.int64) * step)
        rp = torch.sort(torch.cat([rowptr.long()] + pts)).values.to(torch.int32) if pts else rowptr
    else:
        rp = rowptr
    nr = rp.numel() - 1
    hd = _lib.group_create(_lib.CSR, _lib.INT32, [rp.data_ptr()], [col.data_ptr()], None, [nr], [n], [nnz], [1], [args.h], args.h)
    out = torch.empty((nr, args.h), dtype=torch.int32, device=dev)
    for _ in range(2):
        _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(7):
        a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    print(f"  rows longer than {lim:7d} cut ({nr - n:5d} more rows): {min(ts):7.3f} ms   {_lib.group_lds_note(hd)[:60]}", flush=True)
    _lib.group_free(hd)
