#!/usr/bin/env python3
"""Round 5: the density split (lds_hybrid_dev.hpp) on community-structured graphs with shuffled ids.
exp_hybrid.py [--shape ogbn-products] [--h 256] [--dtype i32|f32|i16] [--kinds sbm,uniform] [--min 64,...]
Per graph: the part's own plan (lds_hybrid = 0) against the split (dense cells through the LDS-staged kernel, the rest through the sweep), results compared
element by element (integers: equal; floats: relative to the largest magnitude)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="ogbn-products")
ap.add_argument("--h", type=int, default=256)
ap.add_argument("--dtype", default="i32")
ap.add_argument("--kinds", default="sbm")
ap.add_argument("--min", default="64")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--format", default="CSR")
args = ap.parse_args()
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES[args.shape]
dt = {"f32": torch.float32, "i32": torch.int32, "i16": torch.int16, "i8": torch.int8}[args.dtype]
code = {torch.float32: _lib.FLT32, torch.int32: _lib.INT32, torch.int16: _lib.INT16, torch.int8: _lib.INT8}[dt]
x = synth.features(n, args.h, dt, seed=0, device=dev)
print(f"# {args.shape}-shaped (N = {n}, nnz = {nnz}), {args.dtype} h = {args.h}; ids shuffled", flush=True)
for kind in args.kinds.split(","):
    rowptr, col = synth.make_shape(args.shape, seed=0, device=dev, kind=None if kind == "uniform" else kind)
    fmt, idx0, vals, nnz_g = _lib.CSR, rowptr, None, nnz
    if args.format == "COO":   # as the reference builds it: coalesce() -- duplicates become weights > 1 (backend_pim/spmm.py:40-42)
        r_, col, vals = synth.csr_to_coo_coalesced(rowptr, col, dt)
        fmt, idx0, nnz_g = _lib.COO, r_, int(col.numel())
        print(f"  (COO, coalesced: {nnz_g} entries, {int((vals != 1).sum())} of them with a weight other than 1)", flush=True)
    ref = None
    for hy, mc in [(0, 0)] + [(2 if dt.is_floating_point else 1, int(v)) for v in args.min.split(",")]:
        _lib.set_tunable("lds_hybrid", hy)
        if mc:
            _lib.set_tunable("lds_hybrid_min", mc)
        torch.cuda.synchronize()
        t0 = time.time()
        hd = _lib.group_create(fmt, code, [idx0.data_ptr()], [col.data_ptr()], None if vals is None else [vals.data_ptr()], [n], [n], [nnz_g], [1], [args.h], args.h)
        torch.cuda.synchronize()
        t_create = (time.time() - t0) * 1e3
        out = torch.full((n, args.h), 77, dtype=dt, device=dev)
        for _ in range(2):
            _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
        torch.cuda.synchronize()
        ts = []
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(args.reps):
            a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
        if ref is None:
            ref = out.clone()
            ok = "(reference)"
        elif dt.is_floating_point:
            ok = f"max rel err {float((out - ref).abs().max() / ref.abs().max()):.2e}"
        else:
            ok = "EQUAL" if torch.equal(out, ref) else "MISMATCH"
        note = _lib.group_lds_note(hd)
        print(f"  {kind:8s} lds_hybrid={hy} min_cell={mc:4d}: {min(ts):8.3f} ms (median {sorted(ts)[len(ts) // 2]:8.3f})  create {t_create:7.1f} ms  lds runs {_lib.group_lds_runs(hd)}  {ok}\n      {note[-220:]}", flush=True)
        _lib.group_free(hd)
        del out
