#!/usr/bin/env python3
"""Round 5: graphs WITH structure, node ids shuffled, under consecutive-row tiles and similarity tiles.
exp_structured.py [--shape reddit|ogbn-products] [--h 256] [--dtype f32|i32] [--kinds uniform,clustered,rmat,sbm,sbm-sorted] [--mode 0|1]
Per graph and tile order: kernel ms, chunk fills (32 KiB chunk landings per slice), the share of stored entries served by another entry's
LDS read, labels of the propagation, creation ms.  The checksum (column counts x features) is exact for the driver's integer-valued features."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="reddit")
ap.add_argument("--h", type=int, default=256)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--kinds", default="uniform,clustered,rmat,sbm,sbm-sorted")
ap.add_argument("--mode", type=int, default=1, help="lds_mode: 1 = LDS-staged product whenever planned, 0 = by the reuse rule")
ap.add_argument("--orders", default="0,1")
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--locality", default="", help="panel_locality values to try per graph (the sweep's item order), e.g. 0,1")
args = ap.parse_args()
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES[args.shape]
dt = {"f32": torch.float32, "i32": torch.int32}[args.dtype]
code = {torch.float32: _lib.FLT32, torch.int32: _lib.INT32}[dt]
x = synth.features(n, args.h, dt, seed=0, device=dev)
out = torch.empty((n, args.h), dtype=dt, device=dev)
_lib.set_tunable("lds_mode", args.mode)
print(f"# {args.shape}-shaped (N = {n}, nnz = {nnz}), {args.dtype} h = {args.h}, lds_mode = {args.mode}; ids of rmat / sbm graphs shuffled, sbm-sorted = the same SBM with its communities in id order")
print(f"# {'graph':12s} {'tiles':>12s} {'ms':>8s} {'median':>8s} {'kernel':>30s} {'chunk fills':>12s} {'of all':>7s} {'shared reads':>13s} {'labels':>7s} {'largest':>8s} {'create ms':>10s}  check")
for kind in args.kinds.split(","):
    if kind == "sbm-sorted":
        rowptr, col = synth.make_shape(args.shape, seed=0, device=dev, kind="sbm", shuffle=False)
    else:
        rowptr, col = synth.make_shape(args.shape, seed=0, device=dev, kind=None if kind == "uniform" else kind)
    want = torch.bincount(col.long(), minlength=n).double() @ x.double()
    for order, loc in [(int(v), int(w)) for v in args.orders.split(",") for w in (args.locality.split(",") if args.locality else ["1"])]:
        _lib.set_tunable("lds_tile_order", order)
        _lib.set_tunable("panel_locality", loc)
        torch.cuda.synchronize()
        t0 = time.time()
        hd = _lib.group_create(_lib.CSR, code, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [args.h], args.h)
        torch.cuda.synchronize()
        t_create = (time.time() - t0) * 1e3
        out.zero_()
        for _ in range(2):
            _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
        torch.cuda.synchronize()
        ts = []
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(args.reps):
            a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
        ok = "OK" if torch.equal(out.double().sum(0), want) else "MISMATCH"
        lp, lc, geo, lt = _lib.group_lds_plan(hd), _lib.group_lds_code(hd), _lib.group_lds_geometry(hd), _lib.group_lds_tiles(hd)
        kern = "code stream" if lc["active"] else ("token kernels" if lp["tiles"] else ("L2 sweep" + ("" if not lt["sweep_locality"] else (", items by id block" if lt["sweep_locality"] == 1 else ", items by label"))))
        nchunks = (n + max(geo["chunk_cols"], 1) - 1) // max(geo["chunk_cols"], 1)
        fills = lp["chunk_fills"]
        print(f"  {kind:12s} {'similarity' if lt['similarity'] else 'consecutive':>12s} {min(ts):8.3f} {sorted(ts)[len(ts) // 2]:8.3f} {kern:>30s} {fills:12d} "
              f"{(fills / max(lp['tiles'] * nchunks, 1)):7.3f} {(geo['shared_entries'] / max(lp['nnz'], 1)):13.3f} {lt['labels']:7d} {lt['largest_label_rows']:8d} {t_create:10.1f}  {ok}",
              flush=True)
        _lib.group_free(hd)
    del rowptr, col
