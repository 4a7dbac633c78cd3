#!/usr/bin/env python3
"""Round 4: the code-stream product under its geometry knobs, one graph, many plans.
exp_code_geo.py [--clustered] [--h H] [--dtype f32|i32|i16] [--reps R] cfg [cfg ...]
cfg = waves:nbuf:kc:gsize:nsets  (0 = the library's default for that knob), e.g. 16:2:0:0:0 8:4:160:10:2"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--clustered", action="store_true")
ap.add_argument("--h", type=int, default=256)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--shape", default="reddit")
ap.add_argument("--tune", default="", help="name=value,... extra tunables for every configuration")
ap.add_argument("cfgs", nargs="+")
args = ap.parse_args()
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES[args.shape]
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev, clustered=args.clustered)
dt = {"f32": torch.float32, "i32": torch.int32, "i16": torch.int16, "i8": torch.int8, "f64": torch.float64, "i64": torch.int64}[args.dtype]
code = {torch.float32: _lib.FLT32, torch.int32: _lib.INT32, torch.int16: _lib.INT16, torch.int8: _lib.INT8, torch.float64: _lib.DBL64, torch.int64: _lib.INT64}[dt]
modulus = {torch.int8: 256, torch.int16: 65536}.get(dt)
x = synth.features(n, args.h, dt, seed=0, device=dev)
colcount = torch.bincount(col.long(), minlength=n).double()
want = colcount @ x.double()
out = torch.empty((n, args.h), dtype=dt, device=dev)
_lib.set_tunable("lds_mode", 1)
for kv in filter(None, args.tune.split(",")):
    k, v = kv.split("=")
    _lib.set_tunable(k, int(v))
print(f"# {args.shape}{' clustered' if args.clustered else ''} {args.dtype} h={args.h} {args.tune}", flush=True)
for cfg in args.cfgs:
    w, nbuf, kc, gs, ns = (int(v) for v in cfg.split(":"))
    for k, v in (("lds_code_waves", w), ("lds_code_nbuf", nbuf), ("lds_code_kc", kc), ("lds_code_gsize", gs), ("lds_code_nsets", ns)):
        _lib.set_tunable(k, v)
    t0 = time.time()
    hd = _lib.group_create(_lib.CSR, code, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [args.h], args.h)
    t_create = time.time() - t0
    out.zero_()
    for _ in range(2):
        _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
    torch.cuda.synchronize()
    ts = []
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(args.reps):
        a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    if modulus:   # narrow integers wrap: compare the column sums modulo the element width
        ok = "OK" if bool((((out.long().sum(0) - want.long()) % modulus) == 0).all()) else "MISMATCH"
    else:
        ok = "OK" if torch.equal(out.double().sum(0), want) else "MISMATCH"
    lp, lc = _lib.group_lds_plan(hd), _lib.group_lds_code(hd)
    print(f"cfg {cfg:>16}: {min(ts):7.3f} ms (median {sorted(ts)[len(ts)//2]:7.3f})  checksum {ok}  create {t_create:5.2f} s  plan={lp} code={lc}", flush=True)
    _lib.group_free(hd)
