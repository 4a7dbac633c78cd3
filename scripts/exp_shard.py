#!/usr/bin/env python3
"""What ONE rank of an 8-GPU split computes, timed on one GPU: a 1/8 row share at h = 256, all rows at h = 64 / 32
(feature split), the grids between (1/2 rows x 64, 1/4 rows x 128) -- Reddit-shaped, FLT32 by default.  For each: the whole product
(slice pack + kernel + reduce), the library's own kernel time, the plan it got, and the error against the oracle's stored-order loop on
sampled rows (relative to |A|.|x|: the 1e-5 contract) -- or bit-identity where the plan keeps it.
usage: exp_shard.py [name=value,...tunables] [dtype] [cases: r8,h64,h32,g24,g42,full]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import oracle
from pygim_amd import _lib, synth
from pygim_amd.bench_plans import nnz_balanced_row_split

dev = torch.device("cuda", 0)
_lib.init_ranks(1)
tun = sys.argv[1] if len(sys.argv) > 1 else ""
for kv in filter(None, tun.split(",")):
    k, v = kv.split("=")
    assert _lib.set_tunable(k, int(v)) != -1, k
dtn = sys.argv[2] if len(sys.argv) > 2 else "FLT32"
tdt, code = {"FLT32": (torch.float32, _lib.FLT32), "INT32": (torch.int32, _lib.INT32), "INT8": (torch.int8, _lib.INT8), "INT16": (torch.int16, _lib.INT16),
             "DBL64": (torch.float64, _lib.DBL64), "INT64": (torch.int64, _lib.INT64)}[dtn]
want = (sys.argv[3] if len(sys.argv) > 3 else "r8,h64,h32,g24,g42").split(",")
QUANT = len(sys.argv) > 4 and sys.argv[4] == "quant"   # time the conv layers' quantise -> aggregate -> dequantise (pygim_quant_spmm_run) instead of the plain product
print("#", tun or "defaults", dtn, flush=True)
n, nnz, dmax = synth.SHAPES["reddit"]
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
rp_cpu = rowptr.cpu()
CASES = {"full": (1, 256), "r8": (8, 256), "h64": (1, 64), "h32": (1, 32), "g24": (2, 64), "g42": (4, 128), "r2": (2, 256), "r4": (4, 256), "h128": (1, 128), "h16": (1, 16), "h8": (1, 8), "h24": (1, 24), "h12": (1, 12)}


def timed(hd, x, out, reps=9):
    for _ in range(3):
        _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


for name in want:
    frac, h = CASES[name]
    top = nnz_balanced_row_split(rp_cpu, frac)[1]
    m = int(rp_cpu[top])
    if tdt in (torch.float32, torch.float64):
        x = synth.features(n, h, tdt, seed=1, device=dev, kind="uniform")
    else:
        x = synth.features(n, h, tdt, seed=1, device=dev)
    out = torch.empty((top, h), dtype=tdt, device=dev)
    hd = _lib.group_create(_lib.CSR, code, [rowptr.data_ptr()], [col.data_ptr()], None, [top], [n], [m], [1], [h], h)
    if QUANT:
        xf = torch.randn((n, h), device=dev)
        of = torch.empty((top, h), dtype=torch.float32, device=dev)
        sc = torch.empty((), dtype=torch.float32, device=dev)
        runs0 = _lib.group_lds_runs(hd)
        for _ in range(3):
            _lib.quant_spmm_run(hd, xf.data_ptr(), h, of.data_ptr(), sc.data_ptr(), 0)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(9):
            a.record(); _lib.quant_spmm_run(hd, xf.data_ptr(), h, of.data_ptr(), sc.data_ptr(), 0); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        print(f"{name:5s} rows 1/{frac} ({top}), h={h}: quantise -> aggregate -> dequantise {ts[4]:6.3f} ms  on the LDS-staged kernel: {_lib.group_lds_runs(hd) > runs0}  [{_lib.group_lds_note(hd)[:90]}]", flush=True)
        _lib.group_free(hd)
        continue
    t = timed(hd, x, out)
    _lib.group_kernel_events(hd, True)
    _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
    torch.cuda.synchronize()
    kms = _lib.group_kernel_ms(hd)[0]
    geo, plan, note = _lib.group_lds_geometry(hd), _lib.group_lds_plan(hd), _lib.group_lds_note(hd)
    runs = _lib.group_lds_runs(hd)
    _lib.group_free(hd)
    # sampled rows against the stored-order loop
    rows = np.unique(np.concatenate([np.arange(0, min(top, 64)), np.linspace(0, top - 1, 192).astype(np.int64)]))
    rp = rp_cpu.numpy().astype(np.int64)
    xh = x.cpu().numpy()
    worst, ident = 0.0, True
    for r in rows:
        lo, hi = rp[r], rp[r + 1]
        cc = col[lo:hi].cpu().numpy()
        ref = oracle.spmm_csr(np.array([0, hi - lo], np.int32), cc, None, xh)[0]
        got = out[r].cpu().numpy()
        if tdt in (torch.float32, torch.float64):
            bound = np.abs(xh[cc]).sum(0) + 1e-30
            worst = max(worst, float(np.max(np.abs(got.astype(np.float64) - ref.astype(np.float64)) / bound)))
            ident = ident and got.tobytes() == ref.tobytes()
        else:
            ident = ident and np.array_equal(got, ref)
    print(f"{name:5s} rows 1/{frac} ({top}), h={h}: {t:6.3f} ms  kernel(s) {kms:6.3f} ms  lds_runs {runs}  tiles {plan['tiles']} splits {geo['col_splits']} waves {geo['waves']}x{geo['acc_per_wave']} "
          f"kc {geo['chunk_cols']}x{geo['buffers']}  {'bit-identical' if ident else 'max err %.2e of |A||x|' % worst}  [{note[:60]}]", flush=True)
_lib.release()
