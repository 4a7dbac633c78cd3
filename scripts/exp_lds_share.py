#!/usr/bin/env python3
"""Row shares of the Reddit-shaped graph (what one rank of a multi-GPU row split computes): LDS-staged kernel forced against the
sweep, to place the reuse rule (stored entries per staged column of X).  Also feature windows (h = 32 / 64 / 128)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth
from pygim_amd.bench_plans import nnz_balanced_row_split

dev = torch.device("cuda", 0)
_lib.init_ranks(1)
for kv in filter(None, (sys.argv[1] if len(sys.argv) > 1 else "").split(",")):   # name=value,... tunables for every plan
    k, v = kv.split("=")
    _lib.set_tunable(k, int(v))
print("#", sys.argv[1:] or "defaults", flush=True)
n, nnz, dmax = synth.SHAPES["reddit"]
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
rp_cpu = rowptr.cpu()


def timed(hd, x, out):
    for _ in range(2):
        _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts)


for frac, h in ((1, 256), (2, 256), (3, 256), (4, 256), (6, 256), (8, 256), (16, 256), (8, 128), (4, 128), (2, 128), (1, 128), (1, 64), (2, 64), (4, 64)):
    top = nnz_balanced_row_split(rp_cpu, frac)[1]
    m = int(rp_cpu[top])
    x = synth.features(n, h, torch.float32, seed=0, device=dev)
    out = torch.empty((top, h), dtype=torch.float32, device=dev)
    res = {}
    for mode in (2, 1):
        _lib.set_tunable("lds_mode", mode)
        hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [top], [n], [m], [1], [h], h)
        res[mode] = (timed(hd, x, out), _lib.group_lds_plan(hd), _lib.group_lds_geometry(hd))
        _lib.group_free(hd)
    lp = res[1][1]
    reuse = m / max(lp["tiles"] * n, 1)
    print(f"rows 1/{frac} ({top}), h={h}: sweep {res[2][0]:6.3f} ms   lds {res[1][0]:6.3f} ms   tiles {lp['tiles']:4d}  entries per staged column {reuse:5.2f}  {res[1][2]}", flush=True)
