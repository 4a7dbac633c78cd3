#!/usr/bin/env python3
"""Row shares of the Reddit-shaped graph (a rank's share of an N-GPU row split) with the column-split LDS plan: every row tile split
into S column ranges (S x the workgroups, 1/S of X each; partial sums reduced in range order) against whole-X tiles and the sweep."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth
from pygim_amd.bench_plans import nnz_balanced_row_split

dev = torch.device("cuda", 0)
_lib.init_ranks(1)
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


for frac, h in ((2, 256), (4, 256), (8, 256), (16, 256), (8, 128), (8, 64)):
    top = nnz_balanced_row_split(rp_cpu, frac)[1]
    m = int(rp_cpu[top])
    x = synth.features(n, h, torch.float32, seed=0, device=dev)
    out = torch.empty((top, h), dtype=torch.float32, device=dev)
    ref = None
    line = f"rows 1/{frac} ({top}), h={h}:"
    for name, knobs in (("sweep", {"lds_mode": 2}), ("LDS whole-X tiles", {"lds_mode": 1, "lds_col_split": 1}),
                        ("LDS column-split", {"lds_mode": 1, "lds_col_split": 0, "lds_col_split_f32": 1})):
        old = {k: _lib.set_tunable(k, v) for k, v in knobs.items()}
        hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [top], [n], [m], [1], [h], h)
        t = timed(hd, x, out)
        tiles = _lib.group_lds_plan(hd)["tiles"]
        _lib.group_free(hd)
        for k, v in old.items():
            _lib.set_tunable(k, v)
        ok = ""
        if ref is None:
            ref = out.clone()
        else:
            ok = " =" if torch.equal(out, ref) else " DIFFERENT"   # (integer-valued features: every order is exact)
        line += f"   {name} {t:6.3f} ms ({tiles} tiles){ok}"
    print(line, flush=True)
