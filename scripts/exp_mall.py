#!/usr/bin/env python3
"""X far beyond the 256 MiB Infinity Cache (products-shaped: 2.5 GB): all feature slices in one launch (each XCD
its own slice -> working set = all of X) against a few slices per launch (tunable slice_group_bytes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
shape = sys.argv[1] if len(sys.argv) > 1 else "ogbn-products"
n, nnz, dmax = synth.SHAPES[shape]
h = 256
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=4):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(reps):
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)


for dt, code in ((torch.float32, _lib.FLT32), (torch.int32, _lib.INT32), (torch.int8, _lib.INT8), (torch.float64, _lib.DBL64)):
    x = synth.features(n, h, dt, seed=0, device=dev)
    out = torch.empty((n, h), dtype=dt, device=dev)
    chk = None
    for mb in (0, 160, 320, 640, 1280):
        _lib.set_tunable("slice_group_bytes", mb << 20)
        hd = _lib.group_create(_lib.CSR, code, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
        out.zero_()
        t = timed(lambda: _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st))
        s = out.double().sum().item()
        chk = s if chk is None else chk
        print(f"{shape} {str(dt):14s} slice_group {mb:5d} MiB {t:8.3f} ms same_sum {s == chk}", flush=True)
        _lib.group_free(hd)
    del x, out
