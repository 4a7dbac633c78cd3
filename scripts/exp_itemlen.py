#!/usr/bin/env python3
"""How much of the sweep is per-item overhead?  The benchmark's X (232 965 columns, h = 256 f32) and nnz, with the rows made
fewer-and-longer or more-and-shorter: items of about 30 / 61 / 123 / 246 entries per (row, panel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
ncols, nnz, dmax = synth.SHAPES["reddit"]
h = 256
st = torch.cuda.current_stream().cuda_stream
x = synth.features(ncols, h, torch.float32, seed=0, device=dev)
for scale in (2.0, 1.0, 0.5, 0.25):
    nrows = int(ncols * scale)
    rowptr, col = synth.make_csr(nrows, nnz, int(dmax / scale), seed=0, device=dev, ncols=ncols)
    out = torch.empty((nrows, h), dtype=torch.float32, device=dev)
    hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [nrows], [ncols], [nnz], [1], [h], h)
    for _ in range(2): _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(5):
        a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    plan = _lib.group_plan(hd)
    print(f"rows {nrows:7d} (x{scale}): {min(ts):6.3f} ms  entries per item {nnz / max(plan['n_items'], 1):6.1f}  panels {plan['n_panels']}  gather {nnz * h * 4 / min(ts) / 1e9:6.2f} TB/s", flush=True)
    _lib.group_free(hd)
    del rowptr, col, out
