#!/usr/bin/env python3
"""One rank's share of the N-GPU row split (1/frac of the rows, all columns): panel budget sweep."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth
from bench import nnz_balanced_row_split
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES["reddit"]
h = 256
frac = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
x = synth.features(n, h, torch.float32, seed=0, device=dev)
top = nnz_balanced_row_split(rowptr.cpu(), frac)[1]
hi = int(rowptr[top])
rp, cc = rowptr[: top + 1].contiguous(), col[:hi].contiguous()
out = torch.empty((top, h), dtype=torch.float32, device=dev)
for mb, bt in ((4, 256), (4, 128), (4, 64)):
    _lib.set_tunable("panel_bytes", mb << 20)
    _lib.set_tunable("panel_block", bt)
    hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rp.data_ptr()], [cc.data_ptr()], None, [top], [n], [hi], [1], [h], h)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(8):
        a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    print(f"rows 1/{frac}, block {bt}, budget {mb} MiB: {min(ts):.3f} ms  panels {_lib.group_info(hd)['n_panels']}", flush=True)
    _lib.group_free(hd)
