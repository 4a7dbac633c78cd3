#!/usr/bin/env python3
"""products-shaped graph (avg degree 50, X = 2.5 GB): row-per-wave gathers vs forced single-/multi-panel sweeps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES["ogbn-products"]
h = 256
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
x = synth.features(n, h, torch.float32, seed=0, device=dev)
out = torch.empty((n, h), dtype=torch.float32, device=dev)
chk = None
# (round 3) Infinity-Cache-sized blocks: a column panel of ONE / TWO slices held to ~200 MB
for name, mode, mb, sg in (("row-per-wave", 2, 4, 640), ("sweep, 1 panel", 1, 1 << 20, 640), ("sweep, 64 MiB panels", 1, 64, 640),
                           ("sweep, 16 MiB panels", 1, 16, 640), ("sweep, 100 MiB panels x 2 slices (200 MB blocks)", 1, 100, 200),
                           ("sweep, 200 MiB panels x 1 slice (200 MB blocks)", 1, 200, 200), ("sweep, 1 panel, 1 slice at a time", 1, 1 << 20, 320)):
    _lib.set_tunable("panel_mode", mode); _lib.set_tunable("panel_bytes", int(mb) << 20); _lib.set_tunable("slice_group_bytes", int(sg) << 20)
    hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2): _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(4):
        a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    s = out.double().sum().item()
    chk = s if chk is None else chk
    print(f"{name:24s} {min(ts):8.3f} ms panels {_lib.group_info(hd)['n_panels']} same_sum {s == chk}", flush=True)
    _lib.group_free(hd)
