#!/usr/bin/env python3
"""products-shaped (X = 2.5 GB): one product of 1 KiB rows (default) against 8 / 16 / 32 separate window products of
128 / 64 / 32-byte rows (contiguous [N, w] windows, fuse_windows = 0): does a working set that fits the Infinity Cache
(157 MB at 64 bytes per row) pay for the narrower gathers?"""
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
st = torch.cuda.current_stream().cuda_stream
chk = None
for parts in (1, 8, 16, 32):
    _lib.set_tunable("fuse_windows", 0)
    w = h // parts
    chunks = [x[:, j * w:(j + 1) * w].contiguous() for j in range(parts)]
    hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [parts], [w] * parts, h)
    run = lambda: _lib.spmm_run_group(hd, [c.data_ptr() for c in chunks], out.data_ptr(), st)
    for _ in range(2): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(3):
        a.record(); run(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    s = out.double().sum().item(); chk = s if chk is None else chk
    print(f"{parts:3d} windows of {4*w:4d}-byte rows: {min(ts):8.3f} ms  same_sum {s == chk}", flush=True)
    _lib.group_free(hd)
