#!/usr/bin/env python3
"""slice grouping on a graph whose single slice (3 GB) is far beyond the Infinity Cache: on / off"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24_000_000
nnz = int(n * 14.55)
rowptr, col = synth.make_csr(n, nnz, 100_000, seed=0, device=dev)
st = torch.cuda.current_stream().cuda_stream
for h in (64, 128):
    x = synth.features(n, h, torch.float32, seed=0, device=dev)
    out = torch.empty((n, h), dtype=torch.float32, device=dev)
    for sg in (640 << 20, 0, 8 << 30):
        _lib.set_tunable("slice_group_bytes", sg)
        hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
        for _ in range(2): _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
        for _ in range(4):
            a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
        t = min(ts)
        print(f"row bytes {4*h:4d} slice_group_bytes {sg>>20:6d} MiB: {t:8.3f} ms  {nnz*4*h/t/1e9:7.2f} TB/s of rows", flush=True)
        _lib.group_free(hd)
    del x, out
