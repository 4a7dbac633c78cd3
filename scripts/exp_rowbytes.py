#!/usr/bin/env python3
"""papers100M-like degree (14.5) and column spread on a graph far beyond every cache: how does the product time depend
on the gathered row's BYTES (32 ... 512)?  If 64-byte rows cost what 128-byte rows cost, the memory side moves whole
128-byte lines and a feature split below 128 bytes per rank wastes half of every fetch."""
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
for h in (8, 16, 32, 64, 128):
    x = synth.features(n, h, torch.float32, seed=0, device=dev)
    out = torch.empty((n, h), dtype=torch.float32, device=dev)
    hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
    for _ in range(2): _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(4):
        a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    t = min(ts)
    print(f"n {n} nnz {nnz} row bytes {4*h:4d}: {t:8.3f} ms  {nnz/t/1e6:7.2f} G gathers/s  {nnz*4*h/t/1e9:7.2f} TB/s of rows  panels {_lib.group_info(hd)['n_panels']}", flush=True)
    _lib.group_free(hd)
    del x, out
