#!/usr/bin/env python3
"""The benchmark sweep with fewer resident waves per CU (unused dynamic LDS per block limits the blocks a CU holds):
is it short of waves (latency-bound) or not?"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pygim_amd import _lib, synth
dev = torch.device("cuda", 0); _lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES["reddit"]; h = 256
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
x = synth.features(n, h, torch.float32, seed=0, device=dev); out = torch.empty((n, h), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
for pad, blocks in ((0, 8), (20 * 1024, 7), (23 * 1024, 6), (27 * 1024, 5), (33 * 1024, 4), (41 * 1024, 3), (55 * 1024, 2)):
    _lib.set_tunable("panel_lds_pad", pad)
    for _ in range(2): _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(5):
        a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    print(f"LDS pad {pad:6d} B  (<= {blocks} blocks = {4 * blocks} waves per CU): {min(ts):.3f} ms", flush=True)
