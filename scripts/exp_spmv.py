#!/usr/bin/env python3
"""SpMV end of the path (rows of X of 1 / 2 / 4 elements): LDS-staged vector kernel (round 2) vs the CSR-vector kernel
(lanes over a row's entries, gathers through the caches) vs the sweep."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
st = torch.cuda.current_stream().cuda_stream
for shape in ("reddit", "ogbn-products"):
    n, nnz, dmax = synth.SHAPES[shape]
    rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
    for w in (1, 2, 4):
        x = synth.features(n, w, torch.int32, seed=0, device=dev)
        out = torch.empty((n, w), dtype=torch.int32, device=dev)
        res = []
        for vk, vl in ((1, 1), (1, 0), (0, 0)):
            _lib.set_tunable("vec_kernel", vk)
            _lib.set_tunable("vec_lds", vl)
            hd = _lib.group_create(_lib.CSR, _lib.INT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [w], w)
            for _ in range(3): _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
            for _ in range(6):
                a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
            res.append((min(ts), out.double().sum().item()))
            _lib.group_free(hd)
        print(f"{shape:14s} w={w}: lds-vector {res[0][0]:7.3f} ms   vector {res[1][0]:7.3f} ms   sweep {res[2][0]:7.3f} ms   "
              f"same {res[0][1] == res[1][1] == res[2][1]}", flush=True)
    _lib.set_tunable("vec_kernel", 1)
    _lib.set_tunable("vec_lds", 1)
    del rowptr, col
