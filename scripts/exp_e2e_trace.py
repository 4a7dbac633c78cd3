#!/usr/bin/env python3
"""Which engine moves the windows?  Six groups, three pipelined host calls each (C ABI, pageable X, page-locked C); run under
rocprofv3 --kernel-trace --memory-copy-trace: a fast group's copies are all on the DMA engines, a slow one's ...?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from pygim_amd import _lib, synth

dev = torch.device("cuda", 0)
_lib.init_ranks(1)
for kv in filter(None, (sys.argv[1] if len(sys.argv) > 1 else "").split(",")):
    k, v = kv.split("=")
    assert _lib.set_tunable(k, int(v)) != -1, k
n, nnz, dmax = synth.SHAPES["reddit"]
h = 256
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
x = synth.features(n, h, torch.float32, seed=1)
out = torch.empty((n, h), dtype=torch.float32, pin_memory=True)
junk = []
for gi in range(6):
    hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
    ts = []
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
        ts.append((time.perf_counter() - t0) * 1e3)
    t = _lib.group_timers(hd)
    print(f"group {gi}: " + " ".join(f"{v:.2f}" for v in ts) + f" ms   last: up {t[0]:.2f} + {t[1]:.2f} + {t[2]:.2f}", flush=True)
    _lib.group_free(hd)
    junk.append(torch.cuda.Stream())   # (the process creates streams between groups, as bench.py's legs do)
