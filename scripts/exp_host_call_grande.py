#!/usr/bin/env python3
"""grande's call with CPU tensors (backend_pim/grande.py mul: eight per-unit windows of 32 features, contiguous copies) at the Reddit shape, FLT32 h = 256, through
the Python surface: serial (host_windows = 1) against the pipeline (the narrow windows gathered into two windows of 128 features on the device)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from pygim_amd import _lib, pim_ops, synth
from pygim_amd.backend_pim import grande
from pygim_amd.sparse_tensor import SparseTensorShim

dev = torch.device("cuda", 0)
pim_ops.load("grande")
units = torch.ops.pim_ops.dpu_init_ranks(1)
n, nnz, dmax = synth.SHAPES["reddit"]
h = 256
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
A = grande.SparseTensorCOO(SparseTensorShim(rowptr=rowptr, col=col, sparse_sizes=(n, n)), dtype=torch.float32, dpus_per_rank=units, format="CSR")
A.to_pim_group_csr(h)
x = synth.features(n, h, torch.float32, seed=1, kind="uniform")
want = A.mul(x.to(dev)).cpu()
print(f"# grande, Reddit-shaped CSR FLT32 h = {h}, {units} units per rank; plan: {_lib.group_lds_note(A.sp_info_ptr)}", flush=True)
for hw, direct in ((1, 0), (0, 1), (0, 0), (0, 2)):
    _lib.set_tunable("host_windows", hw)
    _lib.set_tunable("host_direct", direct)
    ts = []
    for it in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = A.mul(x)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    call = _lib.group_host_call(A.sp_info_ptr)
    t = _lib.group_timers(A.sp_info_ptr)
    print(f"host_windows={hw} host_direct={direct}: ran {call}  per mul (wrapper's dense_split included): " + " ".join(f"{v:.2f}" for v in ts) +
          f" ms   library: up {t[0]:.2f} + {t[1]:.2f} + {t[2]:.2f}   {'equal to the device-resident product' if torch.equal(out, want) else 'MISMATCH'}", flush=True)
_lib.set_tunable("host_windows", 0)
_lib.set_tunable("host_direct", 1)
A.free_group()
