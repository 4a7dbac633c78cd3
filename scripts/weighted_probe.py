#!/usr/bin/env python3
"""One weighted (or unit) Reddit-shaped f32 product per run for a profiler.  usage: weighted_probe.py [0|1] [panel_bytes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth
weighted = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
if len(sys.argv) > 2:
    _lib.set_tunable("panel_bytes", int(sys.argv[2]))
n, nnz, dmax = synth.SHAPES["reddit"]
h = 256
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
st = torch.cuda.current_stream().cuda_stream
x = synth.features(n, h, torch.float32, seed=0, device=dev)
out = torch.empty((n, h), dtype=torch.float32, device=dev)
vals = (torch.rand(nnz, device=dev) * 2 - 1) if weighted else None
hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None if vals is None else [vals.data_ptr()], [n], [n], [nnz], [1], [h], h)
for _ in range(2): _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
for _ in range(4):
    a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
print(f"weighted={weighted} panel_bytes={sys.argv[2] if len(sys.argv) > 2 else 'default'}: {min(ts):.3f} ms  plan {_lib.group_plan(hd)}", flush=True)
