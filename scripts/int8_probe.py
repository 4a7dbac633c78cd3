#!/usr/bin/env python3
"""A few fused INT8 quantise -> aggregate -> dequantise calls (Reddit-shaped, h = 256) for a profiler."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES["reddit"]
h = 256
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(n, h, device=dev)
out = torch.empty((n, h), dtype=torch.float32, device=dev)
scale = torch.empty((), dtype=torch.float32, device=dev)
hd = _lib.group_create(_lib.CSR, _lib.INT8, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
for _ in range(2): _lib.quant_spmm_run(hd, x.data_ptr(), h, out.data_ptr(), scale.data_ptr(), st, 0, 0, False)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
for _ in range(4):
    a.record(); _lib.quant_spmm_run(hd, x.data_ptr(), h, out.data_ptr(), scale.data_ptr(), st, 0, 0, False); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
print(f"fused INT8 aggregation: {min(ts):.3f} ms", flush=True)
