#!/usr/bin/env python3
"""Experiment: the output matrix C (re-read and re-written by each of the 8 panel launches) in UNCACHED device memory
(hipExtMallocWithFlags(hipDeviceMallocUncached)), so that its lines do not pass through the L2 that holds the X panel."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES["reddit"]
h = 256
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
x = synth.features(n, h, torch.float32, seed=0, device=dev)
hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
st = torch.cuda.current_stream().cuda_stream
out_t = torch.empty((n, h), dtype=torch.float32, device=dev)
p = ctypes.c_void_p()
rc = hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(n * h * 4), ctypes.c_uint(3))  # hipDeviceMallocUncached = 0x3
print("hipExtMallocWithFlags rc", rc)
def bench(ptr, label):
    for _ in range(3): _lib.spmm_run_group(hd, [x.data_ptr()], ptr, st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(10):
        a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], ptr, st); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); print(f"{label}: median {ts[5]:.3f} ms best {ts[0]:.3f} ms", flush=True)
bench(out_t.data_ptr(), "C in ordinary device memory")
if rc == 0:
    bench(p.value, "C in uncached device memory")
    bench(out_t.data_ptr(), "C in ordinary device memory (again)")
