#!/usr/bin/env python3
"""SpMV end of the path on the Reddit-shaped graph (rows of X of w elements, w <= 4): a few products for a profiler, and the
event-timed product time.  usage: spmv_probe.py [w] [dtype i32|f32] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth
w = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dt = sys.argv[2] if len(sys.argv) > 2 else "i32"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
shape = sys.argv[4] if len(sys.argv) > 4 else "reddit"
tdt, code = {"i32": (torch.int32, _lib.INT32), "f32": (torch.float32, _lib.FLT32)}[dt]
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
st = torch.cuda.current_stream().cuda_stream
if os.environ.get('PYGIM_VEC_MIN_SEG'):
    _lib.set_tunable('vec_lds_min_seg', int(os.environ['PYGIM_VEC_MIN_SEG']))
if os.environ.get('PYGIM_VEC_LDS'):
    _lib.set_tunable('vec_lds', int(os.environ['PYGIM_VEC_LDS']))
n, nnz, dmax = synth.SHAPES[shape]
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
x = synth.features(n, w, tdt, seed=0, device=dev)
out = torch.empty((n, w), dtype=tdt, device=dev)
hd = _lib.group_create(_lib.CSR, code, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [w], w)
for _ in range(3):
    _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(iters):
    a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
wsum = torch.bincount(col.long(), minlength=n).double()
ok = bool(torch.equal(out.double().sum(0), wsum @ x.double()))
print(f"{shape} w={w} {dt}: best {min(ts):.4f} ms  median {sorted(ts)[len(ts)//2]:.4f} ms  plan {_lib.group_plan(hd)}  checksum_ok {ok}", flush=True)
_lib.group_free(hd)
