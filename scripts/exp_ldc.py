#!/usr/bin/env python3
"""Does a C row stride that is not a multiple of 128 B slow the panel sweep?  block_run with ldc = h vs padded ldc."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth

dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES["reddit"]
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
for h, ldx, ldc in ((100, 100, 100), (100, 100, 128), (100, 128, 128), (100, 128, 100), (128, 128, 128)):
    x = torch.zeros((n, ldx), dtype=torch.int32, device=dev)
    x[:, :h] = synth.features(n, h, torch.int32, seed=0, device=dev)
    out = torch.zeros((n, ldc), dtype=torch.int32, device=dev)
    hd = _lib.group_create(_lib.CSR, _lib.INT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: _lib.block_run(hd, 0, x.data_ptr(), ldx, out.data_ptr(), ldc, h, False, st)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        a.record(); run(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    print(f"h={h} ldx={ldx} ldc={ldc}: {min(ts):.3f} ms", flush=True)
    _lib.group_free(hd)

# one 32-column window of a 256-wide X (feature split): strided source rows, packed by k_slice_pack
x = synth.features(n, 256, torch.float32, seed=0, device=dev)
out = torch.zeros((n, 32), dtype=torch.float32, device=dev)
hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [32], 32)
st = torch.cuda.current_stream().cuda_stream
run = lambda: _lib.block_run(hd, 0, x.data_ptr() + 4 * 64, 256, out.data_ptr(), 32, 32, False, st)
for _ in range(2):
    run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    a.record(); run(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
ok = torch.equal(out.double().sum(0), torch.bincount(col.long(), minlength=n).double() @ x[:, 64:96].double())
print(f"window 32 of 256 (ldx=256): {min(ts):.3f} ms checksum {ok}", flush=True)
