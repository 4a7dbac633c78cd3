#!/usr/bin/env python3
"""Reddit-shaped CSR h=256 with real-valued weights (a normalised adjacency) against unit weights."""
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
for dt, code in ((torch.float32, _lib.FLT32), (torch.float64, _lib.DBL64), (torch.int32, _lib.INT32), (torch.int64, _lib.INT64), (torch.int16, _lib.INT16), (torch.int8, _lib.INT8)):
    x = synth.features(n, h, dt, seed=0, device=dev)
    out = torch.empty((n, h), dtype=dt, device=dev)
    for weighted, narrow in ((False, 1), (True, 1)) + (((True, 3), (True, 4)) if dt in (torch.int16, torch.int8) else ()) + (((True, 0),) if dt in (torch.float64, torch.int64) else ()) + (((True, 2),) if dt in (torch.int32, torch.int64) else ()):
        _lib.set_tunable("narrow_vals", min(narrow, 1))   # (8-byte types: values that fit 4 bytes exactly are streamed as such)
        vals = None
        if weighted:
            vals = (torch.rand(nnz, device=dev) * 2 - 1).to(dt) if dt.is_floating_point else torch.randint(-3, 4, (nnz,), device=dev, dtype=dt)
            if narrow >= 3:   # INT16 / INT8: values over the whole range of the type (through an SGPR); 4: without the code stream (token kernel / sweep)
                ii = torch.iinfo(dt)
                vals = torch.randint(ii.min, ii.max + 1, (nnz,), device=dev, dtype=torch.int32).to(dt)
                _lib.set_tunable("lds_code", 0 if narrow == 4 else 1)
            if narrow == 2:   # values of any size (no inline constants: the value travels through an SGPR; INT64: both halves through a pair)
                vals = torch.randint(-2**31, 2**31 - 1, (nnz,), device=dev, dtype=torch.int64).to(torch.int32) if dt == torch.int32 else \
                    torch.randint(-2**62, 2**62, (nnz,), device=dev, dtype=torch.int64)
        hd = _lib.group_create(_lib.CSR, code, [rowptr.data_ptr()], [col.data_ptr()], None if vals is None else [vals.data_ptr()], [n], [n], [nnz], [1], [h], h)
        for _ in range(2): _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
        for _ in range(5):
            a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
        print(f"{str(dt):14s} weighted={weighted}{'' if narrow == 1 else (' (narrow_vals=0)' if narrow == 0 else ' (values of any size)' if narrow < 4 else ' (values of any size, lds_code = 0)')}: {min(ts):7.3f} ms", flush=True)
        _lib.set_tunable("lds_code", 1)
        _lib.group_free(hd)
