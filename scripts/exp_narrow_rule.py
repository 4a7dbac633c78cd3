#!/usr/bin/env python3
"""Where does the LDS-staged kernel stop paying for NARROW products (at most 32 lanes)?  Reddit-sized graphs with 1 / 1, 1 / 2, 1 / 3, 1 / 5 of the entries, 16 FLT32
features: the LDS-staged kernel forced (lds_mode = 1: half-split plan), the sweep forced (lds_mode = 2), and what the rule picks (lds_min_reuse_narrow_x100)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from pygim_amd import _lib, synth

dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES["reddit"]


def timed(hd, x, out, reps=9):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st)
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


for h in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("16", "32"))]:
    for div in [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("1", "2", "3", "5"))]:
        m = nnz // div
        rowptr, col = synth.make_csr(n, m, max(dmax // div, 64), seed=3, device=dev)
        x = synth.features(n, h, torch.float32, seed=1, device=dev)
        out = torch.empty((n, h), dtype=torch.float32, device=dev)
        res = {}
        for name, mode in (("rule", 0), ("LDS forced", 1), ("sweep forced", 2)):
            old = _lib.set_tunable("lds_mode", mode)
            try:
                hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [m], [1], [h], h)
            finally:
                _lib.set_tunable("lds_mode", old)
            _lib.set_tunable("lds_mode", mode)
            t = timed(hd, x, out)
            _lib.set_tunable("lds_mode", old)
            res[name] = (t, _lib.group_lds_plan(hd)["tiles"] > 0)
            _lib.group_free(hd)
        tiles = -(-n // (8 * 228))
        print(f"h = {h:2d}  entries 1/{div} ({m / 1e6:6.1f} M): {m / (tiles * n):5.2f} entries per staged column ({2 * m / (tiles * n):5.2f} per column of a half-split plan)   "
              f"LDS {res['LDS forced'][0]:.3f} ms   sweep {res['sweep forced'][0]:.3f} ms   the rule: {'LDS' if res['rule'][1] else 'sweep'} {res['rule'][0]:.3f} ms", flush=True)
