#!/usr/bin/env python3
"""LDS-staged product (k_lds_spmm) against the L2-blocked sweep: correctness on small graphs (vs the CPU oracle),
then Reddit-shaped CSR h = 256 timing, uniform and clustered columns.  Usage: exp_lds.py [quick]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pygim_amd import _lib, synth
import oracle

dev = torch.device("cuda", 0)
_lib.init_ranks(1)
st = 0


def product(rowptr, col, x, mode, reps=0, accumulate_into=None):
    n = rowptr.numel() - 1
    ncols, h = x.shape
    code = _lib.FLT32 if x.dtype == torch.float32 else _lib.INT32
    _lib.set_tunable("lds_mode", mode)
    t0 = time.time()
    hd = _lib.group_create(_lib.CSR, code, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [ncols], [col.numel()], [1], [h], h)
    torch.cuda.synchronize()
    t_create = time.time() - t0
    out = torch.full((n, h), 77, dtype=x.dtype, device=dev)
    _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st)
    torch.cuda.synchronize()
    ts = []
    if reps:
        for _ in range(2):
            _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(reps):
            a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
    _lib.group_free(hd)
    return out, ts, t_create


def small(nrows, ncols, h, dtype, avg=12, seed=0, long_row=0):
    g = torch.Generator().manual_seed(seed)
    deg = torch.poisson(torch.full((nrows,), float(avg)), generator=g).long()
    deg[torch.rand(nrows, generator=g) < 0.1] = 0
    if long_row:
        deg[0] = long_row
    rowptr = torch.zeros(nrows + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(deg, 0)
    nnz = int(rowptr[-1])
    row = torch.repeat_interleave(torch.arange(nrows), deg)
    col = torch.randint(0, ncols, (nnz,), generator=g)
    key, _ = torch.sort(row * ncols + col)
    col = (key % ncols).int()
    if dtype == torch.float32:
        x = torch.rand((ncols, h), generator=g) * 2 - 1
    else:
        x = torch.randint(-2**31, 2**31 - 1, (ncols, h), generator=g, dtype=torch.int64).int()
    want = oracle.spmm_csr(rowptr.numpy(), col.numpy(), None, x.numpy())
    got, _, _ = product(rowptr.int().to(dev), col.to(dev), x.to(dev), 1)
    got = got.cpu().numpy()
    bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
    tag = f"small n={nrows} ncols={ncols} h={h} {str(dtype):14s} nnz={nnz}"
    if len(bad) == 0:
        print(f"{tag}: BIT-EXACT", flush=True)
        return True
    print(f"{tag}: {len(bad)} of {got.size} elements differ; rows affected {len(set(bad[:,0]))}", flush=True)
    for r, c in bad[:8]:
        print(f"   [{r},{c}] got {got[r, c]!r} want {want[r, c]!r} deg={int(deg[r])}", flush=True)
    return False


ok = True
for waves in (8, 16):
  _lib.set_tunable("lds_waves", waves)
  print(f"--- {waves} waves per workgroup", flush=True)
  for dt in (torch.float32, torch.int32):
    ok &= small(300, 700, 64, dt)
    ok &= small(3000, 2500, 100, dt, long_row=3000)
    ok &= small(1700, 5000, 256, dt, seed=3)
    ok &= small(20000, 20000, 130, dt, avg=40, seed=5)
    ok &= small(6000, 300, 64, dt, avg=300, seed=7)     # few chunks, long slots (the in-loop touches)
print("small cases:", "all bit-exact" if ok else "MISMATCHES", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0 if ok else 1)

n, nnz, dmax = synth.SHAPES["reddit"]
h = 256
for clustered in (False, True):
    rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev, clustered=clustered)
    for dt in (torch.float32, torch.int32):
        x = synth.features(n, h, dt, seed=0, device=dev)
        ref, t_ref, c_ref = product(rowptr, col, x, 2, reps=5)
        got, t_lds, c_lds = product(rowptr, col, x, 1, reps=5)
        same = bool(torch.equal(ref, got))
        nbad = 0 if same else int((ref != got).sum())
        print(f"reddit clustered={clustered} {str(dt):14s}: sweep {min(t_ref):7.3f} ms (create {c_ref:5.2f} s)   lds {min(t_lds):7.3f} ms "
              f"(create {c_lds:5.2f} s)   equal={same} bad={nbad}", flush=True)
        if dt == torch.float32:
            xu = synth.features(n, h, dt, seed=0, device=dev, kind="uniform")
            ref, _, _ = product(rowptr, col, xu, 2)
            got, _, _ = product(rowptr, col, xu, 1)
            print(f"   uniform(-1,1) features: bit-identical to the sweep on {float((ref == got).float().mean())*100:.3f} % of outputs, "
                  f"max abs diff {float((ref - got).abs().max()):.3e}", flush=True)
