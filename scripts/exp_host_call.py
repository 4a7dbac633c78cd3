#!/usr/bin/env python3
"""The reference driver's DEFAULT call (spmm_test.py:29-35 with CPU tensors; spmm_default/pytorch_api.cpp:269-271 returns a CPU tensor) as a pipeline
of feature windows (rt_run.inc run_group_windows): ms per ``mul`` through the C ABI with host pointers, Reddit-shaped CSR,
h = 256 -- serial (host_windows = 1) against the automatic choice (0) and 2 / 3 / 4 windows, X pageable or page-locked, the result page-locked (what
the wrappers allocate) or pageable.  (The first build of the pipeline also had a copy kernel for the result's way down: 9.0 ms against the DMA engine's
8.2 with two windows -- its workgroups queue behind the product's; removed.  profiles/r06_pcie_windows.txt keeps that table.)  Every result is compared element
by element with the device-resident product (FLT32: bit-identical -- a feature window keeps each row's stored order).
usage: exp_host_call.py [dtype=FLT32] [h=256]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from pygim_amd import _lib, synth

dev = torch.device("cuda", 0)
_lib.init_ranks(1)
dtn = sys.argv[1] if len(sys.argv) > 1 else "FLT32"
h = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tdt, code = {"FLT32": (torch.float32, _lib.FLT32), "INT32": (torch.int32, _lib.INT32), "INT8": (torch.int8, _lib.INT8), "INT16": (torch.int16, _lib.INT16),
             "DBL64": (torch.float64, _lib.DBL64), "INT64": (torch.int64, _lib.INT64)}[dtn]
n, nnz, dmax = synth.SHAPES["reddit"]
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
hd = _lib.group_create(_lib.CSR, code, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
x_page = synth.features(n, h, tdt, seed=1)
x_pin = x_page.pin_memory()
want_dev = torch.empty((n, h), dtype=tdt, device=dev)
xd = x_page.to(dev)
_lib.spmm_run_group(hd, [xd.data_ptr()], want_dev.data_ptr(), 0)
torch.cuda.synchronize()
want = want_dev.cpu()
print(f"# Reddit-shaped CSR {dtn} h = {h}: {x_page.numel() * x_page.element_size() / 1e6:.1f} MB up, the same down; plan: {_lib.group_lds_note(hd)}", flush=True)


def run(x, out, reps=7):
    for _ in range(2):
        _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


out_pin = torch.empty((n, h), dtype=tdt, pin_memory=True)
out_page = torch.empty((n, h), dtype=tdt)
for xname, x in (("pageable X", x_page), ("page-locked X", x_pin)):
    for oname, out in (("page-locked C", out_pin), ("pageable C", out_page)):
        for hw, direct in ((1, 0), (0, 1), (0, 2), (2, 2), (4, 2), (0, 0), (2, 0), (3, 0), (4, 0)):
            if oname == "pageable C" and direct:
                continue
            _lib.set_tunable("host_windows", hw)
            _lib.set_tunable("host_direct", direct)
            out.zero_()
            med, best = run(x, out)
            used = _lib.group_host_call(hd)
            t = _lib.group_timers(hd)
            ok = torch.equal(out, want)
            how = "stores straight into C" if used["direct"] else "staged C, DMA engines" if used["windows"] > 1 else "serial"
            print(f"{xname:14s} {oname:14s} host_windows={hw} host_direct={direct} (ran {used['windows']} windows, {how})  {med:7.3f} ms  (min {best:7.3f})  "
                  f"up {t[0]:.2f} + product tail {t[1]:.2f} + down tail {t[2]:.2f}  {'equal to the device-resident product' if ok else 'MISMATCH'}", flush=True)
_lib.set_tunable("host_direct", 1)
_lib.set_tunable("host_windows", 0)
_lib.group_free(hd)

# ---- the same call through the Python surface (backend_pim.spmm.SparseTensorCOO.mul): the result tensor is a NEW page-locked tensor per call (pim_ops._new_out)
from pygim_amd import pim_ops
from pygim_amd.backend_pim import spmm as spmm_mod
from pygim_amd.sparse_tensor import SparseTensorShim

if dtn == "FLT32":
    pim_ops.load("spmm")
    A = spmm_mod.SparseTensorCOO(SparseTensorShim(rowptr=rowptr, col=col, sparse_sizes=(n, n)), dtype=tdt, format="CSR")
    A.to_pim_group(h, 1)
    for hw in (1, 0):
        _lib.set_tunable("host_windows", hw)
        for keep in (False, True):
            held = []
            ts = []
            for it in range(8):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = A.mul(x_page)
                ts.append((time.perf_counter() - t0) * 1e3)
                if keep:
                    held.append(out)    # (every result stays alive: the allocator must hand out fresh blocks)
            t = _lib.group_timers(A.sp_info_ptr)
            print(f"wrapper  host_windows={hw} results {'kept alive' if keep else 'dropped'}: " + " ".join(f"{v:.2f}" for v in ts) +
                  f"   last call: up {t[0]:.2f} + product tail {t[1]:.2f} + down tail {t[2]:.2f}  ptr of the last result {out.data_ptr():#x}", flush=True)
            del held
    _lib.set_tunable("host_windows", 0)
    A.free_group()
