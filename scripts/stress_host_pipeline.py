#!/usr/bin/env python3
"""Randomised check of the host-operand pipeline (rt_run.inc run_group_windows): random shapes, element types, sparse parts (1 or several column blocks -> the
merged matrix), the caller's feature blocks (ds_parts), results in page-locked or pageable memory, forced window counts, the direct-store mode on and off, LDS-staged
plans forced or by the rule -- every result byte-equal to the serial call's (host_windows = 1) and, integers, to the oracle.  usage: stress_host_pipeline.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scipy.sparse as sp
import torch

import oracle
from conftest import NP_DTYPES, driver_features, random_csr
from pygim_amd import _lib

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
_lib.init_ranks(1)
CODE = {"INT8": _lib.INT8, "INT16": _lib.INT16, "INT32": _lib.INT32, "INT64": _lib.INT64, "FLT32": _lib.FLT32, "DBL64": _lib.DBL64}
seen = {"windows": 0, "direct": 0, "assembled": 0}
for c in range(cases):
    dt = list(CODE)[rng.integers(0, 6)]
    npdt = NP_DTYPES[dt]
    es = np.dtype(npdt).itemsize
    n = int(rng.integers(50, 3000))
    ncols = int(rng.integers(50, 3000))
    h = int(rng.choice([32, 64, 100, 128, 192, 256, 320, 512, 96, 7]))
    rowptr, col = random_csr(rng, n, ncols, int(rng.integers(2, 30)), empty_frac=0.1, long_rows=[(int(rng.integers(0, n)), int(rng.integers(0, 3000)))])
    x = driver_features(rng, ncols, h, npdt)
    if np.dtype(npdt).kind == "f":
        x = (x + rng.random((ncols, h))).astype(npdt)
    nparts = int(rng.choice([1, 1, 2, 3]))
    bounds = [0] + sorted(rng.choice(np.arange(1, ncols), size=nparts - 1, replace=False).tolist()) + [ncols] if nparts > 1 else [0, ncols]
    a = sp.csr_matrix((np.ones(len(col), dtype=np.int64), col.copy(), rowptr.copy()), shape=(n, ncols))
    idx0, cols_, ncs = [], [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        blk = a[:, lo:hi].tocsr()
        blk.sort_indices()
        idx0.append(np.ascontiguousarray(blk.indptr, dtype=np.int32))
        cols_.append(np.ascontiguousarray(blk.indices, dtype=np.int32))
        ncs.append(hi - lo)
    ds_parts = int(rng.choice([1, 1, 2, 3, 5]))
    xs = [np.ascontiguousarray(t.numpy()) for t in torch.chunk(torch.from_numpy(x), ds_parts, 1)] if ds_parts > 1 else [x]
    widths = [t.shape[1] for t in xs]
    lds_mode = int(rng.choice([0, 1]))
    old_mode = _lib.set_tunable("lds_mode", lds_mode)
    try:
        hd = _lib.group_create(_lib.CSR, CODE[dt], [i.ctypes.data for i in idx0], [q.ctypes.data for q in cols_], None, [n] * nparts, ncs, [len(q) for q in cols_],
                               [len(xs)] * nparts, widths * nparts, h)
    finally:
        _lib.set_tunable("lds_mode", old_mode)
    try:
        ref = None
        for hw, direct, pinned in [(1, 0, False)] + [(int(rng.integers(2, 6)), int(rng.choice([0, 2])), bool(rng.integers(0, 2))) for _ in range(4)]:
            _lib.set_tunable("host_windows", hw)
            _lib.set_tunable("host_direct", direct)
            out = torch.full((n, h), 77, dtype=torch.from_numpy(x).dtype, pin_memory=pinned)
            runs_before = _lib.group_lds_runs(hd)
            _lib.spmm_run_group(hd, [t.ctypes.data for t in xs], out.data_ptr())
            if ref is None:
                runs_serial = _lib.group_lds_runs(hd) - runs_before
            call = _lib.group_host_call(hd)
            got = out.numpy().copy()
            if ref is None:
                ref = got
                if np.dtype(npdt).kind != "f":
                    want = oracle.group(False, idx0, cols_, None, [n] * nparts, ncs, xs, h)
                    assert np.array_equal(ref, want), (c, dt, n, ncols, h, nparts, ds_parts, "serial vs oracle")
            elif np.dtype(npdt).kind == "f" and not (runs_serial == 1 and _lib.group_lds_runs(hd) - runs_before == call["windows"]):
                # float sums on the L2 sweep depend on the product's width (lanes per long row): a forced window is inside the norm-wise contract, not the serial bits
                bound = np.abs(a) @ np.abs(x.astype(np.float64))
                assert np.all(np.abs(got.astype(np.float64) - ref.astype(np.float64)) <= 2e-5 * bound + 1e-30), (c, dt, n, ncols, h, nparts, ds_parts, hw, direct, pinned, call)
                seen["sweep_floats"] = seen.get("sweep_floats", 0) + 1
            else:
                assert got.tobytes() == ref.tobytes(), (c, dt, n, ncols, h, nparts, ds_parts, hw, direct, pinned, call)
                seen["windows"] += call["windows"] > 1
                seen["direct"] += call["direct"]
                seen["assembled"] += ds_parts > 1 and call["windows"] > 1
    finally:
        _lib.set_tunable("host_windows", 0)
        _lib.set_tunable("host_direct", 1)
        _lib.group_free(hd)
print(f"host pipeline: {cases} cases, every pipelined result byte-equal to the serial call's ({seen['windows']} pipelined calls, {seen['direct']} of them with direct stores, "
      f"{seen['assembled']} over the caller's own feature blocks; {seen.get('sweep_floats', 0)} forced float windows on the sweep checked against the norm-wise bound instead)")
