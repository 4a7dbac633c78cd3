#!/usr/bin/env python3
"""One-off differential check at sizes between the unit tests and the full-size runs: several column panels with 16-bit ids,
a 90 000-entry row (segment kernels), all element types, weights, odd widths -- C-ABI against the CPU oracle (test tooling)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))  # run as: python tests/check_midsize.py
import numpy as np, torch
import oracle
from pygim_amd import _lib
from conftest import NP_DTYPES, ALL_DTYPES
from test_parity_gpu import run_group_host
_lib.init_ranks(1)
rng = np.random.default_rng(2024)
bad = 0
for case in range(18):
    dt = ALL_DTYPES[case % 6]; npdt = NP_DTYPES[dt]
    nrows = int(rng.integers(2000, 6000)); ncols = int(rng.choice([70000, 150000, 300000]))
    deg = rng.integers(0, 400, nrows); deg[rng.integers(0, nrows)] = 90000
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = rng.integers(0, ncols, int(rowptr[-1]))
    for r in range(nrows): col[rowptr[r]:rowptr[r+1]].sort()
    h = int(rng.choice([3, 20, 64, 130]))
    x = rng.integers(-8, 4, (ncols, h)).astype(npdt)
    vals = None if case % 3 else rng.integers(-2, 3, len(col)).astype(npdt)
    ref = oracle.spmm_csr(rowptr.astype(np.int32), col.astype(np.int32), vals, x)
    out, info = run_group_host("CSR", [rowptr.astype(np.int32)], [col.astype(np.int32)], None if vals is None else [vals], [nrows], [ncols], [x], h)
    ok = np.array_equal(out, ref)
    bad += not ok
    print(case, dt, nrows, ncols, h, vals is not None, "panels", info["n_panels"], "long", info["n_long_rows"], "OK" if ok else "MISMATCH", flush=True)
print("bad", bad)
