import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NP_DTYPES = {
    "INT8": np.int8, "INT16": np.int16, "INT32": np.int32, "INT64": np.int64,
    "FLT32": np.float32, "DBL64": np.float64,
}
ALL_DTYPES = list(NP_DTYPES)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # built artefacts are git-ignored: a fresh checkout builds them once (hipcc cross-compiles without a GPU)
    need = [os.path.join(ROOT, "pygim_amd", "libpygim_hip.so"), os.path.join(ROOT, "oracle", "liboracle.so"),
            os.path.join(ROOT, "backend_pim", "spmm_default", "build", "libbackend_pim.so")]
    if not all(os.path.exists(f) for f in need):
        import __graft_entry__

        __graft_entry__.build()


def random_csr(rng, nrows, ncols, avg_deg, max_deg=None, empty_frac=0.1, long_rows=()):
    """Random multigraph CSR (duplicates allowed, columns sorted per row)."""
    deg = rng.poisson(avg_deg, size=nrows).astype(np.int64)
    if max_deg is not None:
        deg = np.minimum(deg, max_deg)
    deg[rng.random(nrows) < empty_frac] = 0
    for r, d in long_rows:
        deg[r] = d
    rowptr = np.zeros(nrows + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    col = rng.integers(0, ncols, size=int(rowptr[-1]), dtype=np.int64)
    for r in range(nrows):
        col[rowptr[r]:rowptr[r + 1]].sort()
    return rowptr.astype(np.int32), col.astype(np.int32)


def driver_features(rng, n, h, np_dtype):
    """X as the reference driver draws it: randint(-8, 4) (spmm_test.py:70)."""
    return rng.integers(-8, 4, size=(n, h)).astype(np_dtype)


def coalesce(rowptr, col, np_dtype):
    """(row, col, val) sorted and de-duplicated like torch's coalesce()."""
    nrows = len(rowptr) - 1
    row = np.repeat(np.arange(nrows, dtype=np.int64), np.diff(rowptr.astype(np.int64)))
    ncols = int(col.max()) + 1 if len(col) else 1
    key = row * ncols + col.astype(np.int64)
    uniq, counts = np.unique(key, return_counts=True)
    return (uniq // ncols).astype(np.int32), (uniq % ncols).astype(np.int32), counts.astype(np_dtype)


@pytest.fixture
def rng():
    return np.random.default_rng(1234)


@pytest.fixture(scope="session", autouse=True)
def _codegen_soak():
    """PYGIM_CODEGEN_VERIFY=1 (a one-off soak on the GPU box): every code stream the suite creates is generated on the device AND compared
    word for word with the host encoder inside the library (lds_codegen = 2); a difference fails the group's creation."""
    if os.environ.get("PYGIM_CODEGEN_VERIFY", "0") == "1":
        from pygim_amd import _lib

        _lib.set_tunable("lds_codegen", 2)
    yield
