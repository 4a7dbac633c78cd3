"""Parity of the HIP path against the CPU oracle, through the C ABI (include/pygim_hip.h).

Run on the GPU box: python -m pytest tests -m gpu.  Bit-exact for the integer types;
floats: bit-exact wherever one wave sums a row in stored order, and within
BASELINE.json's 1e-5 relative bound (relative to |A|.|x|) when rows are cut.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import oracle
from conftest import ALL_DTYPES, NP_DTYPES, coalesce, driver_features, random_csr
from pygim_amd import _lib

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CODE = {"INT8": _lib.INT8, "INT16": _lib.INT16, "INT32": _lib.INT32, "INT64": _lib.INT64,
        "FLT32": _lib.FLT32, "DBL64": _lib.DBL64}
CODE_OF_NP = {np.dtype(NP_DTYPES[k]): v for k, v in CODE.items()}


@pytest.fixture(scope="module", autouse=True)
def backend():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.init_ranks(1)
    yield
    _lib.release()


def _ptr(a):
    return a.ctypes.data if a is not None else 0


def run_group_host(fmt, idx0s, cols, vals, nrows, ncols, xs, h, kind="spmm", n_dense=None, dense_cols=None,
                   lds=None):
    """create -> run -> free with HOST (numpy) pointers; returns out[nrows[0], h]"""
    dt = xs[0].dtype
    idx0s = [np.ascontiguousarray(a, dtype=np.int32) for a in idx0s]
    cols = [np.ascontiguousarray(a, dtype=np.int32) for a in cols]
    vals = None if vals is None else [np.ascontiguousarray(v, dtype=dt) for v in vals]
    xs = [np.ascontiguousarray(x) for x in xs]
    n = len(cols)
    if n_dense is None:
        n_dense = [len(xs)] * n
        dense_cols = [x.shape[1] if x.ndim == 2 else 1 for x in xs] * n
    handle = _lib.group_create(_lib.COO if fmt == "COO" else _lib.CSR, CODE_OF_NP[np.dtype(dt)],
                               [_ptr(a) for a in idx0s], [_ptr(a) for a in cols],
                               None if vals is None else [_ptr(v) for v in vals], nrows, ncols,
                               [len(c) for c in cols], n_dense, dense_cols, h)
    try:
        out = np.full((nrows[0], h if kind != "spmv" else len(xs)), 77, dtype=dt)
        if kind == "spmm":
            _lib.spmm_run_group(handle, [_ptr(x) for x in xs], _ptr(out))
        elif kind == "grande":
            _lib.grande_run_group(handle, [_ptr(x) for x in xs], lds, _ptr(out))
        else:
            _lib.spmv_run_group(handle, [_ptr(x) for x in xs], _ptr(out))
        info = _lib.group_info(handle)
    finally:
        _lib.group_free(handle)
    return out, info


def abs_scale(rowptr, col, vals, x):
    n = len(rowptr) - 1
    v = np.ones(len(col)) if vals is None else np.abs(vals.astype(np.float64))
    a = sp.csr_matrix((v, col.copy(), rowptr.copy()), shape=(n, x.shape[0]))
    return a @ np.abs(x.astype(np.float64))


def test_golden_vectors_on_gpu():
    files = sorted(f for f in os.listdir(GOLDEN) if f.startswith("spmm_") and f.endswith(".npz"))
    assert len(files) >= 20
    for f in files:
        z = np.load(os.path.join(GOLDEN, f))
        fmt = str(z["fmt"])
        x, y = z["x"], z["y"]
        vals = z["vals"] if "vals" in z.files else None
        idx0 = z["rowptr"] if fmt == "CSR" else z["row"]
        nrows = y.shape[0]
        out, _ = run_group_host(fmt, [idx0], [z["col"]], None if vals is None else [vals], [nrows], [x.shape[0]],
                                [x], x.shape[1])
        if "real" in f or ("valued" in f and out.dtype.kind == "f"):
            rp = z["rowptr"]
            assert np.all(np.abs(out.astype(np.float64) - y) <= 1e-5 * abs_scale(rp, z["col"], vals, x) + 1e-30), f
        else:
            assert np.array_equal(out, y), f


def test_group_golden_vectors_on_gpu():
    """sp_parts x ds_parts fixtures: outputs of the reference's own group drivers (spmm_default/ops.hpp:42-62,97-118,
    compiled in place, tests/golden/make_golden.py group_vectors) against the HIP path through the C ABI"""
    files = sorted(f for f in os.listdir(GOLDEN) if f.startswith("group_") and f.endswith(".npz"))
    assert len(files) >= 16
    for f in files:
        z = np.load(os.path.join(GOLDEN, f))
        fmt, n, y = str(z["fmt"]), int(z["n_parts"]), z["y"]
        xs = [np.ascontiguousarray(c) for c in np.array_split(z["x"], int(z["ds_parts"]), axis=1) if c.shape[1] > 0]
        idx0, cols, vals = ([z[f"{k}_{i}"] for i in range(n)] for k in ("idx0", "col", "vals"))
        out, _ = run_group_host(fmt, idx0, cols, vals, [y.shape[0]] * n, z["ncols"].tolist(), xs, y.shape[1])
        if y.dtype.kind == "f":
            # column blocks are merged into one matrix at creation: a row's entries are summed in
            # one pass instead of block by block -> floats within the bound, not necessarily the same bits
            full = sp.hstack([sp.csr_matrix((np.abs(v.astype(np.float64)), c, r), shape=(y.shape[0], int(nc))) if fmt == "CSR"
                              else sp.coo_matrix((np.abs(v.astype(np.float64)), (r, c)), shape=(y.shape[0], int(nc))).tocsr()
                              for r, c, v, nc in zip(idx0, cols, vals, z["ncols"])]).tocsr()
            bound = 1e-5 * (full @ np.abs(z["x"].astype(np.float64))) + 1e-30
            assert np.all(np.abs(out.astype(np.float64) - y) <= bound), f
        else:
            assert np.array_equal(out, y), f


@pytest.mark.parametrize("dt", ALL_DTYPES)
@pytest.mark.parametrize("fmt", ["CSR", "COO"])
def test_widths_and_long_rows(rng, dt, fmt):
    """every element type x format over odd / tiny / wide feature counts, with rows long enough
    to be cut into segments (threshold lowered so the long-row kernels run)"""
    npdt = NP_DTYPES[dt]
    old = _lib.set_tunable("long_row_threshold", 256)
    try:
        rowptr, col = random_csr(rng, 500, 400, 11, empty_frac=0.2, long_rows=[(0, 3000), (77, 257), (499, 1200)])
        for h in (1, 3, 9, 32, 100, 256, 300):
            x = driver_features(rng, 400, h, npdt)
            if fmt == "CSR":
                ref = oracle.spmm_csr(rowptr, col, None, x)
                out, info = run_group_host("CSR", [rowptr], [col], None, [500], [400], [x], h)
                assert info["n_long_rows"] == 3
            else:
                r, c, v = coalesce(rowptr, col, npdt)
                ref = oracle.spmm_coo(r, c, v, x, 500)
                out, info = run_group_host("COO", [r], [c], [v], [500], [400], [x], h)
            # driver features are small integers: float sums are exact in any order too
            assert np.array_equal(out, ref), (dt, fmt, h)
    finally:
        _lib.set_tunable("long_row_threshold", old)


@pytest.mark.parametrize("dt", ["INT8", "INT32", "FLT32", "DBL64"])
def test_valued_entries(rng, dt):
    npdt = NP_DTYPES[dt]
    rowptr, col = random_csr(rng, 300, 300, 25)
    if np.issubdtype(npdt, np.integer):
        vals = rng.integers(-5, 6, size=len(col)).astype(npdt)
        x = driver_features(rng, 300, 64, npdt)
    else:
        vals = (rng.random(len(col)) * 2 - 1).astype(npdt)
        x = (rng.random((300, 64)) * 2 - 1).astype(npdt)
    ref = oracle.spmm_csr(rowptr, col, vals, x)
    out, info = run_group_host("CSR", [rowptr], [col], [vals], [300], [300], [x], 64)
    assert info["all_ones"] == 0
    # one wave per row, stored order, separate multiply and add: identical to the CPU loop
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("fmt", ["CSR", "COO"])
@pytest.mark.parametrize("sp_parts,ds_parts", [(1, 1), (2, 1), (3, 4), (8, 3), (1, 8)])
def test_group_partitions(rng, fmt, sp_parts, ds_parts):
    npdt = np.int32
    n, h = 333, 100
    rowptr, col = random_csr(rng, n, n, 9)
    x = driver_features(rng, n, h, npdt)
    a = sp.csr_matrix((np.ones(len(col), dtype=np.int64), col.copy(), rowptr.copy()), shape=(n, n))
    step = (n + sp_parts - 1) // sp_parts
    idx0, cols, vals, nrows, ncols = [], [], [], [], []
    for i in range(sp_parts):
        blk = a[:, i * step:min(n, (i + 1) * step)].tocsr()
        blk.sum_duplicates()
        blk.sort_indices()
        idx0.append(blk.indptr if fmt == "CSR" else blk.tocoo().row)
        cols.append(blk.indices)
        vals.append(blk.data.astype(npdt))
        nrows.append(n)
        ncols.append(blk.shape[1])
    xs = [np.ascontiguousarray(t.numpy()) for t in torch.chunk(torch.from_numpy(x), ds_parts, 1)] if ds_parts > 1 else [x]
    ref = oracle.group(fmt == "COO", idx0, cols, vals, nrows, ncols, xs, h)
    out, _ = run_group_host(fmt, idx0, cols, vals, nrows, ncols, xs, h)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("dt", ALL_DTYPES)
def test_host_operands_as_a_pipeline_of_feature_windows(rng, dt):
    """the reference driver's default call (spmm_test.py:29-35: CPU tensors in, a CPU tensor back, pytorch_api.cpp:269-271) as a pipeline of feature
    windows (rt_run.inc run_group_windows): 2 / 3 / 4 windows and the caller's own feature blocks (ds_parts), one sparse part and three (the merged
    matrix), CSR and COO, the result in pageable and in page-locked memory -- equal to the oracle AND, bit for bit, to the serial call; the group
    reports how many windows the call really moved"""
    npdt = NP_DTYPES[dt]
    es = np.dtype(npdt).itemsize
    n = 700
    for h, fmt, bounds in ((256, "CSR", [0, n]), (200, "COO", [0, n]), (320, "CSR", [0, 250, 251, n])):
        rowptr, col = random_csr(rng, n, n, 11, long_rows=[(5, 2300)])
        x = driver_features(rng, n, h, npdt)
        if np.dtype(npdt).kind == "f":
            x = (x + rng.random((n, h))).astype(npdt)
        a = sp.csr_matrix((np.ones(len(col), dtype=np.int64), col.copy(), rowptr.copy()), shape=(n, n))
        idx0, cols, nrows, ncols = [], [], [], []
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            blk = a[:, lo:hi].tocsr()
            blk.sort_indices()
            rows_of = np.repeat(np.arange(n), np.diff(blk.indptr))
            idx0.append(np.ascontiguousarray(blk.indptr if fmt == "CSR" else rows_of, dtype=np.int32))
            cols.append(np.ascontiguousarray(blk.indices, dtype=np.int32))
            nrows.append(n)
            ncols.append(hi - lo)
        # (duplicates were summed by scipy's slicing?  no: csr slicing keeps them; the reference multiplies a unit pattern, so weights stay 1 per stored entry)
        ref = oracle.group(fmt == "COO", idx0, cols, None, nrows, ncols, [x], h)
        slices = (h * es + 255) // 256
        for ds_parts in (1, 3):
            xs = [np.ascontiguousarray(t.numpy()) for t in torch.chunk(torch.from_numpy(x), ds_parts, 1)] if ds_parts > 1 else [x]
            widths = [t.shape[1] for t in xs]
            hd = _lib.group_create(_lib.COO if fmt == "COO" else _lib.CSR, CODE[dt], [_ptr(i) for i in idx0], [_ptr(c) for c in cols], None, nrows, ncols,
                                   [len(c) for c in cols], [len(xs)] * len(cols), widths * len(cols), h)
            try:
                outs = {}
                for hw in (1, 2, 3, 4):
                    old = _lib.set_tunable("host_windows", hw)
                    try:
                        for pinned in (False, True):
                            out_t = torch.full((n, h), 77, dtype=torch.from_numpy(x).dtype, pin_memory=pinned)
                            _lib.spmm_run_group(hd, [_ptr(t) for t in xs], out_t.data_ptr())
                            if hw == 1:
                                want_windows = 1
                            elif ds_parts == 1:
                                want_windows = min(hw, slices)
                            elif all(wk * es >= 512 for wk in widths):   # blocks of 512 bytes of a row and more: each is a window
                                want_windows = len(xs)
                            else:   # narrower blocks are gathered into windows of >= ceil(h / hw) columns
                                target, cur, want_windows = -(-h // hw), 0, 0
                                for i, wk in enumerate(widths):
                                    cur += wk
                                    if cur >= target or i + 1 == len(widths):
                                        want_windows, cur = want_windows + 1, 0
                            assert _lib.group_host_windows(hd) == want_windows, (dt, h, fmt, ds_parts, hw)
                            outs[(hw, pinned)] = out_t.numpy().copy()
                    finally:
                        _lib.set_tunable("host_windows", old)
                for key, got in outs.items():
                    assert np.array_equal(got, outs[(1, False)]), (dt, h, fmt, ds_parts, key, "differs from the serial call")
                if np.dtype(npdt).kind == "f":
                    scale = abs_scale(np.concatenate([[0], np.cumsum(np.diff(rowptr))]).astype(np.int32), col, None, x)
                    assert np.all(np.abs(outs[(1, False)].astype(np.float64) - ref.astype(np.float64)) <= 1e-5 * scale + 1e-30)
                else:
                    assert np.array_equal(outs[(1, False)], ref), (dt, h, fmt, ds_parts)
            finally:
                _lib.group_free(hd)
    # small operands keep the serial call when nothing is forced
    assert _lib.set_tunable("host_windows", 0) == 0


@pytest.mark.parametrize("dt", ["INT8", "INT32", "FLT32", "DBL64"])
def test_grande_windows_with_host_operands_are_gathered_into_pipeline_windows(rng, dt):
    """grande's call with CPU tensors (grande.py:12-23, 95-107: per-unit feature windows of `pad` columns each, contiguous copies, neighbours overlapping by the
    padding): with ONE sparse part the narrow windows go up as they are, are laid into their columns on the device and multiplied as windows of several units --
    equal to the oracle and to the serial call, for a result in page-locked and in pageable memory"""
    npdt = NP_DTYPES[dt]
    es = np.dtype(npdt).itemsize
    mul = 8 // es
    n, units = 900, 8
    for h in (256, 100):
        rowptr, col = random_csr(rng, n, n, 12, long_rows=[(3, 1500)])
        x = driver_features(rng, n, h, npdt)
        if np.dtype(npdt).kind == "f":
            x = (x + rng.random((n, h))).astype(npdt)
        ref = oracle.spmm_csr(rowptr, col, None, x)
        base, extra = divmod(h, units)
        widths = [base + (1 if u < extra else 0) for u in range(units)]
        pad = (widths[0] + mul - 1) // mul * mul
        tail = widths[-1] % pad
        xp = np.pad(x, ((0, 0), (0, pad - tail))) if tail else x
        wins, start = [], 0
        for w in widths:
            wins.append(np.ascontiguousarray(xp[:, start:start + pad]))
            start += w
        rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
        hd = _lib.group_create(_lib.CSR, CODE[dt], [_ptr(rp)], [_ptr(ci)], None, [n], [n], [len(ci)], [units], widths, h)
        try:
            outs = {}
            for hw in (1, 2, 3):
                old = _lib.set_tunable("host_windows", hw)
                try:
                    for pinned in (False, True):
                        out_t = torch.full((n, h), 77, dtype=torch.from_numpy(x).dtype, pin_memory=pinned)
                        _lib.grande_run_group(hd, [_ptr(w) for w in wins], [pad] * units, out_t.data_ptr())
                        got_windows = _lib.group_host_windows(hd)
                        assert (got_windows == 1) if hw == 1 else (2 <= got_windows <= units), (dt, h, hw, got_windows)
                        outs[(hw, pinned)] = out_t.numpy().copy()
                finally:
                    _lib.set_tunable("host_windows", old)
            for key, got in outs.items():
                assert np.array_equal(got, outs[(1, False)]), (dt, h, key, "differs from the serial call")
            if np.dtype(npdt).kind == "f":
                scale = abs_scale(rowptr, col, None, x)
                assert np.all(np.abs(outs[(1, False)].astype(np.float64) - ref.astype(np.float64)) <= 1e-5 * scale + 1e-30)
            else:
                assert np.array_equal(outs[(1, False)], ref), (dt, h)
        finally:
            _lib.group_free(hd)


def test_device_pointers_and_block_run(rng):
    """device-resident operands: no staging, result stays in HBM"""
    npdt = np.float32
    rowptr, col = random_csr(rng, 1000, 1000, 30, long_rows=[(5, 6000)])
    x = (rng.random((1000, 256)) * 2 - 1).astype(npdt)
    ref = oracle.spmm_csr(rowptr, col, None, x)
    d = lambda a: torch.from_numpy(a).cuda()
    drp, dcol, dx = d(rowptr), d(col), d(x)
    handle = _lib.group_create(_lib.CSR, _lib.FLT32, [drp.data_ptr()], [dcol.data_ptr()], None, [1000], [1000],
                               [len(col)], [1], [256], 256)
    out = torch.empty((1000, 256), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.spmm_run_group(handle, [dx.data_ptr()], out.data_ptr(), st)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    scale = abs_scale(rowptr, col, None, x)
    assert np.all(np.abs(got.astype(np.float64) - ref) <= 1e-5 * scale)
    short = np.diff(rowptr) <= 4096
    assert np.array_equal(got[short], ref[short])  # single-wave rows: bit-identical
    # accumulate form of the block product: C += A.X
    out2 = out.clone()
    _lib.block_run(handle, 0, dx.data_ptr(), 256, out2.data_ptr(), 256, 256, True, st)
    torch.cuda.synchronize()
    assert np.all(np.abs(out2.cpu().numpy().astype(np.float64) - 2 * ref.astype(np.float64)) <= 2e-5 * scale)
    _lib.group_free(handle)


def test_error_behaviour(rng):
    rowptr, col = random_csr(rng, 50, 50, 5)
    x = driver_features(rng, 50, 8, np.int32)
    r, c, v = coalesce(rowptr, col, np.int32)
    with pytest.raises(_lib.PygimError) as e:  # unsorted COO
        run_group_host("COO", [r[::-1].copy()], [c], [v], [50], [50], [x], 8)
    assert e.value.code == _lib.ERR_UNSORTED
    bad = col.copy()
    bad[0] = 50
    with pytest.raises(_lib.PygimError):  # column id out of range
        run_group_host("CSR", [rowptr], [bad], None, [50], [50], [x], 8)
    with pytest.raises(_lib.PygimError):  # dense widths do not add up to h
        run_group_host("CSR", [rowptr], [col], None, [50], [50], [x], 9)
    with pytest.raises(_lib.PygimError):
        _lib.spmm_run_group(12345, [x.ctypes.data], x.ctypes.data)
    # empty matrix: output must be all zeros
    out, _ = run_group_host("CSR", [np.zeros(51, np.int32)], [np.zeros(0, np.int32)], None, [50], [50], [x], 8)
    assert not out.any()
    out, _ = run_group_host("COO", [np.zeros(0, np.int32)], [np.zeros(0, np.int32)], [np.zeros(0, np.int32)], [50],
                            [50], [x], 8)
    assert not out.any()
    # no rows at all
    out, _ = run_group_host("CSR", [np.zeros(1, np.int32)], [np.zeros(0, np.int32)], None, [0], [50], [x], 8)
    assert out.shape == (0, 8)


@pytest.mark.parametrize("dt", ALL_DTYPES)
def test_panel_kernel_all_types(rng, dt):
    """the L2-blocked panel sweep (forced, with tiny panels so every row crosses many of them)"""
    npdt = NP_DTYPES[dt]
    old = [_lib.set_tunable("panel_mode", 1), _lib.set_tunable("panel_bytes", 128 * 40)]
    try:
        rowptr, col = random_csr(rng, 700, 500, 14, empty_frac=0.15, long_rows=[(3, 2500), (699, 900)])
        for h in (16, 32, 100, 256, 300):
            x = driver_features(rng, 500, h, npdt)
            ref = oracle.spmm_csr(rowptr, col, None, x)
            out, _ = run_group_host("CSR", [rowptr], [col], None, [700], [500], [x], h)
            assert np.array_equal(out, ref), (dt, h)
    finally:
        _lib.set_tunable("panel_mode", old[0])
        _lib.set_tunable("panel_bytes", old[1])


@pytest.mark.parametrize("dt", ALL_DTYPES)
def test_panel_sweep_takes_rows_of_any_alignment(rng, dt):
    """feature widths whose rows are not 16-byte multiples (41, 255, 9 ...), X / C base pointers off by one element,
    padded strides, accumulation over unmerged sparse parts: the sweep's byte-aligned 16-byte accesses give the oracle's
    result (these shapes used the row-per-wave kernels before: Reddit f32 h = 255 23 -> 7.7 ms, int8 h = 100 14.8 -> 1.4 ms)"""
    npdt = NP_DTYPES[dt]
    tdt = torch.from_numpy(np.zeros(1, dtype=npdt)).dtype
    nrows, ncols = 400, 600
    rowptr, col = random_csr(rng, nrows, ncols, 25, empty_frac=0.1, long_rows=[(8, 2500)])
    vals = rng.integers(-3, 4, size=len(col)).astype(npdt)
    old = {k: _lib.set_tunable(k, v) for k, v in {"panel_mode": 1, "panel_bytes": 128 * 150, "merge_parts": 0}.items()}
    try:
        for h in (9, 41, 255, 33):
            x = driver_features(rng, ncols, h, npdt)
            for v in (None, vals):
                ref = oracle.spmm_csr(rowptr, col, v, x)
                out, _ = run_group_host("CSR", [rowptr], [col], None if v is None else [v], [nrows], [ncols], [x], h)
                assert np.array_equal(out, ref), (dt, h, v is not None, "host")
            # device operands inside larger buffers: base off by one element, row strides h + 3 / h + 1
            ldx, ldc = h + 3, h + 1
            xb = torch.zeros(ncols * ldx + 1, dtype=tdt, device="cuda")
            xb[1:].view(ncols, ldx)[:, :h] = torch.from_numpy(x).cuda()
            cb = torch.full((nrows * ldc + 1,), 7, dtype=tdt, device="cuda")
            es = xb.element_size()
            hd = _lib.group_create(_lib.CSR, CODE_OF_NP[np.dtype(npdt)], [_ptr(rowptr)], [_ptr(col)], None, [nrows], [ncols], [len(col)],
                                   [1], [h], h)
            try:
                st = torch.cuda.current_stream().cuda_stream
                _lib.block_run(hd, 0, xb.data_ptr() + es, ldx, cb.data_ptr() + es, ldc, h, False, st)
                _lib.block_run(hd, 0, xb.data_ptr() + es, ldx, cb.data_ptr() + es, ldc, h, True, st)   # C += A.X once more
                torch.cuda.synchronize()
            finally:
                _lib.group_free(hd)
            got = cb[1:].view(nrows, ldc).cpu().numpy()
            want = oracle.spmm_csr(rowptr, col, None, x)
            want2 = (want.astype(np.int64) * 2).astype(npdt) if np.issubdtype(npdt, np.integer) else want * 2
            assert np.array_equal(got[:, :h], want2), (dt, h, "device, strided")
            assert (got[:, h:] == 7).all() and int(cb[0]) == 7, "wrote outside the window"
    finally:
        for k, v in old.items():
            _lib.set_tunable(k, v)


@pytest.mark.parametrize("dt", ALL_DTYPES)
def test_csr_vector_kernel_for_rows_of_one_to_four_elements(rng, dt):
    """rows of X of 1..4 elements (the SpMV end, k_csr_vec: lanes over a row's entries, fixed butterfly): every lane-group
    size (picked from the mean degree), weights, empty and very long rows, strided X / C with accumulation, against the
    sweep and the oracle (driver features: float sums exact in any order)"""
    npdt = NP_DTYPES[dt]
    tdt = torch.from_numpy(np.zeros(1, dtype=npdt)).dtype
    for deg in (3, 12, 30, 60, 150):
        nrows, ncols = 300, 400
        rowptr, col = random_csr(rng, nrows, ncols, deg, empty_frac=0.1, long_rows=[(6, 5000)])
        vals = rng.integers(-3, 4, size=len(col)).astype(npdt)
        for w in (1, 2, 3, 4):
            x = driver_features(rng, ncols, w, npdt)
            for v in (None, vals):
                ref = oracle.spmm_csr(rowptr, col, v, x)
                for vk in (1, 0):
                    old = _lib.set_tunable("vec_kernel", vk)
                    try:
                        out, _ = run_group_host("CSR", [rowptr], [col], None if v is None else [v], [nrows], [ncols], [x], w)
                    finally:
                        _lib.set_tunable("vec_kernel", old)
                    assert np.array_equal(out, ref), (dt, deg, w, v is not None, vk)
            # strided device operands, C += A.X on top of a first product
            ldx, ldc = w + 2, w + 1
            xb = torch.zeros((ncols, ldx), dtype=tdt, device="cuda")
            xb[:, :w] = torch.from_numpy(x).cuda()
            cb = torch.full((nrows, ldc), 5, dtype=tdt, device="cuda")
            hd = _lib.group_create(_lib.CSR, CODE_OF_NP[np.dtype(npdt)], [_ptr(rowptr)], [_ptr(col)], None, [nrows], [ncols], [len(col)],
                                   [1], [w], w)
            try:
                st = torch.cuda.current_stream().cuda_stream
                _lib.block_run(hd, 0, xb.data_ptr(), ldx, cb.data_ptr(), ldc, w, False, st)
                _lib.block_run(hd, 0, xb.data_ptr(), ldx, cb.data_ptr(), ldc, w, True, st)
                torch.cuda.synchronize()
            finally:
                _lib.group_free(hd)
            one = oracle.spmm_csr(rowptr, col, None, x)
            two = (one.astype(np.int64) * 2).astype(npdt) if np.issubdtype(npdt, np.integer) else one * 2
            got = cb.cpu().numpy()
            assert np.array_equal(got[:, :w], two) and (got[:, w:] == 5).all(), (dt, deg, w, "strided")


@pytest.mark.parametrize("dt", ALL_DTYPES)
def test_spmv_group_through_the_c_abi(rng, dt):
    """pygim_spmv_run_group: `groups` right-hand sides of one column each -> out[nrows, groups]; one and three sparse
    parts (column blocks, merged or part by part), host and device vectors"""
    npdt = NP_DTYPES[dt]
    nrows, ncols = 333, 500
    rowptr, col = random_csr(rng, nrows, ncols, 9, empty_frac=0.1, long_rows=[(4, 1500)])
    rows_of = np.repeat(np.arange(nrows), np.diff(rowptr))
    for groups in (1, 3, 8):
        x = driver_features(rng, ncols, groups, npdt)
        ref = oracle.spmm_csr(rowptr, col, None, x)
        vecs = [np.ascontiguousarray(x[:, j]) for j in range(groups)]
        for bounds in ([0, ncols], [0, 170, 171, ncols]):
            rp, cl, nc = [], [], []
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                k = (col >= lo) & (col < hi)
                rp.append(np.concatenate([[0], np.cumsum(np.bincount(rows_of[k], minlength=nrows))]).astype(np.int32))
                cl.append((col[k] - lo).astype(np.int32))
                nc.append(hi - lo)
            n = len(nc)
            for merge in (1, 0):
                old = _lib.set_tunable("merge_parts", merge)
                try:
                    out, _ = run_group_host("CSR", rp, cl, None, [nrows] * n, nc, vecs, groups, kind="spmv", n_dense=[groups] * n,
                                            dense_cols=[1] * groups * n)
                    assert np.array_equal(out, ref), (dt, groups, n, merge, "host")
                    hd = _lib.group_create(_lib.CSR, CODE_OF_NP[np.dtype(npdt)], [_ptr(a) for a in rp], [_ptr(a) for a in cl], None,
                                           [nrows] * n, nc, [len(c) for c in cl], [groups] * n, [1] * groups * n, groups)
                    dv = [torch.from_numpy(v).cuda() for v in vecs]
                    od = torch.empty((nrows, groups), dtype=dv[0].dtype, device="cuda")
                    _lib.spmv_run_group(hd, [v.data_ptr() for v in dv], od.data_ptr(), torch.cuda.current_stream().cuda_stream)
                    torch.cuda.synchronize()
                    _lib.group_free(hd)
                    assert np.array_equal(od.cpu().numpy(), ref), (dt, groups, n, merge, "device")
                finally:
                    _lib.set_tunable("merge_parts", old)


@pytest.mark.parametrize("dt", ["INT8", "INT16", "INT32", "INT64"])
@pytest.mark.parametrize("fmt", ["CSR", "COO"])
def test_mostly_unit_integer_weights_split_into_pattern_plus_corrections(rng, dt, fmt):
    """a coalesced multigraph's weights (1 almost everywhere, spmm.py:40-42): the group keeps the unit pattern in the hot
    loop and a small correction part with (weight - 1); exact in modular arithmetic, equal to the unsplit run and the oracle,
    including weights 0, negative and wrapping ones, rows made only of corrections, and sp_parts > 1"""
    npdt = NP_DTYPES[dt]
    nrows, ncols, h = 700, 900, 96
    rowptr, col = random_csr(rng, nrows, ncols, 40, empty_frac=0.05, long_rows=[(11, 3000)])
    if fmt == "COO":
        # strictly increasing columns per row (coalesced pattern)
        keep = np.ones(len(col), dtype=bool)
        keep[1:] = (col[1:] != col[:-1]) | np.isin(np.arange(1, len(col)), rowptr[1:-1])
        rows_of = np.repeat(np.arange(nrows), np.diff(rowptr))[keep]
        col = col[keep]
        rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows_of, minlength=nrows))]).astype(np.int32)
    vals = np.ones(len(col), dtype=npdt)
    pick = rng.choice(len(col), size=max(len(col) // 200, 3), replace=False)
    info = np.iinfo(npdt)
    vals[pick] = rng.choice(np.array([0, 2, 3, -1, -5, info.max, info.min], dtype=npdt), size=len(pick))
    x = driver_features(rng, ncols, h, npdt)
    ref = oracle.spmm_csr(rowptr, col, vals, x)
    rows_of = np.repeat(np.arange(nrows), np.diff(rowptr)).astype(np.int32)
    outs = {}
    for split in (1, 0):
        old = _lib.set_tunable("split_unit_pattern", split)
        try:
            outs[split], _ = run_group_host(fmt, [rowptr if fmt == "CSR" else rows_of], [col], [vals], [nrows], [ncols], [x], h)
        finally:
            _lib.set_tunable("split_unit_pattern", old)
        assert np.array_equal(outs[split], ref), (dt, fmt, split)
    # two sparse parts (column split), device-resident operands
    step = ncols // 2
    parts = []
    for lo, hi in ((0, step), (step, ncols)):
        k = (col >= lo) & (col < hi)
        parts.append((np.concatenate([[0], np.cumsum(np.bincount(rows_of[k], minlength=nrows))]).astype(np.int32),
                      (col[k] - lo).astype(np.int32), rows_of[k], vals[k], hi - lo))
    out2, _ = run_group_host(fmt, [p[0] if fmt == "CSR" else p[2] for p in parts], [p[1] for p in parts], [p[3] for p in parts],
                             [nrows, nrows], [p[4] for p in parts], [x], h)
    assert np.array_equal(out2, ref), (dt, fmt, "two parts")


@pytest.mark.parametrize("dt", ALL_DTYPES)
def test_panel_sweep_with_16bit_panel_local_column_ids(rng, dt):
    """the plan's 16-bit panel-local column ids (tunable panel_col16, default on; panels of <= 65536 columns) against
    the 32-bit ids and the oracle: many small panels, one panel, ragged row ends at odd 2-byte alignments, weighted,
    wave-cooperative long items, > 65536 columns in one panel (falls back to 32-bit ids)"""
    npdt = NP_DTYPES[dt]
    cases = [(500, 700, 128 * 50, 11), (300, 60000, 1 << 40, 5), (200, 70000, 1 << 40, 4), (64, 3000, 128 * 1000, 300)]
    for nrows, ncols, pbytes, deg in cases:
        rowptr, col = random_csr(rng, nrows, ncols, deg, empty_frac=0.1, long_rows=[(2, 1200)])
        vals = rng.integers(-3, 4, size=len(col)).astype(npdt)
        x = driver_features(rng, ncols, 64, npdt)
        for v in (None, vals):
            ref = oracle.spmm_csr(rowptr, col, v, x)
            for c16 in (1, 0):
                old = {k: _lib.set_tunable(k, val) for k, val in
                       {"panel_mode": 1, "panel_bytes": pbytes, "panel_col16": c16, "panel_coop": 64}.items()}
                try:
                    out, _ = run_group_host("CSR", [rowptr], [col], None if v is None else [v], [nrows], [ncols], [x], 64)
                finally:
                    for k, val in old.items():
                        _lib.set_tunable(k, val)
                assert np.array_equal(out, ref), (dt, nrows, ncols, c16, v is not None)


@pytest.mark.parametrize("dt", ALL_DTYPES)
def test_dense_windows_fused_into_one_product(rng, dt):
    """several dense windows per sparse part (ds_parts chunks; grande's per-unit windows with padded strides) run as ONE
    product of the full width (tunable fuse_windows, default on): same result as window by window and as the oracle,
    for host and device operands, separate and side-by-side windows"""
    npdt = NP_DTYPES[dt]
    n, ncols_parts = 400, [150, 170]
    rowptr, col = random_csr(rng, n, sum(ncols_parts), 12, empty_frac=0.1, long_rows=[(7, 900)])
    # column split with local ids (spmm.py:127-136)
    rp, cl = [], []
    lo = 0
    for w in ncols_parts:
        keep = (col >= lo) & (col < lo + w)
        rows_of = np.repeat(np.arange(n), np.diff(rowptr))
        rp.append(np.concatenate([[0], np.cumsum(np.bincount(rows_of[keep], minlength=n))]).astype(np.int32))
        cl.append((col[keep] - lo).astype(np.int32))
        lo += w
    for widths in ([32, 32, 32, 32], [3, 2, 2], [40, 24], [5] * 9):
        h = sum(widths)
        x = driver_features(rng, sum(ncols_parts), h, npdt)
        ref = oracle.spmm_csr(rowptr, col, None, x)
        chunks = [np.ascontiguousarray(x[:, a:a + w]) for a, w in zip(np.cumsum([0] + widths[:-1]), widths)]
        outs = {}
        for fuse in (1, 0):
            old = _lib.set_tunable("fuse_windows", fuse)
            try:
                outs[fuse], _ = run_group_host("CSR", rp, cl, None, [n, n], ncols_parts, chunks, h)
            finally:
                _lib.set_tunable("fuse_windows", old)
            assert np.array_equal(outs[fuse], ref), (dt, widths, fuse)
        # grande: per-part windows with strides padded to 8 bytes (grande.py:12-23), on the device
        pad = lambda w: -(-w * npdt().itemsize // 8) * 8 // npdt().itemsize
        wins, lds = [], []
        lo = 0
        for wcols in ncols_parts:
            a = 0
            for w in widths:
                buf = np.full((wcols, pad(w)), 99, dtype=npdt)
                buf[:, :w] = x[lo:lo + wcols, a:a + w]
                wins.append(buf)
                lds.append(pad(w))
                a += w
            lo += wcols
        out_g, _ = run_group_host("CSR", rp, cl, None, [n, n], ncols_parts, wins, h, kind="grande",
                                  n_dense=[len(widths)] * 2, dense_cols=widths * 2, lds=lds)
        assert np.array_equal(out_g, ref), (dt, widths, "grande")
        # device operands, windows side by side inside one row-major matrix: used in place
        xd = torch.from_numpy(x).cuda()
        od = torch.empty((n, h), dtype=xd.dtype, device="cuda")
        es = xd.element_size()
        hd = _lib.group_create(_lib.CSR, CODE_OF_NP[np.dtype(npdt)], [_ptr(a) for a in rp], [_ptr(a) for a in cl], None, [n, n],
                               ncols_parts, [len(c) for c in cl], [len(widths)] * 2, widths * 2, h)
        try:
            ptrs, ldl = [], []
            lo = 0
            for wcols in ncols_parts:
                a = 0
                for w in widths:
                    ptrs.append(xd.data_ptr() + (lo * h + a) * es)
                    ldl.append(h)
                    a += w
                lo += wcols
            _lib.grande_run_group(hd, ptrs, ldl, od.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        finally:
            _lib.group_free(hd)
        assert np.array_equal(od.cpu().numpy(), ref), (dt, widths, "in place")


@pytest.mark.parametrize("dt", ALL_DTYPES)
@pytest.mark.parametrize("group_slices", [1, 3])
def test_panel_sweep_in_slice_groups(rng, dt, group_slices):
    """X beyond the Infinity Cache is swept a few feature slices per launch (tunable slice_group_bytes):
    forced here on a small graph, 1 or 3 slices at a time, ragged last group, weighted and unweighted,
    slice-major copy and row-major gathers"""
    npdt = NP_DTYPES[dt]
    ncols = 500
    old = {k: _lib.set_tunable(k, v) for k, v in
           {"panel_mode": 1, "panel_bytes": 128 * 200, "slice_group_bytes": ncols * 128 * group_slices}.items()}
    try:
        rowptr, col = random_csr(rng, 300, ncols, 9, empty_frac=0.1, long_rows=[(5, 1500)])
        vals = rng.integers(-3, 4, size=len(col)).astype(npdt)
        for pack in (1, 0):
            old_pack = _lib.set_tunable("panel_pack", pack)
            for h in (32, 100, 256, 264):
                x = driver_features(rng, ncols, h, npdt)
                for v in (None, vals):
                    ref = oracle.spmm_csr(rowptr, col, v, x)
                    out, _ = run_group_host("CSR", [rowptr], [col], None if v is None else [v], [300], [ncols], [x], h)
                    assert np.array_equal(out, ref), (dt, h, pack, v is not None)
            _lib.set_tunable("panel_pack", old_pack)
    finally:
        for k, v in old.items():
            _lib.set_tunable(k, v)


@pytest.mark.parametrize("dt", ["FLT32", "DBL64"])
def test_panel_kernel_keeps_stored_order(rng, dt):
    """real-valued features and weights: the panel sweep continues each row's running sum from C,
    so the result is bit-identical to the sequential CPU loop (no tolerance needed)"""
    npdt = NP_DTYPES[dt]
    old = [_lib.set_tunable("panel_mode", 1), _lib.set_tunable("panel_bytes", 128 * 64)]
    old_long = _lib.set_tunable("long_row_threshold", 1 << 20)  # keep every row in one ordered sweep
    old_coop = _lib.set_tunable("panel_coop", 1 << 20)           # ... walked by ONE lane group (stored order)
    try:
        rowptr, col = random_csr(rng, 400, 600, 40, long_rows=[(11, 5000)])
        x = (rng.random((600, 64)) * 2 - 1).astype(npdt)
        vals = (rng.random(len(col)) * 2 - 1).astype(npdt)
        for v in (None, vals):
            ref = oracle.spmm_csr(rowptr, col, v, x)
            out, _ = run_group_host("CSR", [rowptr], [col], None if v is None else [v], [400], [600], [x], 64)
            assert np.array_equal(out, ref)
        # sp_parts = 2 with panels: second part accumulates onto the first
        half = 300
        a = sp.csr_matrix((np.ones(len(col)), col.copy(), rowptr.copy()), shape=(400, 600))
        parts = [a[:, :half].tocsr(), a[:, half:].tocsr()]
        for b in parts:
            b.sort_indices()
        xs = [x]
        ref = oracle.group(False, [b.indptr for b in parts], [b.indices for b in parts],
                           [b.data.astype(npdt) for b in parts], [400, 400], [half, 600 - half], xs, 64)
        out, _ = run_group_host("CSR", [b.indptr for b in parts], [b.indices for b in parts],
                                [b.data.astype(npdt) for b in parts], [400, 400], [half, 600 - half], xs, 64)
        scale = abs_scale(rowptr, col, None, x)
        assert np.all(np.abs(out.astype(np.float64) - ref) <= 1e-5 * scale)
    finally:
        _lib.set_tunable("panel_mode", old[0])
        _lib.set_tunable("panel_bytes", old[1])
        _lib.set_tunable("long_row_threshold", old_long)
        _lib.set_tunable("panel_coop", old_coop)


def test_panel_kernel_banded_rows(rng):
    """rows whose entries sit in one or two column panels only (community-like graphs): a row
    appears in the work list of a panel only when it has entries there; empty rows give zeros"""
    npdt = np.float32
    n, ncols, h = 900, 4000, 96
    deg = rng.integers(0, 60, size=n)
    deg[::7] = 0
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    col = np.empty(int(rowptr[-1]), dtype=np.int64)
    for r in range(n):
        centre = int(r * ncols / n)
        c = np.clip(centre + rng.integers(-150, 150, size=deg[r]), 0, ncols - 1)
        col[rowptr[r]:rowptr[r + 1]] = np.sort(c)
    rowptr, col = rowptr.astype(np.int32), col.astype(np.int32)
    x = (rng.random((ncols, h)) * 2 - 1).astype(npdt)
    ref = oracle.spmm_csr(rowptr, col, None, x)
    old = [_lib.set_tunable("panel_mode", 1), _lib.set_tunable("panel_bytes", 128 * 256)]
    try:
        out, _ = run_group_host("CSR", [rowptr], [col], None, [n], [ncols], [x], h)
        assert np.array_equal(out, ref)
    finally:
        _lib.set_tunable("panel_mode", old[0])
        _lib.set_tunable("panel_bytes", old[1])


@pytest.mark.parametrize("dt", ALL_DTYPES)
def test_panel_cooperative_items(rng, dt):
    """items longer than panel_coop are walked by a whole wave (8 lane groups, partial sums added):
    exact for integers and for the integer-valued driver features in every float type; real-valued
    floats stay inside the 1e-5 bound"""
    npdt = NP_DTYPES[dt]
    old = [_lib.set_tunable("panel_mode", 1), _lib.set_tunable("panel_bytes", 128 * 300), _lib.set_tunable("panel_coop", 64)]
    try:
        rowptr, col = random_csr(rng, 300, 900, 30, long_rows=[(0, 9000), (7, 65), (100, 700), (299, 2049)])
        for h in (32, 96, 256):
            x = driver_features(rng, 900, h, npdt)
            vals = rng.integers(1, 4, size=len(col)).astype(npdt)
            for v in (None, vals):
                ref = oracle.spmm_csr(rowptr, col, v, x)
                out, _ = run_group_host("CSR", [rowptr], [col], None if v is None else [v], [300], [900], [x], h)
                assert np.array_equal(out, ref), (dt, h, v is None)
        if dt in ("FLT32", "DBL64"):
            x = (rng.random((900, 64)) * 2 - 1).astype(npdt)
            ref = oracle.spmm_csr(rowptr, col, None, x)
            out, _ = run_group_host("CSR", [rowptr], [col], None, [300], [900], [x], 64)
            assert np.all(np.abs(out.astype(np.float64) - ref) <= 1e-5 * abs_scale(rowptr, col, None, x))
    finally:
        _lib.set_tunable("panel_mode", old[0])
        _lib.set_tunable("panel_bytes", old[1])
        _lib.set_tunable("panel_coop", old[2])


def test_unsorted_columns_inside_rows(rng):
    """CSR whose rows hold their column ids in arbitrary order: column panels (binary search per row) must not
    be used; the result is still exact"""
    rowptr, col = random_csr(rng, 400, 3000, 40)
    col = col.copy()
    for r in range(400):
        rng.shuffle(col[rowptr[r]:rowptr[r + 1]])
    x = driver_features(rng, 3000, 64, np.int32)
    ref = oracle.spmm_csr(rowptr, col, None, x)
    old = [_lib.set_tunable("panel_mode", 1), _lib.set_tunable("panel_bytes", 128 * 100)]
    try:
        out, info = run_group_host("CSR", [rowptr], [col], None, [400], [3000], [x], 64)
        assert info["n_panels"] == 1
        assert np.array_equal(out, ref)
    finally:
        _lib.set_tunable("panel_mode", old[0])
        _lib.set_tunable("panel_bytes", old[1])


@pytest.mark.parametrize("dt", ALL_DTYPES)
def test_lds_staged_spmv_kernel(rng, dt):
    """k_spmv_lds (a column panel of X staged in a workgroup's LDS): rows of X of 1..4 elements, several panels with more than
    32 entries per (row, panel), unit and real weights, CSR and COO, dense and strided X, two unmerged sparse parts (the
    second accumulates), against the oracle and against the cache-path vector kernel (tunable vec_lds = 0)"""
    npdt = NP_DTYPES[dt]
    n = 2600
    rowptr, col = random_csr(rng, n, n, 160, long_rows=[(7, 3000), (n - 2, 900)], empty_frac=0.03)
    old = {k: _lib.set_tunable(k, v) for k, v in (("panel_bytes", 128 * 900), ("merge_parts", 0))}
    try:
        for w in (1, 2, 3, 4):
            for weighted in (False, True):
                x = driver_features(rng, n, w, npdt)
                vals = rng.integers(-3, 4, size=len(col)).astype(npdt) if weighted else None
                ref = oracle.spmm_csr(rowptr, col, vals, x)
                outs = {}
                for lds_on in (1, 0):
                    _lib.set_tunable("vec_lds", lds_on)
                    out, info = run_group_host("CSR", [rowptr], [col], None if vals is None else [vals], [n], [n], [x], w)
                    outs[lds_on] = out
                    if lds_on:
                        assert info["n_panels"] >= 3, info  # the LDS kernel's rule is met: 160 / 3 entries per (row, panel)
                if np.issubdtype(npdt, np.integer) or not weighted:
                    assert np.array_equal(outs[1], ref) and np.array_equal(outs[0], ref), (dt, w, weighted)
                else:
                    bound = 1e-5 * abs_scale(rowptr, col, vals, x) + 1e-30
                    assert np.all(np.abs(outs[1].astype(np.float64) - ref) <= bound), (dt, w)
        _lib.set_tunable("vec_lds", 1)
        # COO (coalesced: values > 1), w = 1
        r, c, v = coalesce(rowptr, col, npdt)
        x = driver_features(rng, n, 1, npdt)
        out, _ = run_group_host("COO", [r], [c], [v], [n], [n], [x], 1)
        assert np.array_equal(out, oracle.spmm_coo(r, c, v, x, n))
        # the SpMV entry point: 2 right-hand sides packed into [n, 2]
        xs = [driver_features(rng, n, 1, npdt) for _ in range(2)]
        out, _ = run_group_host("CSR", [rowptr], [col], None, [n], [n], xs, 2, kind="spmv", n_dense=[2], dense_cols=[1, 1])
        assert np.array_equal(out, np.concatenate([oracle.spmm_csr(rowptr, col, None, a) for a in xs], axis=1))
        # strided X and C through pygim_block_run (element-wise staging), then a second part that accumulates
        hd = None
        rp_d, cl_d = torch.from_numpy(rowptr).cuda(), torch.from_numpy(col).cuda()
        tdt = {"INT8": torch.int8, "INT16": torch.int16, "INT32": torch.int32, "INT64": torch.int64, "FLT32": torch.float32,
               "DBL64": torch.float64}[dt]
        hd = _lib.group_create(_lib.CSR, CODE_OF_NP[np.dtype(npdt)], [rp_d.data_ptr()], [cl_d.data_ptr()], None, [n], [n],
                               [len(col)], [1], [2], 2)
        xw = torch.from_numpy(driver_features(rng, n, 7, npdt)).cuda()
        cw = torch.zeros((n, 5), dtype=tdt, device="cuda")
        es = xw.element_size()
        _lib.block_run(hd, 0, xw.data_ptr() + 3 * es, 7, cw.data_ptr() + 1 * es, 5, 2, False)
        _lib.block_run(hd, 0, xw.data_ptr() + 3 * es, 7, cw.data_ptr() + 1 * es, 5, 2, True)  # accumulate: twice the product
        torch.cuda.synchronize()
        want = oracle.spmm_csr(rowptr, col, None, xw[:, 3:5].cpu().numpy())
        got = cw.cpu().numpy()
        assert np.array_equal(got[:, 1:3], (want.astype(np.float64) * 2).astype(npdt) if np.issubdtype(npdt, np.floating)
                              else (want.astype(np.int64) * 2).astype(npdt))
        assert not got[:, 0].any() and not got[:, 3:].any()
        _lib.group_free(hd)
    finally:
        _lib.set_tunable("vec_lds", 1)
        for k, v in old.items():
            _lib.set_tunable(k, v)


@pytest.mark.parametrize("dt", ["INT32", "FLT32", "INT8"])
def test_lds_staged_spmv_more_panels_than_workgroups(rng, dt):
    """k_spmv_lds when the (panel, slot) units outnumber the CUs: a workgroup then takes several units in turn, re-staging
    its LDS panel between them (panels of 8 columns -> 400 panels; the entries-per-panel rule is switched off), with all four
    length classes present (items of 0..600 entries inside one panel are impossible at 8 columns, so a second, wide-panel run
    covers the long classes)"""
    npdt = NP_DTYPES[dt]
    n = 3200
    rowptr, col = random_csr(rng, n, n, 40, long_rows=[(5, 2500), (n - 1, 700)], empty_frac=0.05)
    for panel_bytes, want_panels in ((128 * 8, 300), (128 * 1600, 2)):
        old = {k: _lib.set_tunable(k, v) for k, v in (("panel_bytes", panel_bytes), ("vec_lds_min_seg", 0), ("merge_parts", 0),
                                                       ("panel_mode", 1))}
        try:
            for w, weighted in ((1, False), (2, True), (4, False)):
                x = driver_features(rng, n, w, npdt)
                vals = rng.integers(-3, 4, size=len(col)).astype(npdt) if weighted else None
                ref = oracle.spmm_csr(rowptr, col, vals, x)
                out, info = run_group_host("CSR", [rowptr], [col], None if vals is None else [vals], [n], [n], [x], w)
                assert info["n_panels"] >= want_panels, info
                if np.issubdtype(npdt, np.integer) or not weighted:
                    assert np.array_equal(out, ref), (dt, w, weighted, panel_bytes)
                else:
                    assert np.all(np.abs(out.astype(np.float64) - ref) <= 1e-5 * abs_scale(rowptr, col, vals, x) + 1e-30)
        finally:
            for k, v in old.items():
                _lib.set_tunable(k, v)


@pytest.mark.parametrize("dt", ["INT32", "FLT32", "INT8"])
def test_degenerate_shapes(dt):
    """no stored entries at all (CSR and COO), a 1 x 1 matrix, one row / one column, widths 1 and 5: the result is what the
    oracle's loops give (zeros where nothing is stored), for the sweep as for the SpMV end"""
    npdt = NP_DTYPES[dt]
    for nrows, ncols, entries in ((7, 5, []), (1, 1, [(0, 0)]), (1, 9, [(0, 3), (0, 3), (0, 8)]), (6, 1, [(2, 0), (5, 0)])):
        rows = np.array([e[0] for e in entries], dtype=np.int32)
        cols = np.array([e[1] for e in entries], dtype=np.int32)
        rowptr = np.zeros(nrows + 1, dtype=np.int32)
        np.add.at(rowptr, rows + 1, 1)
        rowptr = np.cumsum(rowptr).astype(np.int32)
        for h in (1, 5):
            x = (np.arange(ncols * h).reshape(ncols, h) % 7 - 3).astype(npdt)
            ref = oracle.spmm_csr(rowptr, cols, None, x)
            out, _ = run_group_host("CSR", [rowptr], [cols], None, [nrows], [ncols], [x], h)
            assert out.shape == (nrows, h) and np.array_equal(out, ref), (dt, nrows, ncols, h, "csr")
            if len(entries) == len(set(entries)):   # COO input is coalesced
                vals = np.ones(len(entries), dtype=npdt)
                out, _ = run_group_host("COO", [rows], [cols], [vals], [nrows], [ncols], [x], h)
                assert np.array_equal(out, oracle.spmm_coo(rows, cols, vals, x, nrows)), (dt, nrows, ncols, h, "coo")


@pytest.mark.parametrize("dt", ["DBL64", "INT64"])
@pytest.mark.parametrize("kind", ["exact", "inexact", "one_wide", "specials"])
def test_eight_byte_values_streamed_in_four_bytes_when_exact(rng, dt, kind):
    """INT64 / DBL64 values that all survive the round trip through int32 / float are streamed narrow by the sweep
    (kernels.hpp SweepVal, narrow_values at group creation); one value that does not, and the group keeps the 8-byte stream.
    Either way the product is bit for bit the one with the knob off, and the oracle's (integers: exactly)."""
    npdt = NP_DTYPES[dt]
    nrows, ncols, h = 400, 700, 80
    rowptr, col = random_csr(rng, nrows, ncols, 30, empty_frac=0.1, long_rows=[(5, 2000)])
    if dt == "DBL64":
        vals = (rng.random(len(col)) * 2 - 1).astype(np.float32).astype(np.float64)
        x = (rng.random((ncols, h)) * 2 - 1).astype(np.float64)
        if kind == "inexact":
            vals = rng.random(len(col)) * 2 - 1
        elif kind == "one_wide":
            vals[len(vals) // 2] = 0.1
        elif kind == "specials":
            vals[:4] = [-0.0, np.inf, 2.0 ** -140, 3.0e38]   # (a float32 denormal and a near-max value: both exact)
            x[col[1]] = np.abs(x[col[1]]) + 0.5             # inf * positive: no NaN from 0 * inf
    else:
        vals = rng.integers(-1000, 1000, size=len(col)).astype(np.int64)
        x = driver_features(rng, ncols, h, npdt)
        if kind == "inexact":
            vals = rng.integers(-2 ** 40, 2 ** 40, size=len(col)).astype(np.int64)
        elif kind == "one_wide":
            vals[len(vals) // 2] = 2 ** 31
        elif kind == "specials":
            vals[:4] = [-2 ** 31, 2 ** 31 - 1, 0, -1]
    ref = oracle.spmm_csr(rowptr, col, vals, x)
    outs = {}
    for narrow in (1, 0):
        old = {"narrow_vals": _lib.set_tunable("narrow_vals", narrow), "panel_mode": _lib.set_tunable("panel_mode", 1)}
        try:
            outs[narrow], info = run_group_host("CSR", [rowptr], [col], [vals], [nrows], [ncols], [x], h)
        finally:
            for k, v in old.items():
                _lib.set_tunable(k, v)
        assert info["all_ones"] == 0
        if dt == "INT64":
            assert np.array_equal(outs[narrow], ref), (dt, kind, narrow)
    # the widening conversion is exact: the narrow stream changes no bit of the result (floats: same kernel, same order)
    assert np.array_equal(outs[1].view(np.int64), outs[0].view(np.int64)), (dt, kind)
    if dt == "DBL64":
        # (the 2 000-entry row is summed in segments: not the oracle's order, so a bound instead of bits)
        scale = oracle.spmm_csr(rowptr, col, np.abs(vals), np.abs(x))
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(outs[1]), fin) and np.array_equal(outs[1][~fin], ref[~fin])
        assert np.all(np.abs(outs[1][fin] - ref[fin]) <= 1e-13 * scale[fin] + 1e-300), (dt, kind)
