"""The CPU oracle against independent libraries, the reference's own partition.c
(oracle/_ref) and the committed golden vectors.  No GPU needed."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import oracle
from conftest import ALL_DTYPES, NP_DTYPES, coalesce, driver_features, random_csr

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _dense_ref(rowptr, col, vals, x):
    """exact reference through scipy in int64/float64, reduced to the element type"""
    n = len(rowptr) - 1
    dt = x.dtype
    wide = np.int64 if np.issubdtype(dt, np.integer) else np.float64
    v = np.ones(len(col), dtype=wide) if vals is None else vals.astype(wide)
    a = sp.csr_matrix((v, col, rowptr), shape=(n, x.shape[0]))
    y = a @ x.astype(wide)
    return y.astype(dt) if np.issubdtype(dt, np.integer) else y


@pytest.mark.parametrize("dt", ALL_DTYPES)
def test_csr_matches_scipy_and_torch(rng, dt):
    npdt = NP_DTYPES[dt]
    rowptr, col = random_csr(rng, 300, 257, 9, long_rows=[(5, 700)])
    x = driver_features(rng, 257, 32, npdt)
    y = oracle.spmm_csr(rowptr, col, None, x)
    ref = _dense_ref(rowptr, col, None, x)
    if np.issubdtype(npdt, np.integer):
        assert np.array_equal(y, ref)  # int8 / int16 wrap: row 5 has 700 entries
    else:
        np.testing.assert_allclose(y, ref, rtol=1e-5)
        # driver features are small integers -> float sums are exact
        assert np.array_equal(y, ref.astype(npdt))
    # torch.sparse.mm on the coalesced COO form (the only torch CPU path with int support)
    r, c, v = coalesce(rowptr, col, npdt)
    a = torch.sparse_coo_tensor(torch.tensor(np.stack([r, c]).astype(np.int64)), torch.from_numpy(v), (300, 257))
    yt = torch.sparse.mm(a, torch.from_numpy(x)).numpy()
    assert np.array_equal(y, yt)
    # the oracle's own COO loop on the coalesced triples (values > 1 where edges repeat)
    yc = oracle.spmm_coo(r, c, v, x, 300)
    assert np.array_equal(y, yc)


@pytest.mark.parametrize("dt", ["INT32", "FLT32", "DBL64", "INT8"])
def test_valued_and_rowpar_bitwise(rng, dt):
    npdt = NP_DTYPES[dt]
    rowptr, col = random_csr(rng, 200, 180, 12)
    vals = rng.integers(-3, 4, size=len(col)).astype(npdt)
    if not np.issubdtype(npdt, np.integer):
        vals = (rng.random(len(col)) * 2 - 1).astype(npdt)
        x = (rng.random((180, 17)) * 2 - 1).astype(npdt)
    else:
        x = driver_features(rng, 180, 17, npdt)
    y = oracle.spmm_csr(rowptr, col, vals, x)
    ref = _dense_ref(rowptr, col, vals, x)
    if np.issubdtype(npdt, np.integer):
        assert np.array_equal(y, ref)
    else:
        np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-6 if npdt == np.float32 else 1e-12)
    for nt in (1, 3):
        assert np.array_equal(y, oracle.spmm_csr_rowpar(rowptr, col, vals, x, nthreads=nt))


@pytest.mark.parametrize("fmt", ["CSR", "COO"])
@pytest.mark.parametrize("sp_parts,ds_parts", [(1, 1), (2, 1), (3, 4), (8, 3)])
def test_group_semantics(rng, fmt, sp_parts, ds_parts):
    """sum over column blocks, concatenation over feature blocks (ops.hpp:42-62, spmm.py:9-13,127-136)"""
    npdt = np.int32
    n, h = 150, 10
    rowptr, col = random_csr(rng, n, n, 7)
    x = driver_features(rng, n, h, npdt)
    full = oracle.spmm_csr(rowptr, col, None, x)
    step = (n + sp_parts - 1) // sp_parts
    a = sp.csr_matrix((np.ones(len(col), dtype=np.int64), col, rowptr), shape=(n, n))
    idx0, cols, vals, nrows, ncols = [], [], [], [], []
    for i in range(sp_parts):
        blk = a[:, i * step:min(n, (i + 1) * step)].tocsr()
        blk.sum_duplicates()
        blk.sort_indices()
        if fmt == "CSR":
            idx0.append(blk.indptr)
        else:
            idx0.append(blk.tocoo().row)
        cols.append(blk.indices)
        vals.append(blk.data.astype(npdt))
        nrows.append(n)
        ncols.append(blk.shape[1])
    xs = [np.ascontiguousarray(c) for c in np.array_split(x, ds_parts, axis=1)] if ds_parts > 1 else [x]
    xs = [c for c in xs if c.shape[1] > 0]
    out = oracle.group(fmt == "COO", idx0, cols, vals, nrows, ncols, xs, h)
    assert np.array_equal(out, full)


def test_partition_restatement_matches_reference_build(rng):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (reference tree absent)")
    for trial in range(40):
        nrows = int(rng.integers(1, 400))
        rowptr, _ = random_csr(rng, nrows, 50, float(rng.uniform(0.2, 20)), empty_frac=0.3)
        for nparts in (1, 2, 3, 7, 16, 64):
            mine = oracle.partition_by_nnz(rowptr, nparts)
            ref = oracle.ref_partition_by_nnz_csr(rowptr, nparts)
            assert np.array_equal(mine, ref), (trial, nparts)
            hist = np.diff(rowptr.astype(np.int64)).astype(np.uint32)
            assert np.array_equal(mine, oracle.ref_partition_by_nnz_rgrn_coo(hist, nparts))
            assert np.array_equal(oracle.partition_by_row(nrows, nparts), oracle.ref_partition_by_row_csr(nrows, nparts))
            nnz = int(rowptr[-1])
            assert np.array_equal(oracle.partition_equal_nnz(nnz, nparts), oracle.ref_partition_tsklt_by_nnz_coo(nnz, nparts))


def _case_inputs(rng, npdt, nrows, ncols, h, valued, big):
    """random graph with a wrap-inducing long row; features / values large enough that INT8 / INT16 products wrap"""
    rowptr, col = random_csr(rng, nrows, ncols, float(rng.uniform(1, 14)), empty_frac=0.25,
                             long_rows=[(int(rng.integers(0, nrows)), int(rng.integers(300, 1200)))])
    if np.issubdtype(npdt, np.integer):
        lim = min(np.iinfo(npdt).max, 30000) if big else 8
        x = rng.integers(-lim, lim, size=(ncols, h)).astype(npdt)
        vals = rng.integers(-lim, lim, size=len(col)).astype(npdt) if valued else None
    else:
        x = ((rng.random((ncols, h)) * 2 - 1) * (1e3 if big else 1)).astype(npdt)
        vals = (rng.random(len(col)) * 2 - 1).astype(npdt) if valued else None
    return rowptr, col, vals, x


needs_ref_host = pytest.mark.skipif(not oracle.have_ref_host(),
                                    reason="oracle/_ref/libref_host_* not built (reference tree absent and no prebuilt copy)")


@needs_ref_host
@pytest.mark.parametrize("dt", ALL_DTYPES)
def test_oracle_is_bit_identical_to_the_reference_host_loops(dt):
    """THE PIN of the arithmetic: oracle.spmm_csr / spmm_coo against the reference's own spmm_host_csr (valued: grande,
    spmm_grande/spmm_mul_csr.c:119-136; unit weights: spmm_default/spmm_mul_csr.c:100-113), spmm_host_coo
    (spmm_default/spmm_mul_coo.c:40-51) and spmm_host (spmv_sparseP/spmv_mul_coo.c:92-103), compiled in place by
    oracle/build_ref_host.sh -- same bytes for all six types (INT8 / INT16 wrap included, floats not merely close)."""
    npdt = NP_DTYPES[dt]
    rng = np.random.default_rng(2024)
    for trial in range(24):
        nrows, ncols, h = int(rng.integers(1, 400)), int(rng.integers(1, 300)), int(rng.choice([1, 3, 9, 32, 65]))
        valued, big = bool(trial & 1), bool(trial & 2)
        rowptr, col, vals, x = _case_inputs(rng, npdt, nrows, ncols, h, valued, big)
        y = oracle.spmm_csr(rowptr, col, vals, x)
        assert y.tobytes() == oracle.ref_spmm_host_csr(rowptr, col, vals, x, variant="grande").tobytes(), (dt, trial)
        if not valued:
            assert y.tobytes() == oracle.ref_spmm_host_csr(rowptr, col, None, x, variant="default").tobytes(), (dt, trial)
        # padded X stride of grande (ncols_pad, spmm_grande/spmm_mul_csr.c:131)
        pad = h + int(rng.integers(1, 5))
        xp = np.zeros((ncols, pad), dtype=npdt)
        xp[:, :h] = x
        assert y.tobytes() == oracle.ref_spmm_host_csr(rowptr, col, vals, xp, variant="grande", ldx=(h, pad)).tobytes()
        # COO: coalesced triples (values > 1 where edges repeat), optionally re-weighted
        r, c, v = coalesce(rowptr, col, npdt)
        if valued:
            v = (v * (vals[: len(v)] if len(v) else v)).astype(npdt)
        yc = oracle.spmm_coo(r, c, v, x, nrows)
        assert yc.tobytes() == oracle.ref_spmm_host_coo(r, c, v, x, nrows, variant="default").tobytes(), (dt, trial)
        assert yc.tobytes() == oracle.ref_spmm_host_coo(r, c, v, x, nrows, variant="spmv").tobytes(), (dt, trial)
        # the row-parallel form bench.py times as cpu_baseline: same bits again
        assert y.tobytes() == oracle.spmm_csr_rowpar(rowptr, col, vals, x, nthreads=3).tobytes()


@needs_ref_host
def test_every_golden_vector_against_the_reference_build():
    """each committed spmm_* / group_* fixture is the byte-for-byte output of the reference build (re-checked wherever
    oracle/_ref travels: here and on the GPU box)"""
    for f in sorted(os.listdir(GOLDEN)):
        z = np.load(os.path.join(GOLDEN, f)) if f.endswith(".npz") else None
        if f.startswith("spmm_"):
            vals = z["vals"] if "vals" in z.files else None
            if str(z["fmt"]) == "CSR":
                y = oracle.ref_spmm_host_csr(z["rowptr"], z["col"], vals, z["x"], variant="grande")
            else:
                y = oracle.ref_spmm_host_coo(z["row"], z["col"], vals, z["x"], int(z["nrows"]))
            assert y.tobytes() == z["y"].tobytes(), f
        elif f.startswith("group_"):
            n = int(z["n_parts"])
            xs = [np.ascontiguousarray(c) for c in np.array_split(z["x"], int(z["ds_parts"]), axis=1) if c.shape[1] > 0]
            y = oracle.ref_group(str(z["fmt"]) == "COO", [z[f"idx0_{i}"] for i in range(n)], [z[f"col_{i}"] for i in range(n)],
                                 [z[f"vals_{i}"] for i in range(n)], [z["y"].shape[0]] * n, z["ncols"].tolist(), xs, z["y"].shape[1])
            assert y.tobytes() == z["y"].tobytes(), f


def test_no_fixture_is_unpinned():
    """every arithmetic fixture names the reference build that produced it"""
    seen = 0
    for f in sorted(os.listdir(GOLDEN)):
        if f.endswith(".npz") and (f.startswith("spmm_") or f.startswith("group_") or f.startswith("quant_")):
            pin = str(np.load(os.path.join(GOLDEN, f))["pinned_by"])
            assert pin.startswith("reference ") and "unpinned" not in pin, (f, pin)
            if f.startswith("quant_"):  # outputs of the reference's own quantiser and layer code, not of a restatement
                assert "models/quantize.py:20-42" in pin and "pyg_gcn_conv.py:130-137" in pin and "ast" in pin, (f, pin)
            seen += 1
    assert seen >= 49


@needs_ref_host
@pytest.mark.parametrize("dt", ALL_DTYPES)
def test_merge_helpers_against_the_reference(rng, dt):
    """add_2D / memadd_2D / memcpy_2D / matrix_add (spmm_default/spmm_mul_csr.c:41-86, spmv_sparseP/spmv_mul_coo.c:54-115):
    the oracle's add_2d and the plain numpy statement of the window operations give the reference's bytes"""
    npdt = NP_DTYPES[dt]
    draw = (lambda shape: rng.integers(-120, 120, size=shape).astype(npdt)) if np.issubdtype(npdt, np.integer) else \
        (lambda shape: (rng.random(shape) * 2 - 1).astype(npdt))
    for _ in range(10):
        R, C = int(rng.integers(4, 40)), int(rng.integers(4, 40))
        lx, ly = int(rng.integers(1, R)), int(rng.integers(1, C))
        ox, oy = int(rng.integers(0, R - lx + 1)), int(rng.integers(0, C - ly + 1))
        src_c = ly + int(rng.integers(0, 4))
        dest0, src = draw((R, C)), draw((lx, src_c))
        want_add = dest0.copy()
        want_add[ox:ox + lx, oy:oy + ly] += src[:, :ly]          # wraps at the element width like C
        want_cpy = dest0.copy()
        want_cpy[ox:ox + lx, oy:oy + ly] = src[:, :ly]
        for variant in ("default", "spmv"):
            assert oracle.ref_merge("add_2D", dest0.copy(), src, ox, oy, lx, ly, variant).tobytes() == want_add.tobytes()
            assert oracle.ref_merge("memadd_2D", dest0.copy(), src, ox, oy, lx, ly, variant).tobytes() == want_add.tobytes()
            assert oracle.ref_merge("memcpy_2D", dest0.copy(), src, ox, oy, lx, ly, variant).tobytes() == want_cpy.tobytes()
        assert oracle.ref_merge("add_2D", dest0.copy(), src, ox, oy, lx, ly, "grande").tobytes() == want_add.tobytes()
        assert oracle.add_2d(dest0.copy(), src, ox, oy, lx, ly).tobytes() == want_add.tobytes()
        b = draw((R, C))
        assert oracle.ref_matrix_add(dest0.copy(), b).tobytes() == (dest0 + b).astype(npdt).tobytes()


@needs_ref_host
@pytest.mark.parametrize("fmt", ["CSR", "COO"])
@pytest.mark.parametrize("dt", ["INT8", "INT32", "FLT32", "DBL64"])
def test_group_driver_against_the_reference(rng, fmt, dt):
    """oracle.group == spmm_host_csr_group / spmm_host_coo_group (spmm_default/ops.hpp:42-62,97-118) and, for COO,
    spmm_host_group (spmv_sparseP/spmv_mul_coo.c:128-148), on the reference's own structs"""
    npdt = NP_DTYPES[dt]
    n, h = 140, 21
    rowptr, col = random_csr(rng, n, n, 9)
    x = driver_features(rng, n, h, npdt) if np.issubdtype(npdt, np.integer) else (rng.random((n, h)) * 2 - 1).astype(npdt)
    a = sp.csr_matrix((np.ones(len(col), dtype=np.int64), col, rowptr), shape=(n, n))
    for sp_parts, ds_parts in ((1, 1), (2, 3), (5, 2), (8, 8)):
        step = (n + sp_parts - 1) // sp_parts
        idx0, cols, vals, ncols = [], [], [], []
        for i in range(sp_parts):
            blk = a[:, i * step:min(n, (i + 1) * step)].tocsr()
            blk.sum_duplicates()
            blk.sort_indices()
            idx0.append(blk.indptr if fmt == "CSR" else blk.tocoo().row)
            cols.append(blk.indices)
            vals.append(np.ones(blk.nnz, dtype=npdt) if fmt == "CSR" else blk.data.astype(npdt))
            ncols.append(blk.shape[1])
        xs = [np.ascontiguousarray(c) for c in np.array_split(x, ds_parts, axis=1) if c.shape[1] > 0]
        mine = oracle.group(fmt == "COO", idx0, cols, vals, [n] * sp_parts, ncols, xs, h)
        assert mine.tobytes() == oracle.ref_group(fmt == "COO", idx0, cols, vals, [n] * sp_parts, ncols, xs, h).tobytes()
        if fmt == "COO":
            assert mine.tobytes() == oracle.ref_group(True, idx0, cols, vals, [n] * sp_parts, ncols, xs, h, variant="spmv").tobytes()


def test_golden_vectors():
    files = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    assert files, "no golden vectors committed"
    for f in files:
        z = np.load(os.path.join(GOLDEN, f))
        if f.startswith("partition"):
            for k in range(int(z["n_cases"])):
                rp, nparts = z[f"rowptr_{k}"], int(z[f"nparts_{k}"])
                assert np.array_equal(oracle.partition_by_nnz(rp, nparts), z[f"by_nnz_{k}"])
                assert np.array_equal(oracle.partition_by_row(len(rp) - 1, nparts), z[f"by_row_{k}"])
            continue
        if f.startswith("group_"):
            n = int(z["n_parts"])
            xs = [np.ascontiguousarray(c) for c in np.array_split(z["x"], int(z["ds_parts"]), axis=1) if c.shape[1] > 0]
            y = oracle.group(str(z["fmt"]) == "COO", [z[f"idx0_{i}"] for i in range(n)], [z[f"col_{i}"] for i in range(n)],
                             [z[f"vals_{i}"] for i in range(n)], [z["y"].shape[0]] * n, z["ncols"].tolist(), xs, z["y"].shape[1])
            assert y.tobytes() == z["y"].tobytes(), f
            continue
        if not f.startswith("spmm_"):
            continue  # quant_gcn_* and mtx_ref have their own tests below
        vals = z["vals"] if "vals" in z.files else None
        if str(z["fmt"]) == "CSR":
            y = oracle.spmm_csr(z["rowptr"], z["col"], vals, z["x"])
        else:
            y = oracle.spmm_coo(z["row"], z["col"], vals, z["x"], int(z["nrows"]))
        assert y.tobytes() == z["y"].tobytes(), f  # floats too: same loop order, same rounding


@pytest.mark.parametrize("name", ["INT8", "INT16", "INT32", "FLT32"])
def test_quantiser_golden_vectors(name):
    """quantise -> aggregate -> dequantise of the conv layers with a fixed x.  The fixture holds outputs of the
    REFERENCE'S OWN symmetric_quantize / symmetric_dequantize / GCNConv.message_and_aggregate (cut out by name and executed by
    tests/golden/make_golden.py, `pinned_by`); the numpy restatement (the checker) and the torch statement (pygim_amd.quantize, the
    product's host side) both reproduce it"""
    from pygim_amd import quantize as qz

    z = np.load(os.path.join(GOLDEN, f"quant_gcn_{name}.npz"))
    npdt = NP_DTYPES[name]
    scale, xq = oracle.symmetric_quantize(z["x"], npdt)
    assert scale == z["scale"] and np.array_equal(xq, z["xq"])
    out_q = oracle.spmm_csr(z["rowptr"], z["col"], None, xq)
    assert np.array_equal(out_q, z["out_q"])
    assert np.array_equal(oracle.symmetric_dequantize(out_q, 1.0, scale), z["out"])
    tdt = {"INT8": torch.int8, "INT16": torch.int16, "INT32": torch.int32, "FLT32": torch.float32}[name]
    s_t, xq_t = qz.symmetric_quantize(torch.from_numpy(z["x"]), tdt)
    assert np.float32(s_t.item()) == z["scale"] and np.array_equal(xq_t.numpy(), z["xq"])
    assert np.array_equal(qz.symmetric_dequantize(torch.from_numpy(z["out_q"]), 1.0, s_t).numpy(), z["out"])
    assert str(z["pinned_by"]).startswith("reference symmetric_quantize")


def test_matrix_market_reader_against_reference_vectors(tmp_path):
    """tests/golden/mtx_ref.npz holds outputs of the reference's own readCOOMatrix + coo2csr (utils.hpp:15-127, built in
    place); the Python registration's reader (pim_ops._read_mtx_csr) reproduces them from the same file text"""
    from pygim_amd import pim_ops

    z = np.load(os.path.join(GOLDEN, "mtx_ref.npz"))
    for k in range(int(z["n_cases"])):
        path = tmp_path / f"case{k}.mtx"
        path.write_bytes(z[f"text_{k}"].tobytes())
        m = pim_ops._read_mtx_csr(str(path))
        assert list(m.shape) == z[f"shape_{k}"].tolist()
        assert np.array_equal(np.asarray(m.indptr), z[f"rowptr_{k}"]) and np.array_equal(np.asarray(m.indices), z[f"colind_{k}"])
        assert np.array_equal(np.asarray(m.data), z[f"values_{k}"])
        if oracle.have_ref_utils():
            n_, m_, rp_, ci_, va_ = oracle.ref_read_matrix_csr(str(path))
            assert [n_, m_] == z[f"shape_{k}"].tolist() and np.array_equal(rp_.astype(np.int32), z[f"rowptr_{k}"])
            assert np.array_equal(ci_.astype(np.int32), z[f"colind_{k}"]) and np.array_equal(va_, z[f"values_{k}"])
