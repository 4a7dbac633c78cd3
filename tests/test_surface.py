"""The reference's Python surface on CPU: SparseTensor stand-in, partition helpers and the three
backend_pim wrappers driven end to end through torch.ops.pim_ops with a test double for the C ABI
(tests/fake_abi.py, oracle inside).  Mirrors how spmm_test.py drives the reference."""
import types

import numpy as np
import pytest
import torch

import oracle
from conftest import ALL_DTYPES, NP_DTYPES, random_csr
from fake_abi import FakeLib
from pygim_amd import pim_ops
from pygim_amd.backend_pim import grande as grande_mod
from pygim_amd.backend_pim import spmm as spmm_mod
from pygim_amd.backend_pim import spmv as spmv_mod
from pygim_amd.sparse_tensor import SparseTensorShim, _shim_matmul

TORCH_OF = {"INT8": torch.int8, "INT16": torch.int16, "INT32": torch.int32, "INT64": torch.int64,
            "FLT32": torch.float32, "DBL64": torch.float64}


@pytest.fixture
def fake(monkeypatch):
    f = FakeLib()
    monkeypatch.setattr(pim_ops, "_lib", f)
    monkeypatch.setattr(pim_ops, "_variant", None)
    yield f
    if pim_ops._library is not None:
        assert pim_ops._library != "native", "a test loaded the C++ shim in the pytest process"
        pim_ops._library._destroy()
        pim_ops._library = None
    pim_ops._variant = None
    pim_ops._groups.clear()


def make_adj(rng, n=150, deg=7, value=False):
    rowptr, col = random_csr(rng, n, n, deg)
    val = torch.from_numpy(rng.integers(1, 4, size=len(col)).astype(np.float32)) if value else None
    adj = SparseTensorShim(rowptr=torch.from_numpy(rowptr), col=torch.from_numpy(col), value=val, sparse_sizes=(n, n))
    return adj, rowptr, col


def args_of(**kw):
    return types.SimpleNamespace(**kw)


def test_sparse_tensor_shim_basics(rng):
    adj, rowptr, col = make_adj(rng)
    rp, c, v = adj.csr()
    assert v is None and rp.dtype == torch.int64 and np.array_equal(rp.numpy(), rowptr)
    assert adj.nnz() == len(col) and adj.size(0) == 150 and adj.sizes() == [150, 150]
    sub = adj[:, 40:90]
    assert sub.sizes() == [150, 50] and int(sub.coo()[1].max()) < 50
    dense = adj.to_dense(torch.float32)
    assert torch.equal(sub.to_dense(torch.float32), dense[:, 40:90])
    x = torch.randn(150, 4)
    assert torch.allclose(_shim_matmul(adj, x), dense @ x, atol=1e-4)
    assert callable(adj.dtype)  # a method, as on torch_sparse.SparseTensor


def test_dense_split_semantics():
    b = torch.arange(40).reshape(4, 10)
    parts = spmm_mod.dense_split(b, 3)
    assert [p.shape[1] for p in parts] == [4, 4, 2] and all(p.is_contiguous() for p in parts)
    assert spmm_mod.dense_split(b, 1)[0] is not None
    # grande: windows padded to 8 bytes, starting at the running sum of the true widths
    b32 = torch.arange(4 * 7, dtype=torch.int32).reshape(4, 7)
    wins = grande_mod.dense_split(b32, torch.tensor([3, 2, 2], dtype=torch.int32))
    assert [w.shape[1] for w in wins] == [4, 4, 4]
    assert torch.equal(wins[1][:, :2], b32[:, 3:5]) and torch.equal(wins[2][:, :2], b32[:, 5:7])


@pytest.mark.parametrize("fmt", ["CSR", "COO"])
@pytest.mark.parametrize("sp_parts,ds_parts", [(1, 1), (4, 1), (3, 4), (32, 1)])
def test_spmm_wrapper_matches_cpu_path(rng, fake, fmt, sp_parts, ds_parts):
    pim_ops.load("spmm")
    adj, rowptr, col = make_adj(rng)
    h = 20
    torch.ops.pim_ops.dpu_init_ranks(sp_parts * ds_parts)
    a = args_of(data_type=torch.int32, sp_format=fmt, sp_parts=sp_parts, ds_parts=ds_parts, hidden_size=h)
    A = spmm_mod.prepare_pim_spmm(adj, a)
    assert len(A.parts) == sp_parts
    x = torch.randint(-8, 4, (150, h), dtype=torch.int32)
    out = spmm_mod.pim_spmm(x, A)
    ref = oracle.spmm_csr(rowptr, col, None, x.numpy())
    assert out.dtype == torch.int32 and np.array_equal(out.numpy(), ref)
    with pytest.raises(AssertionError):
        A.mul(torch.zeros(150, h + 1, dtype=torch.int32))
    with pytest.raises(RuntimeError):  # dtype mismatch surfaces as an error, like data_ptr<val_dt>()
        A.mul(torch.zeros(150, h, dtype=torch.int64))
    torch.ops.pim_ops.dpu_release()


@pytest.mark.parametrize("dt", ALL_DTYPES)
def test_spmm_wrapper_all_dtypes_and_edge_values(rng, fake, dt):
    pim_ops.load("spmm")
    adj, rowptr, col = make_adj(rng, value=True)
    tdt = TORCH_OF[dt]
    torch.ops.pim_ops.dpu_init_ranks(2)
    a = args_of(data_type=tdt, sp_format="CSR", sp_parts=2, ds_parts=1, hidden_size=8)
    A = spmm_mod.prepare_pim_spmm(adj, a)
    x = torch.randint(-8, 4, (150, 8)).to(tdt)
    vals = adj.storage.value().to(tdt).numpy()
    ref = oracle.spmm_csr(rowptr, col, vals, x.numpy())
    assert np.array_equal(A.mul(x).numpy(), ref)


def test_grande_wrapper(rng, fake):
    pim_ops.load("grande")
    adj, rowptr, col = make_adj(rng)
    dpus_per_rank = torch.ops.pim_ops.dpu_init_ranks(3)
    assert list(dpus_per_rank) == [8, 8, 8]
    h = 21  # not divisible by 8 windows: widths 3,3,3,3,3,2,2,2
    a = args_of(data_type=torch.int16, sp_format="CSR", sp_parts=3, hidden_size=h)
    A = grande_mod.prepare_pim_spmm_grande(adj, a, dpus_per_rank)
    assert A.dense_ncols[0].tolist() == [3, 3, 3, 3, 3, 2, 2, 2]
    x = torch.randint(-8, 4, (150, h), dtype=torch.int16)
    ref = oracle.spmm_csr(rowptr, col, None, x.numpy())
    assert np.array_equal(grande_mod.pim_spmm_grande(x, A).numpy(), ref)


def test_spmv_wrapper(rng, fake):
    pim_ops.load("spmv")
    adj, rowptr, col = make_adj(rng, n=149)  # not a multiple of 64/bits -> padded matrix
    torch.ops.pim_ops.dpu_init_ranks(4)
    a = args_of(data_type=torch.int32, sp_format="COO", sp_parts=1, ds_parts=4, hidden_size=12)
    A = spmv_mod.prepare_pim_spmv(adj, a)
    assert A.coo[0].size(0) == 150
    x = torch.randint(-8, 4, (149, 12), dtype=torch.int32)
    out = spmv_mod.pim_spmv(x, A)
    assert out.shape == (149, 12)
    assert np.array_equal(out.numpy(), oracle.spmm_csr(rowptr, col, None, x.numpy()))
    with pytest.raises(AssertionError):
        spmv_mod.prepare_pim_spmv(adj, args_of(data_type=torch.int32, sp_format="CSR", sp_parts=1, ds_parts=4))


def test_variant_schemas(fake):
    pim_ops.load("spmm")
    assert torch.ops.pim_ops.dpu_init_ranks(2) is None
    pim_ops.load_library("/not/built/backend_pim/spmm_grande/build/libbackend_pim.so")  # absent file -> Python registration
    assert pim_ops.current_variant() == "grande" and list(torch.ops.pim_ops.dpu_init_ranks(2)) == [8, 8]
    pim_ops.load_library("/not/built/backend_pim/spmv_sparseP/build/libbackend_pim.so")
    assert pim_ops.current_variant() == "spmv" and hasattr(torch.ops.pim_ops, "spmv_coo_run_group")


def test_quantiser_matches_oracle_restatement(rng):
    """pygim_amd.quantize (torch) vs the oracle's numpy restatement of models/quantize.py:20-42"""
    from pygim_amd import quantize as qz

    x = torch.from_numpy(rng.standard_normal((300, 40)).astype(np.float32))
    for tdt, npdt in ((torch.int8, np.int8), (torch.int16, np.int16), (torch.int32, np.int32), (torch.float32, np.float32)):
        scale, xq = qz.symmetric_quantize(x, dtype=tdt)
        s_ref, q_ref = oracle.symmetric_quantize(x.numpy(), npdt)
        assert np.float32(scale.item()) == s_ref and np.array_equal(xq.numpy(), q_ref)
    # the cpu path hands the quantiser SparseTensor.dtype (a bound method): float branch
    adj, rowptr, col = make_adj(rng)
    scale, xq = qz.symmetric_quantize(x[:150], dtype=adj.dtype)
    assert xq.dtype == torch.float32
    out = qz.message_and_aggregate(adj, x[:150])
    ref = oracle.symmetric_dequantize(oracle.spmm_csr(rowptr, col, None, xq.numpy()), 1.0, np.float32(scale.item()))
    assert np.allclose(out.numpy(), ref, rtol=1e-6, atol=1e-6)


def test_partition_chooser():
    from pygim_amd import autotune as at

    # Reddit-shaped, one GPU: an on-chip-blocked kernel is chosen (the LDS-staged code-stream form) and priced near the measured 2.07 ms of round 4
    best, table = at.choose(232965, 232965, 114615892, 256, 4, 1)
    assert best.panel and 1.7e-3 < best.seconds < 2.6e-3 and len(table) == 1
    # 8 GPUs: every divisor grid is priced; low-degree graph (products-shaped) never uses panels
    best8, table8 = at.choose(232965, 232965, 114615892, 256, 4, 8)
    assert {(c.row_parts, c.feat_parts) for c in table8} == {(1, 8), (2, 4), (4, 2), (8, 1)}
    assert best8.seconds < best.seconds
    _, t = at.choose(2449029, 2449029, 123718280, 256, 4, 1)
    assert not t[0].panel
    cfg = at.autotune(232965, 232965, 114615892, 256)
    assert len(cfg) == 5 and cfg[0] * cfg[1] == 8


def _write_mtx(path, nrows, ncols, rows, cols, vals=None, comments=2):
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        for k in range(comments):
            f.write("% comment line {}\n".format(k))
        f.write("{} {} {}\n".format(nrows, ncols, len(rows)))
        for k, (r, c) in enumerate(zip(rows, cols)):
            v = 1.0 if vals is None else vals[k]
            f.write("{} {} {}\n".format(r + 1, c + 1, v))


def _restated_reader(path, nrows, ncols, rows, cols):
    """what utils.hpp:15-127 produces: shape padded to even, ones, file order kept inside each row"""
    nr, nc = nrows + nrows % 2, ncols + ncols % 2
    order = np.argsort(np.asarray(rows), kind="stable")
    rowptr = np.zeros(nr + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=nr), out=rowptr[1:])
    return nr, nc, rowptr, np.asarray(cols)[order]


@pytest.mark.parametrize("shape", [(30, 20), (31, 17), (7, 7)])
def test_matrix_market_debug_ops(fake, tmp_path, shape):
    """read_matrix_* of the default variant (spmm_default/utils.hpp:139-173): .mtx -> int32 CSR arrays with the
    reference reader's semantics (even-padded shape, value column ignored = ones, unsorted columns kept in file
    order) and, where oracle/_ref/libref_utils.so exists, equal to the reference reader itself."""
    import oracle

    rng = np.random.default_rng(shape[0])
    nnz = 5 * shape[0]
    rows, cols = rng.integers(0, shape[0], nnz), rng.integers(0, shape[1], nnz)
    path = str(tmp_path / "tiny.mtx")
    _write_mtx(path, shape[0], shape[1], rows, cols, vals=rng.integers(2, 9, nnz))
    pim_ops.load("spmm")
    nr, nc, rowptr, colind = _restated_reader(path, shape[0], shape[1], rows, cols)
    ops = torch.ops.pim_ops
    assert ops.read_matrix_nrows(path) == nr and ops.read_matrix_ncols(path) == nc
    got_ptr, got_col, got_val = ops.read_matrix_rowptr(path), ops.read_matrix_colind(path), ops.read_matrix_values(path)
    assert got_ptr.dtype == got_col.dtype == got_val.dtype == torch.int32
    assert got_ptr.tolist() == rowptr.tolist() and got_col.tolist() == colind.tolist()
    assert got_val.tolist() == [1] * nnz
    if oracle.have_ref_utils():
        rn, rc, rptr, rcol, rval = oracle.ref_read_matrix_csr(path)
        assert (rn, rc) == (nr, nc)
        assert rptr.tolist() == got_ptr.tolist() and rcol.tolist() == got_col.tolist() and rval.tolist() == got_val.tolist()
