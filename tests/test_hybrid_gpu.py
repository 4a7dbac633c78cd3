"""GPU: the density split (pygim_amd/csrc/lds_hybrid_dev.hpp, round 5) -- a community-structured part with shuffled ids as
A_dense (the (tile, chunk) cells that hold many entries: LDS-staged product over a copy of X in the propagation's order) + A_sparse (the rest: the
L2 sweep, adding).  The reference deals consecutive rows to its DPUs (support/partition.c:51-99) and has nothing of the kind; what must hold is the
result: integers bit-exact against the oracle (sums reordered, arithmetic modular), floats inside the north star's 1e-5 (the order of a row's
products changes, which is why floats take the split only when asked: lds_hybrid = 2).
"""
import numpy as np
import pytest
import torch

import oracle
from pygim_amd import _lib, synth

pytestmark = pytest.mark.gpu

TUNE = {"lds_mode": 0, "lds_min_reuse_x100": 10 ** 6, "panel_locality": 2, "lds_hybrid_min": 32, "lds_codegen": 2}   # (no whole-part LDS plan; propagation whatever the size)


@pytest.fixture(scope="module", autouse=True)
def backend():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.init_ranks(1)
    yield
    _lib.release()


@pytest.fixture()
def split_forced():
    old = {k: _lib.set_tunable(k, v) for k, v in TUNE.items()}
    yield
    for k, v in old.items():
        _lib.set_tunable(k, v)
    _lib.set_tunable("lds_hybrid", 1)


def _graph(dev, n=36_000, deg=40, blocks=30, seed=11):
    return synth.make_sbm(n, n * deg, 2_000, blocks, p_in=0.8, seed=seed, device=dev, shuffle=True)


def _run(fmt, code, rowptr, col, x, hybrid, vals=None, twice=False):
    n = rowptr.numel() - 1
    _lib.set_tunable("lds_hybrid", hybrid)
    if fmt == _lib.COO:
        row = torch.repeat_interleave(torch.arange(n, device=col.device, dtype=torch.int32), (rowptr[1:] - rowptr[:-1]))
        idx0 = row
    else:
        idx0 = rowptr
    hd = _lib.group_create(fmt, code, [idx0.data_ptr()], [col.data_ptr()], None if vals is None else [vals.data_ptr()], [n], [n], [col.numel()], [1], [x.shape[1]], x.shape[1])
    try:
        note = _lib.group_lds_note(hd)
        out = torch.full((n, x.shape[1]), 77, dtype=x.dtype, device=x.device)
        _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
        if twice:   # the caller vouches that X did not change: the staged copy (in the split's order) is taken again
            out.fill_(55)
            _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0, x_unchanged=True)
        torch.cuda.synchronize()
        runs = _lib.group_lds_runs(hd)
    finally:
        _lib.group_free(hd)
    return out.cpu().numpy(), note, runs


@pytest.mark.parametrize("dt,code,h", [(torch.int32, _lib.INT32, 256), (torch.int32, _lib.INT32, 100), (torch.int16, _lib.INT16, 256)])
def test_split_product_is_the_oracles_for_integers(split_forced, dt, code, h):
    dev = torch.device("cuda", 0)
    rowptr, col = _graph(dev)
    n = rowptr.numel() - 1
    ii = torch.iinfo(dt)
    x = torch.randint(ii.min, ii.max, (n, h), device=dev, dtype=torch.int64, generator=torch.Generator(device=dev).manual_seed(5)).to(dt)
    want = oracle.spmm_csr(rowptr.cpu().numpy(), col.cpu().numpy(), None, x.cpu().numpy())
    got, note, runs = _run(_lib.CSR, code, rowptr, col, x, 1, twice=True)
    assert "density split" in note and runs == 2, (note, runs)
    assert got.tobytes() == want.tobytes()
    got0, note0, runs0 = _run(_lib.CSR, code, rowptr, col, x, 0)
    assert "density split" not in note0 and runs0 == 0, (note0, runs0)
    assert got0.tobytes() == want.tobytes()


def test_split_of_a_coalesced_coo_multigraph(split_forced):
    """COO as the reference builds it (coalesce(): duplicates become values > 1, backend_pim/spmm.py:40-42): the pattern is split by density, the few
    weights other than 1 ride the correction part as before"""
    dev = torch.device("cuda", 0)
    rowptr, col = _graph(dev, blocks=6, seed=12)   # (large communities: few duplicate edges, so the weights other than 1 fit the correction part)
    n = rowptr.numel() - 1
    r, c, v = synth.csr_to_coo_coalesced(rowptr, col, torch.int32)
    rp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rp[1:] = torch.cumsum(torch.bincount(r.long(), minlength=n), 0)
    x = synth.features(n, 128, torch.int32, seed=6, device=dev)
    got, note, runs = _run(_lib.COO, _lib.INT32, rp.to(torch.int32), c, x, 1, vals=v)
    assert "density split" in note and runs >= 1, (note, runs)
    want = oracle.spmm_csr(rowptr.cpu().numpy(), col.cpu().numpy(), None, x.cpu().numpy())   # (the multigraph with unit weights = its coalesced form)
    assert got.tobytes() == want.tobytes()


def test_floats_take_the_split_only_when_asked_and_stay_inside_the_tolerance(split_forced):
    dev = torch.device("cuda", 0)
    rowptr, col = _graph(dev, seed=13)
    n = rowptr.numel() - 1
    x = synth.features(n, 256, torch.float32, seed=7, device=dev, kind="uniform")
    want = oracle.spmm_csr(rowptr.cpu().numpy(), col.cpu().numpy(), None, x.cpu().numpy())
    got1, note1, runs1 = _run(_lib.CSR, _lib.FLT32, rowptr, col, x, 1)
    assert "density split" not in note1 and runs1 == 0                                # default: the part's own plan (the sweep: long rows are summed by a whole wave)
    assert np.max(np.abs(got1 - want)) <= 1e-5 * np.max(np.abs(want))
    got2, note2, runs2 = _run(_lib.CSR, _lib.FLT32, rowptr, col, x, 2)
    assert "density split" in note2 and runs2 >= 1, (note2, runs2)
    assert np.max(np.abs(got2 - want)) <= 1e-5 * np.max(np.abs(want))                # north star: 1e-5 relative for FLT32


def test_calls_the_lds_kernel_does_not_take_keep_the_parts_own_plan(split_forced):
    """an odd block of columns (pygim_block_run with a width below the LDS kernel's) and an accumulating call run the part's own sweep"""
    dev = torch.device("cuda", 0)
    rowptr, col = _graph(dev, seed=14)
    n = rowptr.numel() - 1
    x = synth.features(n, 256, torch.int32, seed=8, device=dev)
    _lib.set_tunable("lds_hybrid", 1)
    hd = _lib.group_create(_lib.CSR, _lib.INT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [col.numel()], [1], [256], 256)
    try:
        assert "density split" in _lib.group_lds_note(hd)
        out = torch.full((n, 256), 3, dtype=torch.int32, device=dev)
        _lib.block_run(hd, 0, x.data_ptr(), 256, out.data_ptr(), 256, 4, accumulate=False)           # 4 columns: too narrow for the LDS kernel (lds_min_width = 5 lanes)
        _lib.block_run(hd, 0, x.data_ptr() + 4 * 64, 256, out.data_ptr() + 4 * 64, 256, 192, accumulate=True)    # accumulating: the part's own plan
        torch.cuda.synchronize()
        runs = _lib.group_lds_runs(hd)
    finally:
        _lib.group_free(hd)
    xc = x.cpu().numpy()
    full = oracle.spmm_csr(rowptr.cpu().numpy(), col.cpu().numpy(), None, xc)
    got = out.cpu().numpy()
    assert runs == 0
    assert np.array_equal(got[:, :4], full[:, :4])
    assert np.array_equal(got[:, 64:256], full[:, 64:256] + 3)
    assert np.all(got[:, 4:64] == 3)


def test_split_under_feature_windows(split_forced):
    """ds_parts = 2 (dense_split, backend_pim/spmm.py:9-13): the group runs one block product per feature window; each takes the split"""
    dev = torch.device("cuda", 0)
    rowptr, col = _graph(dev, seed=15)
    n = rowptr.numel() - 1
    x = synth.features(n, 256, torch.int32, seed=9, device=dev)
    xa, xb = x[:, :128].contiguous(), x[:, 128:].contiguous()
    _lib.set_tunable("lds_hybrid", 1)
    hd = _lib.group_create(_lib.CSR, _lib.INT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [col.numel()], [2], [128, 128], 256)
    try:
        assert "density split" in _lib.group_lds_note(hd)
        out = torch.full((n, 256), 9, dtype=torch.int32, device=dev)
        _lib.spmm_run_group(hd, [xa.data_ptr(), xb.data_ptr()], out.data_ptr(), 0)
        torch.cuda.synchronize()
        runs = _lib.group_lds_runs(hd)
    finally:
        _lib.group_free(hd)
    want = oracle.spmm_csr(rowptr.cpu().numpy(), col.cpu().numpy(), None, x.cpu().numpy())
    assert runs >= 1
    assert out.cpu().numpy().tobytes() == want.tobytes()
