"""GPU: similarity tiles (pygim_amd/csrc/lds_reorder_dev.hpp, round 5) -- rows ordered by label propagation instead of by index.

The reference hands each DPU a range of CONSECUTIVE rows (support/partition.c:51-99).  Which rows share a tile changes nothing in the result
(every row is summed by one wave in stored order, whichever tile holds it): the products below must be the oracle's bit for bit, floats
included, and the device-generated code stream must still be the host encoder's word for word (lds_codegen = 2) for the same row order.
"""
import numpy as np
import pytest
import torch

import oracle
from pygim_amd import _lib, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def backend():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.init_ranks(1)
    yield
    _lib.release()


@pytest.fixture()
def forced():
    old = _lib.set_tunable("lds_mode", 1), _lib.set_tunable("lds_tile_order", 1), _lib.set_tunable("lds_codegen", 2), _lib.set_tunable("lds_col_split", 1)
    yield
    for k, v in zip(("lds_mode", "lds_tile_order", "lds_codegen", "lds_col_split"), old):
        _lib.set_tunable(k, v)


def _product(rowptr, col, x, code):
    n = rowptr.numel() - 1
    hd = _lib.group_create(_lib.CSR, code, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [col.numel()], [1], [x.shape[1]], x.shape[1])
    try:
        info = _lib.group_lds_tiles(hd), _lib.group_lds_code(hd), _lib.group_lds_geometry(hd), _lib.group_lds_plan(hd)
        out = torch.empty((n, x.shape[1]), dtype=x.dtype, device=x.device)
        _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
        torch.cuda.synchronize()
    finally:
        _lib.group_free(hd)
    return out.cpu().numpy(), info


@pytest.mark.parametrize("kind", ["sbm", "rmat", None])
@pytest.mark.parametrize("name,h", [("products-mini", 100), ("reddit-mini", 256)])
def test_similarity_tiles_give_the_oracles_bits(forced, kind, name, h):
    dev = torch.device("cuda", 0)
    rowptr, col = synth.make_shape(name, seed=3, device=dev, kind=kind)
    n = rowptr.numel() - 1
    for dt, code in ((torch.float32, _lib.FLT32), (torch.int32, _lib.INT32)):
        x = synth.features(n, h, dt, seed=5, device=dev, kind="uniform") if dt == torch.float32 else synth.features(n, h, dt, seed=5, device=dev)
        got, (tiles, code_info, geo, plan) = _product(rowptr, col, x, code)
        assert tiles["similarity"] == 1 and tiles["labels"] >= 1 and code_info["active"] == 1 and code_info["device_generated"] == 1, (tiles, code_info)
        want = oracle.spmm_csr(rowptr.cpu().numpy(), col.cpu().numpy(), None, x.cpu().numpy())
        assert got.tobytes() == want.tobytes(), (kind, name, dt)


def test_the_propagation_finds_planted_communities_and_tiles_share_reads(forced):
    """products-mini as a stochastic block model with shuffled ids (40 communities of 500 nodes, 80 % of a row's entries inside its own):
    label propagation ends with about one label per community, and tiles built from it serve far more entries from a shared LDS read than
    tiles of consecutive (= random) rows do"""
    dev = torch.device("cuda", 0)
    rowptr, col = synth.make_shape("products-mini", seed=1, device=dev, kind="sbm")
    n = rowptr.numel() - 1
    x = synth.features(n, 128, torch.int32, seed=2, device=dev)
    shared = {}
    for order in (0, 1):
        _lib.set_tunable("lds_tile_order", order)
        got, (tiles, code_info, geo, plan) = _product(rowptr, col, x, _lib.INT32)
        assert tiles["similarity"] == order
        shared[order] = geo["shared_entries"] / plan["nnz"]
        if order:
            assert 30 <= tiles["labels"] <= 80 and tiles["largest_label_rows"] <= 1500, tiles
        assert got.tobytes() == oracle.spmm_csr(rowptr.cpu().numpy(), col.cpu().numpy(), None, x.cpu().numpy()).tobytes()
    assert shared[1] > 2 * shared[0] and shared[1] > 0.3, shared


@pytest.mark.parametrize("kind", ["sbm", "clustered", None])
def test_sweep_items_in_locality_order_give_the_same_product(kind):
    """the L2 sweep with its work items in locality order (panel_locality: blocks of the propagated / id order behind the wave-cooperative
    prefix) -- the same rows, the same stored order inside each: integers exact, floats bit-identical wherever they were before"""
    dev = torch.device("cuda", 0)
    rowptr, col = synth.make_shape("products-mini", seed=4, device=dev, kind=kind)
    n = rowptr.numel() - 1
    old = _lib.set_tunable("lds_mode", 2), _lib.set_tunable("panel_locality", 2)
    try:
        for dt, code in ((torch.int32, _lib.INT32), (torch.float32, _lib.FLT32)):
            x = synth.features(n, 100, dt, seed=6, device=dev)
            got, (tiles, code_info, geo, plan) = _product(rowptr, col, x, code)
            assert plan["tiles"] == 0 and tiles["sweep_locality"] in (1, 2), (tiles, plan)
            assert got.tobytes() == oracle.spmm_csr(rowptr.cpu().numpy(), col.cpu().numpy(), None, x.cpu().numpy()).tobytes(), (kind, dt)
    finally:
        _lib.set_tunable("lds_mode", old[0])
        _lib.set_tunable("panel_locality", old[1])


@pytest.mark.parametrize("kind", ["sbm", "clustered"])
def test_narrow_blocks_on_a_group_whose_sweep_items_are_in_locality_order(kind):
    """a group created for WIDE products (h = 100) gets its sweep items in locality order -- no longer longest-first inside a
    panel -- while the LDS-staged SpMV kernel hands out lane groups by length class from prefixes of a length-sorted list.  A narrow call
    (pygim_block_run, 1..4 features) on such a group must not take that kernel: every width against the oracle, exact."""
    dev = torch.device("cuda", 0)
    rowptr, col = synth.make_shape("products-mini", seed=9, device=dev, kind=kind)
    n = rowptr.numel() - 1
    # (long rows too: the classes above 32 / 64 / 128 / 256 entries are the ones a misplaced item would be truncated in)
    old = _lib.set_tunable("lds_mode", 2), _lib.set_tunable("panel_locality", 2)
    try:
        for vals in (False, True):
            v = torch.randint(-3, 4, (col.numel(),), dtype=torch.int32, device=dev) if vals else None
            hd = _lib.group_create(_lib.CSR, _lib.INT32, [rowptr.data_ptr()], [col.data_ptr()], [v.data_ptr()] if vals else None, [n], [n], [col.numel()], [1], [100], 100)
            try:
                assert _lib.group_lds_tiles(hd)["sweep_locality"] in (1, 2)
                for w in (1, 2, 3, 4):
                    x = synth.features(n, w, torch.int32, seed=10 + w, device=dev)
                    out = torch.empty((n, w), dtype=torch.int32, device=dev)
                    _lib.block_run(hd, 0, x.data_ptr(), w, out.data_ptr(), w, w, 0, 0)
                    torch.cuda.synchronize()
                    want = oracle.spmm_csr(rowptr.cpu().numpy(), col.cpu().numpy(), v.cpu().numpy() if vals else None, x.cpu().numpy())
                    assert np.array_equal(out.cpu().numpy(), want), (kind, vals, w)
            finally:
                _lib.group_free(hd)
    finally:
        _lib.set_tunable("lds_mode", old[0])
        _lib.set_tunable("panel_locality", old[1])
