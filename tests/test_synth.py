"""CPU: the seeded graph generators (pygim_amd/synth.py) -- what the GPU-side measurements of structured graphs stand on.
Shapes and degrees as asked, rows column-sorted (the CSR the reference's wrappers hand over, backend_pim/spmm.py:44-55), seeds reproducible,
the planted structure really there (SBM), the relabelling really a relabelling (shuffle_ids)."""
import numpy as np
import torch

from pygim_amd import synth


def _check_csr(rowptr, col, n, nnz):
    assert rowptr.dtype == torch.int32 and col.dtype == torch.int32
    assert rowptr.numel() == n + 1 and int(rowptr[0]) == 0 and int(rowptr[-1]) == nnz == col.numel()
    assert int(col.min()) >= 0 and int(col.max()) < n
    row = torch.repeat_interleave(torch.arange(n), (rowptr[1:] - rowptr[:-1]).long())
    key = row * n + col.long()
    assert bool((key[1:] >= key[:-1]).all()), "rows are not column-sorted"


def test_sbm_has_the_planted_communities_and_shuffling_hides_them():
    n, nnz, blocks = 6000, 6000 * 40, 12
    rp, ci = synth.make_sbm(n, nnz, 2000, blocks, p_in=0.8, seed=3, shuffle=False)
    _check_csr(rp, ci, n, nnz)
    size = (n + blocks - 1) // blocks
    row = torch.repeat_interleave(torch.arange(n), (rp[1:] - rp[:-1]).long())
    inside = float(((row // size) == (ci.long() // size)).float().mean())
    assert 0.78 < inside < 0.86, inside                     # 80 % inside + the background's own 1 / blocks share
    rs, cs = synth.make_sbm(n, nnz, 2000, blocks, p_in=0.8, seed=3, shuffle=True)
    _check_csr(rs, cs, n, nnz)
    rows = torch.repeat_interleave(torch.arange(n), (rs[1:] - rs[:-1]).long())
    assert float(((rows // size) == (cs.long() // size)).float().mean()) < 0.15      # ids carry no locality any more
    assert sorted((rp[1:] - rp[:-1]).tolist()) == sorted((rs[1:] - rs[:-1]).tolist())  # the same rows under new names
    again = synth.make_sbm(n, nnz, 2000, blocks, p_in=0.8, seed=3, shuffle=True)
    assert torch.equal(again[0], rs) and torch.equal(again[1], cs)


def test_rmat_is_skewed_on_both_sides():
    n, nnz = 5000, 5000 * 60
    rp, ci = synth.make_rmat(n, nnz, seed=1, shuffle=False)
    _check_csr(rp, ci, n, nnz)
    deg = (rp[1:] - rp[:-1]).double()
    indeg = torch.bincount(ci.long(), minlength=n).double()
    assert float(deg.max()) > 8 * float(deg.mean()) and float(indeg.max()) > 8 * float(indeg.mean())
    rs, cs = synth.make_rmat(n, nnz, seed=1, shuffle=True)
    _check_csr(rs, cs, n, nnz)
    assert sorted(deg.tolist()) == sorted((rs[1:] - rs[:-1]).double().tolist())


def test_shuffle_ids_is_a_relabelling():
    rp, ci = synth.make_shape("cora", seed=0)
    n = rp.numel() - 1
    rs, cs = synth.shuffle_ids(rp, ci, seed=5)
    _check_csr(rs, cs, n, ci.numel())
    # the product with an all-ones vector (row degrees) and with its transpose (column degrees) keeps its multiset
    assert sorted((rp[1:] - rp[:-1]).tolist()) == sorted((rs[1:] - rs[:-1]).tolist())
    assert sorted(torch.bincount(ci.long(), minlength=n).tolist()) == sorted(torch.bincount(cs.long(), minlength=n).tolist())


def test_make_shape_kinds():
    for kind in (None, "clustered", "sbm", "rmat"):
        rp, ci = synth.make_shape("reddit-mini", seed=0, kind=kind)
        n, nnz, _ = synth.SHAPES["reddit-mini"]
        _check_csr(rp, ci, n, nnz)
