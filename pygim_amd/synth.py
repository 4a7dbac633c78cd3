"""Seeded synthetic graphs in the shapes BASELINE.json names (no datasets on the GPU box).

Generator: out-degrees from a discretised log-normal clipped to
[0, d_max] and adjusted so that they add up to nnz exactly; column ids uniform in [0, N)
(or clustered around the row id), sorted inside each row, duplicates kept for CSR (the
reference's CSR path does not coalesce, backend_pim/spmm.py:44-55).  Adjacency values are
all ones (spmm.py:36-37, 48-49) and features follow the reference driver's generator
``torch.randint(-8, 4, (N, h))`` (spmm_test.py:70: ``-2^6, 2^6`` is XOR in Python).
Dataset statistics are the public PyG / OGB figures, pinned here as constants.
"""
from __future__ import annotations

import torch

# name -> (N, nnz, d_max)
SHAPES = {
    "cora": (2_708, 10_556, 168),
    "reddit": (232_965, 114_615_892, 21_657),
    "ogbn-products": (2_449_029, 123_718_280, 17_481),
    "ogbn-papers100M": (111_059_956, 1_615_685_872, 100_000),
    # small stand-ins with the same mean degree / skew, for tests
    "reddit-mini": (4_096, 4_096 * 123, 1_500),
    "products-mini": (20_000, 20_000 * 50, 4_000),
}
# (nodes, edges, max degree) of the datasets the reference drivers know by name (spmm_test.py:42-53)
DATASETS = {"PubMed": (19_717, 88_648, 171), "Reddit": SHAPES["reddit"], "Cora": SHAPES["cora"],
            "AmazonProducts": (500_000, 84_000_000, 30_000), "ogbn-arxiv": (169_343, 1_166_243, 13_161),
            "ogbn-proteins": (132_534, 79_122_504, 7_750), "ogbn-products": SHAPES["ogbn-products"]}


def _gen(seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def degrees(n, nnz, d_max, seed=0, device="cpu", sigma=1.0):
    """int64 out-degrees, log-normal shaped, each <= d_max, summing to nnz exactly."""
    assert nnz <= n * d_max, "d_max too small for this nnz"
    g = _gen(seed, device)
    z = torch.randn(n, generator=g, device=device, dtype=torch.float64)
    w = torch.exp(sigma * z)
    deg = torch.floor(w * (nnz / float(w.sum()))).clamp_(max=d_max).to(torch.int64)
    order = torch.randperm(n, generator=g, device=device)
    diff = int(nnz - int(deg.sum()))
    while diff != 0:
        room = (deg[order] < d_max) if diff > 0 else (deg[order] > 0)
        cand = order[room]
        k = min(abs(diff), int(cand.numel()))
        deg[cand[:k]] += 1 if diff > 0 else -1
        diff -= k if diff > 0 else -k
    return deg


def make_csr(n, nnz, d_max, seed=0, device="cpu", clustered=False, ncols=None, sigma=1.0):
    """(rowptr int32 [n+1], col int32 [nnz]) of an n x ncols matrix, columns sorted per row."""
    ncols = n if ncols is None else ncols
    assert nnz < 2 ** 31 and n < 2 ** 31
    deg = degrees(n, nnz, d_max, seed, device, sigma)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(deg, 0, out=rowptr[1:])
    g = _gen(seed + 1, device)
    row = torch.repeat_interleave(torch.arange(n, device=device, dtype=torch.int64), deg)
    if clustered:
        # neighbours within ~1% of the id range of the row (community-like locality)
        spread = max(ncols // 100, 8)
        off = (torch.randn(nnz, generator=g, device=device) * spread).round().to(torch.int64)
        col = torch.remainder(row * ncols // max(n, 1) + off, ncols)
    else:
        col = torch.randint(0, ncols, (nnz,), generator=g, device=device, dtype=torch.int64)
    key = row * ncols + col
    del row
    key, _ = torch.sort(key)
    col = torch.remainder(key, ncols).to(torch.int32)
    return rowptr.to(torch.int32), col


def _csr_from_pairs(row, col, n, ncols):
    key = row * ncols + col
    del row, col
    key, _ = torch.sort(key)
    r = torch.div(key, ncols, rounding_mode="floor")
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=key.device)
    rowptr[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
    return rowptr.to(torch.int32), torch.remainder(key, ncols).to(torch.int32)


def shuffle_ids(rowptr, col, seed=0):
    """The same graph under a random relabelling of its nodes (rows and columns alike): whatever locality the generator's id order
    carried is gone, the structure (who shares neighbours with whom) stays -- what a dataset with arbitrary node ids looks like."""
    n = rowptr.numel() - 1
    dev = col.device
    perm = torch.randperm(n, generator=_gen(seed + 7, dev), device=dev)
    deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    row = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), deg)
    return _csr_from_pairs(perm[row], perm[col.to(torch.int64)], n, n)


def make_sbm(n, nnz, d_max, blocks, p_in=0.8, seed=0, device="cpu", sigma=1.0, shuffle=True):
    """Degree-corrected stochastic block model in a dataset's shape: ``blocks`` communities of equal size (consecutive ids before the
    shuffle), a row's degree from the same log-normal as make_csr, a fraction ``p_in`` of its entries inside its own community (uniform
    over its members), the rest uniform over all nodes.  ``shuffle``: node ids relabelled at random afterwards (shuffle_ids)."""
    assert nnz < 2 ** 31 and n < 2 ** 31 and 1 <= blocks <= n
    deg = degrees(n, nnz, d_max, seed, device, sigma)
    g = _gen(seed + 1, device)
    row = torch.repeat_interleave(torch.arange(n, device=device, dtype=torch.int64), deg)
    size = (n + blocks - 1) // blocks
    b0 = torch.div(row, size, rounding_mode="floor") * size                      # first id of the row's community
    bs = torch.clamp(n - b0, max=size)                                             # its size (the last one may be short)
    inside = torch.rand(nnz, generator=g, device=device) < p_in
    u = torch.rand(nnz, generator=g, device=device, dtype=torch.float64)
    col = torch.where(inside, b0 + (u * bs).to(torch.int64), (u * n).to(torch.int64)).clamp_(max=n - 1)
    rowptr, colind = _csr_from_pairs(row, col, n, n)
    return shuffle_ids(rowptr, colind, seed) if shuffle else (rowptr, colind)


def make_rmat(n, nnz, a=0.57, b=0.19, c=0.19, seed=0, device="cpu", shuffle=True):
    """R-MAT (Chakrabarti et al.) edges in a dataset's shape: ids drawn bit by bit with quadrant probabilities (a, b, c, 1 - a - b - c) in
    the smallest power-of-two id space >= n and folded into [0, n) -- skewed degrees on BOTH sides (hub rows and hub columns), weak
    community structure.  Multi-edges kept (CSR does not coalesce, backend_pim/spmm.py:44-55)."""
    assert nnz < 2 ** 31 and n < 2 ** 31
    bits = max(1, (n - 1).bit_length())
    g = _gen(seed + 1, device)
    row = torch.zeros(nnz, dtype=torch.int64, device=device)
    col = torch.zeros(nnz, dtype=torch.int64, device=device)
    for _ in range(bits):
        u = torch.rand(nnz, generator=g, device=device)
        rbit = (u >= a + b).to(torch.int64)                                        # quadrants c, d: lower half of the rows
        cbit = (((u >= a) & (u < a + b)) | (u >= a + b + c)).to(torch.int64)       # quadrants b, d: right half of the columns
        row = row * 2 + rbit
        col = col * 2 + cbit
    rowptr, colind = _csr_from_pairs(torch.remainder(row, n), torch.remainder(col, n), n, n)
    return shuffle_ids(rowptr, colind, seed) if shuffle else (rowptr, colind)


# communities of the structured stand-ins: Reddit has 41 labelled subreddit classes and far finer communities; products 47 categories
SBM_BLOCKS = {"reddit": 50, "ogbn-products": 1200, "reddit-mini": 8, "products-mini": 40, "cora": 7}


def make_shape(name, seed=0, device="cpu", clustered=False, kind=None, shuffle=True):
    """kind: None / "uniform", "clustered" (columns near the row id), "sbm" (communities, ids shuffled unless shuffle=False), "rmat"."""
    n, nnz, d_max = SHAPES[name]
    if kind == "sbm":
        return make_sbm(n, nnz, d_max, SBM_BLOCKS.get(name, max(1, n // 4000)), seed=seed, device=device, shuffle=shuffle)
    if kind == "rmat":
        return make_rmat(n, nnz, seed=seed, device=device, shuffle=shuffle)
    return make_csr(n, nnz, d_max, seed, device, clustered or kind == "clustered")


def csr_to_coo_coalesced(rowptr, col, dtype):
    """Row-sorted, duplicate-free (row, col, value) like torch's coalesce(): duplicates of the
    multigraph become values > 1 (backend_pim/spmm.py:40-42)."""
    n = rowptr.numel() - 1
    deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    row = torch.repeat_interleave(torch.arange(n, device=col.device, dtype=torch.int64), deg)
    ncols = int(col.max()) + 1 if col.numel() else 1
    key = row * ncols + col.to(torch.int64)
    uniq, counts = torch.unique_consecutive(key, return_counts=True)
    return (uniq // ncols).to(torch.int32), (uniq % ncols).to(torch.int32), counts.to(dtype)


def features(n, h, dtype, seed=0, device="cpu", kind="driver"):
    """X of the reference driver (``randint(-8, 4)``, exact in every dtype) or uniform(-1, 1)
    floats (``kind='uniform'``, exercises the floating-point tolerance)."""
    g = _gen(seed + 2, device)
    if kind == "uniform":
        assert dtype in (torch.float32, torch.float64)
        return torch.rand((n, h), generator=g, device=device, dtype=dtype) * 2 - 1
    return torch.randint(-8, 4, (n, h), generator=g, device=device, dtype=torch.int64).to(dtype)


def algorithmic_bytes(nrows, ncols, nnz, h, elem_bytes, fmt="CSR", with_values=True):
    """Compulsory traffic of one product, every array touched once:
    idx + nnz*sizeof(val) + ncols*h*sizeof(val) + nrows*h*sizeof(val)."""
    idx = 4 * (nrows + 1) + 4 * nnz if fmt == "CSR" else 8 * nnz
    vals = nnz * elem_bytes if with_values else 0
    return idx + vals + ncols * h * elem_bytes + nrows * h * elem_bytes


def gather_bytes(nrows, nnz, h, elem_bytes, fmt="CSR"):
    """Gather-model traffic: every stored entry pulls one row of X through the cache hierarchy."""
    idx = 4 * (nrows + 1) + 4 * nnz if fmt == "CSR" else 8 * nnz
    return idx + nnz * h * elem_bytes + nrows * h * elem_bytes


def flops(nnz, h):
    return 2 * nnz * h


__all__ = ["SHAPES", "degrees", "make_csr", "make_sbm", "make_rmat", "shuffle_ids", "make_shape", "csr_to_coo_coalesced", "features",
           "algorithmic_bytes", "gather_bytes", "flops"]
