"""Row / nnz partitioning with the reference's semantics (host logic, torch).

These are the walks of backend_pim/spmm_default/support/partition.c, used here to cut
a graph across GPUs instead of across DPUs:
  partition_by_row  -- equal row counts, remainder on the first parts   (partition.c:14-46)
  partition_by_nnz  -- greedy nnz balance at row granularity            (partition.c:51-99)
  partition_equal_nnz -- equal nnz ranges, rows may straddle            (partition.c:231-262)
Checked against the reference's own partition.c (oracle/_ref) in tests/.
"""
from __future__ import annotations

import torch


def partition_by_row(nrows: int, nparts: int):
    if nparts == 1:
        return [0, nrows]
    base, rest = divmod(nrows, nparts)
    out, cur = [0], 0
    for p in range(nparts):
        cur = min(cur + base + (1 if p < rest else 0), nrows)
        out.append(cur)
    return out


def partition_by_nnz(rowptr: torch.Tensor, nparts: int):
    """Close a part as soon as its running nnz reaches floor(nnz / nparts); every part holds at
    least one row while rows remain; leftovers merge into the last part; missing parts are empty."""
    n = rowptr.numel() - 1
    if nparts == 1:
        return [0, n]
    rp = rowptr.to(torch.int64).cpu()
    target = int(rp[-1] - rp[0]) // nparts
    closed, base = [], 0
    while len(closed) < nparts and base < n:
        r = int(torch.searchsorted(rp, rp[base] + target, right=False))
        r = max(r, base + 1)
        if r > n:
            break  # the remaining rows never reach the target
        closed.append(r)
        base = r
    out = [0] + closed + [n] * (nparts - len(closed))
    out[nparts] = n
    return out


def partition_equal_nnz(nnz: int, nparts: int):
    base, rest = divmod(nnz, nparts)
    out = [0]
    for p in range(nparts):
        out.append(out[-1] + base + (1 if p < rest else 0))
    return out


def split_widths(total: int, nparts: int):
    """Feature split of the reference: ceil-sized blocks, remainder last (spmm.py:62-72)."""
    width = (total + nparts - 1) // nparts
    sizes = [width] * nparts
    if nparts * width != total:
        sizes[-1] = total - (nparts - 1) * width
    return sizes
