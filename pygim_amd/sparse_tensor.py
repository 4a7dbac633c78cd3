"""Minimal stand-in for ``torch_sparse.SparseTensor`` (used when torch_sparse is absent).

The reference's wrappers (backend_pim/spmm.py, grande.py, spmv.py) and drivers
(spmm_test.py:11,56; models/pyg_*_conv.py) take the adjacency as a
``torch_sparse.SparseTensor`` produced by ``T.ToSparseTensor`` and only touch the
handful of methods implemented here: ``csr() coo() nnz() size() sizes()
sparse_sizes() device() int() __getitem__[:, a:b] storage.value()``.
When the real package is importable it is used instead (see ``SparseTensor`` at the
bottom); this class exists because torch_sparse cannot be installed on the GPU box.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


class _Storage:
    def __init__(self, owner):
        self._o = owner

    def value(self):
        return self._o._value

    def row(self):
        return self._o._row

    def col(self):
        return self._o._col

    def rowptr(self):
        return self._o._rowptr


class SparseTensorShim:
    """Row-major sorted sparse matrix: (rowptr, row, col, optional value)."""

    def __init__(self, row=None, rowptr=None, col=None, value=None, sparse_sizes=None, is_sorted=False):
        assert col is not None and (row is not None or rowptr is not None)
        col = col.to(torch.int64)
        if row is None:
            rowptr = rowptr.to(torch.int64)
            counts = rowptr[1:] - rowptr[:-1]
            row = torch.repeat_interleave(torch.arange(rowptr.numel() - 1, device=col.device), counts)
            is_sorted = True
        row = row.to(torch.int64)
        if sparse_sizes is None:
            m = int(row.max()) + 1 if row.numel() else 0
            n = int(col.max()) + 1 if col.numel() else 0
            sparse_sizes = (m, n)
        self._sizes = (int(sparse_sizes[0]), int(sparse_sizes[1]))
        if not is_sorted and row.numel():
            key = row * max(self._sizes[1], 1) + col
            perm = torch.argsort(key, stable=True)
            row, col = row[perm], col[perm]
            if value is not None:
                value = value[perm]
        self._row, self._col, self._value = row, col, value
        if rowptr is None:
            counts = torch.bincount(row, minlength=self._sizes[0]) if row.numel() else torch.zeros(
                self._sizes[0], dtype=torch.int64, device=col.device)
            rowptr = torch.zeros(self._sizes[0] + 1, dtype=torch.int64, device=col.device)
            torch.cumsum(counts, 0, out=rowptr[1:])
        self._rowptr = rowptr
        self.storage = _Storage(self)

    # ---- constructors ------------------------------------------------------
    @classmethod
    def from_edge_index(cls, edge_index, edge_attr=None, sparse_sizes=None, is_sorted=False):
        return cls(row=edge_index[0], col=edge_index[1], value=edge_attr, sparse_sizes=sparse_sizes,
                   is_sorted=is_sorted)

    @classmethod
    def from_scipy(cls, mat, has_value=True):
        m = mat.tocsr()
        m.sort_indices()
        val = torch.from_numpy(m.data) if has_value else None
        return cls(rowptr=torch.from_numpy(m.indptr.astype("int64")), col=torch.from_numpy(m.indices.astype("int64")),
                   value=val, sparse_sizes=m.shape, is_sorted=True)

    def to_scipy(self, layout="csr", dtype=None):
        import numpy as np
        import scipy.sparse as sp

        val = self._value.cpu().numpy() if self._value is not None else np.ones(self.nnz(), dtype=dtype or np.float32)
        m = sp.csr_matrix((val, self._col.cpu().numpy(), self._rowptr.cpu().numpy()), shape=self._sizes)
        return m if layout == "csr" else m.asformat(layout)

    # ---- accessors the reference uses -----------------------------------------
    def csr(self) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
        return self._rowptr, self._col, self._value

    def coo(self) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
        return self._row, self._col, self._value

    def nnz(self) -> int:
        return int(self._col.numel())

    def size(self, dim: int) -> int:
        return self._sizes[dim]

    def sizes(self):
        return list(self._sizes)

    def sparse_sizes(self):
        return self._sizes

    def device(self):
        return self._col.device

    def dtype(self):  # a METHOD on torch_sparse.SparseTensor (quantize takes its float branch)
        return self._value.dtype if self._value is not None else torch.float

    def has_value(self) -> bool:
        return self._value is not None

    def set_value(self, value, layout=None):
        return SparseTensorShim(row=self._row, rowptr=self._rowptr, col=self._col, value=value,
                                sparse_sizes=self._sizes, is_sorted=True)

    def int(self):
        v = self._value
        return self if v is None else self.set_value(v.to(torch.int32))

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        return SparseTensorShim(row=mv(self._row), rowptr=mv(self._rowptr), col=mv(self._col), value=mv(self._value),
                                sparse_sizes=self._sizes, is_sorted=True)

    def t(self):
        return SparseTensorShim(row=self._col, col=self._row, value=self._value,
                                sparse_sizes=(self._sizes[1], self._sizes[0]))

    def to_dense(self, dtype=None):
        val = self._value if self._value is not None else torch.ones(self.nnz(), dtype=dtype or torch.float32)
        out = torch.zeros(self._sizes, dtype=val.dtype)
        out.index_put_((self._row, self._col), val, accumulate=True)
        return out

    # ---- column slicing: raw[:, a:b] (backend_pim/spmm.py:132-133) ---------------
    def __getitem__(self, index):
        if not (isinstance(index, tuple) and len(index) == 2):
            raise NotImplementedError("only [rows, cols] slicing is supported")
        rs, cs = index
        out = self
        if isinstance(cs, slice) and cs != slice(None):
            a, b, step = cs.indices(self._sizes[1])
            assert step == 1
            b = max(a, b)
            keep = (out._col >= a) & (out._col < b)
            val = None if out._value is None else out._value[keep]
            out = SparseTensorShim(row=out._row[keep], col=out._col[keep] - a, value=val,
                                   sparse_sizes=(out._sizes[0], b - a), is_sorted=True)
        if isinstance(rs, slice) and rs != slice(None):
            a, b, step = rs.indices(out._sizes[0])
            assert step == 1
            b = max(a, b)
            lo, hi = int(out._rowptr[a]), int(out._rowptr[b])
            val = None if out._value is None else out._value[lo:hi]
            out = SparseTensorShim(rowptr=out._rowptr[a:b + 1] - lo, col=out._col[lo:hi], value=val,
                                   sparse_sizes=(b - a, out._sizes[1]), is_sorted=True)
        return out

    def __repr__(self):
        return f"SparseTensorShim(sizes={self._sizes}, nnz={self.nnz()}, value={'yes' if self.has_value() else 'none'})"


def _shim_matmul(src, other, reduce: str = "sum"):
    """``torch_sparse.matmul`` stand-in for the version=cpu path: torch's own COO kernel.

    Sum-reduce only (the reference's call sites use the default, spmm_test.py:25).
    """
    assert reduce in ("sum", "add")
    if other.is_floating_point() and not other.is_cuda:
        # floats: torch's CSR x dense kernel (MKL, all cores) -- about 20x faster than the COO path on the
        # Reddit-shaped graph; integer element types are not implemented there
        try:
            rowptr, col, value = src.csr()
            if value is None:
                value = torch.ones(col.numel(), dtype=other.dtype)
            import warnings

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")  # "Sparse CSR tensor support is in beta state"
                a = torch.sparse_csr_tensor(rowptr, col, value.to(other.dtype), size=tuple(src.sizes()))
            return a @ other
        except RuntimeError:
            pass
    if not other.is_floating_point() and not other.is_cuda and other.numel() > 0 and src.nnz() > 0:
        # integers: torch has no CSR kernel for them and its COO kernel takes 16-23 s on the Reddit-shaped graph.  When every
        # sum provably stays below 2**53 the float64 CSR kernel gives the same integers exactly (1-2 s); the cast back
        # wraps like the element type's own arithmetic (what torch_sparse.matmul's native loop does).
        try:
            rowptr, col, value = src.csr()
            deg_max = int((rowptr[1:] - rowptr[:-1]).max())
            v_max = 1 if value is None else int(value.abs().max())
            x_max = int(other.abs().max())
            if 0 <= deg_max * v_max * x_max < (1 << 53) and v_max >= 0 and x_max >= 0:
                import warnings

                vals = torch.ones(col.numel(), dtype=torch.float64) if value is None else value.to(torch.float64)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    a = torch.sparse_csr_tensor(rowptr, col, vals, size=tuple(src.sizes()))
                return (a @ other.to(torch.float64)).to(torch.int64).to(other.dtype)
        except RuntimeError:
            pass
    row, col, value = src.coo()
    if value is None:
        value = torch.ones(src.nnz(), dtype=other.dtype, device=other.device)
    a = torch.sparse_coo_tensor(torch.stack([row, col]), value.to(other.dtype), src.sizes())
    return torch.sparse.mm(a, other)


try:  # pragma: no cover - not installable in the build image
    from torch_sparse import SparseTensor, matmul  # type: ignore

    HAVE_TORCH_SPARSE = True
except Exception:  # ModuleNotFoundError, or a broken binary wheel
    SparseTensor = SparseTensorShim
    matmul = _shim_matmul
    HAVE_TORCH_SPARSE = False
