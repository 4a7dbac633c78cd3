"""Shared machinery of the three ``backend_pim`` wrappers.

Mirrors what backend_pim/spmm.py, grande.py and spmv.py of the reference each
re-implement: holding the raw SparseTensor, cutting it into ``sp_parts`` column
blocks (spmm.py:127-136), materialising per-part int32 CSR / coalesced COO arrays
(spmm.py:31-55) and calling ``torch.ops.pim_ops``.
"""
from __future__ import annotations

import torch

TORCH_TYPES = {"INT64": torch.int64, "INT32": torch.int32, "INT16": torch.int16, "INT8": torch.int8,
               "FLT32": torch.float32, "DBL64": torch.float64}


class CSRPart:
    """int32 CSR arrays of one column block (what the reference reads back from a
    torch.sparse_csr_tensor: crow_indices / col_indices / values)."""

    def __init__(self, crow, col, values, shape):
        self._crow, self._col, self._values, self._shape = crow, col, values, tuple(shape)

    def crow_indices(self):
        return self._crow

    def col_indices(self):
        return self._col

    def values(self):
        return self._values

    def size(self, dim):
        return self._shape[dim]


def part_values(item, dtype):
    """Edge values of a part in the compute dtype; all ones when the adjacency has none
    (spmm.py:36-39, 48-51)."""
    value = item.storage.value()
    if value is None:
        return torch.ones(item.nnz(), dtype=dtype, device=item.device())
    return value.type(dtype)


def split_widths(total, nparts):
    """ceil-sized blocks with the remainder in the last one (spmm.py:62-72, 129-133)."""
    width = (total + nparts - 1) // nparts
    sizes = [width] * nparts
    if nparts * width != total:
        sizes[-1] = total - (nparts - 1) * width
    return sizes


class SparseGroupBase:
    def __init__(self, coo, dtype=torch.int32, format=""):
        self.raw = coo
        self.dtype = dtype
        self.format = format
        self._handle = None  # (handle, generation) of the device group this object owns
        self.sp_info_ptr = None
        self.result = None
        self.parts = [self.raw]
        self.csr, self.coo = [], []
        self.dense_parts = 0
        self.hidden_size = 0
        self.nparts = 1

    # -- the device group's lifetime ----------------------------------------------
    # The reference leaves spmm_free_group (spmm_default/pytorch_api.cpp:198-201) to the caller and its wrappers never call
    # it: a DPU group is a few arrays.  Here a group owns its CSR, its plans and -- for the LDS-staged product -- up to a GB of
    # executable code, so the wrapper frees what it created: when the handle is replaced (to_pim_group again) and when the
    # object dies.  A handle is an address: after dpu_release (every group freed) or a spmm_free_group by the caller it may
    # belong to a later group, so the wrapper remembers the group's creation serial (pygim_group_serial, never reused) and
    # frees only while the handle still names THAT group.
    @property
    def sp_info_ptr(self):
        return self._handle[0] if self._handle is not None else None

    @sp_info_ptr.setter
    def sp_info_ptr(self, handle):
        self.free_group()
        if handle is not None:
            from .. import _lib

            try:
                self._handle = (int(handle), _lib.group_serial(handle))
            except _lib.PygimError:   # not a group of this library (another pim_ops backend registered under the same names): not ours to free
                self._handle = (int(handle), None)

    def free_group(self):
        """spmm_free_group on the group this object created, if it is still alive."""
        from .. import _lib

        held, self._handle = getattr(self, "_handle", None), None
        if held is None or held[1] is None or not _lib.is_initialized():
            return
        try:
            alive = _lib.group_serial(held[0]) == held[1]
        except _lib.PygimError:  # freed by the caller or by dpu_release
            alive = False
        if alive:
            torch.ops.pim_ops.spmm_free_group(held[0])

    def __del__(self):
        try:
            self.free_group()
        except Exception:  # interpreter shutdown: torch.ops or the library may be gone already
            pass

    # -- partitioning -----------------------------------------------------------
    def col_split(self, nparts=4):
        assert nparts > 0
        if nparts != len(self.parts):
            assert len(self.parts) == 1
            step = (self.raw.size(1) + nparts - 1) // nparts
            cuts = [(i * step, (i + 1) * step) for i in range(nparts - 1)] + [((nparts - 1) * step, None)]
            self.parts = [self.raw[:, a:b] if b is not None else self.raw[:, a:] for a, b in cuts]
            self.csr, self.coo = [], []
        return self.parts

    def row_split(self, nparts=4):
        # not implemented in the reference either (spmm.py:124-125); the multi-GPU row split
        # lives in pygim_amd.dist
        assert False

    # -- per-part arrays ----------------------------------------------------------
    def build_csr(self):
        self.csr = []
        for item in self.parts:
            rowptr, col, _ = item.csr()
            self.csr.append(CSRPart(rowptr.int().contiguous(), col.int().contiguous(),
                                    part_values(item, self.dtype).contiguous(), item.sizes()))

    def _coalesced(self, item, shape):
        row, col, _ = item.coo()
        return torch.sparse_coo_tensor(torch.stack([row, col], dim=0), part_values(item, self.dtype),
                                       shape).coalesce()

    def build_coo(self):
        self.coo = [self._coalesced(item, item.sizes()) for item in self.parts]

    def _coo_arrays(self):
        self.row_indices = [c.indices()[0].int().contiguous() for c in self.coo]
        self.col_indices = [c.indices()[1].int().contiguous() for c in self.coo]
        self.values = [c.values() for c in self.coo]
        return [c.size(0) for c in self.coo], [c.size(1) for c in self.coo]

    # -- quantise -> aggregate -> dequantise in one device call (pygim_amd/quantize.py) ---------------
    def mul_quantized(self, x: torch.Tensor, post=None):
        """``dequantize(self.mul(quantize(x)))`` with models/quantize.py's arithmetic, on device.
        x: float32 CUDA tensor [ncols, hidden_size].  Returns (out float32 [nrows, hidden_size], scale).
        ``post = (col_mul, col_add, relu)``: per-column epilogue ``out = col_mul * out + col_add`` (then ReLU) applied in
        the sweep's last store (a GCN layer's bias + eval-mode BatchNorm + ReLU folded into one affine map)."""
        from .. import _lib

        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.size(1) == self.hidden_size
        x = x.contiguous()
        out = torch.empty((self.raw.size(0), self.hidden_size), dtype=torch.float32, device=x.device)
        scale = torch.empty((), dtype=torch.float32, device=x.device)
        mul_p = add_p = 0
        relu = False
        if post is not None:
            col_mul, col_add, relu = post
            col_mul = col_mul.to(x.device, torch.float32).contiguous()
            col_add = col_add.to(x.device, torch.float32).contiguous()
            assert col_mul.numel() == self.hidden_size and col_add.numel() == self.hidden_size
            mul_p, add_p = col_mul.data_ptr(), col_add.data_ptr()
        _lib.quant_spmm_run(self.sp_info_ptr, x.data_ptr(), x.size(1), out.data_ptr(), scale.data_ptr(),
                            torch.cuda.current_stream(x.device).cuda_stream, mul_p, add_p, relu)
        return out, scale
