"""Default SpMM wrapper -- same public surface as the reference's backend_pim/spmm.py:
``dense_split``, ``SparseTensorCOO`` (col_split / to_pim_group[_csr|_coo] / mul),
``TORCH_TYPES``, ``prepare_pim_spmm`` and ``pim_spmm``.

A = [A_0 | A_1 | ...] by columns (``sp_parts``), X = [X_0 | X_1 | ...] by features
(``ds_parts``); the result is sum_i A_i . X[rows_i, :] with the feature blocks side
by side (spmm.py:57-122).  The arithmetic runs in the HIP library behind
``torch.ops.pim_ops`` (pygim_amd/pim_ops.py).
"""
import torch

from ._common import TORCH_TYPES, SparseGroupBase, split_widths  # noqa: F401 (TORCH_TYPES re-exported)


def dense_split(B, nparts, dim=1):
    """Feature blocks of X as contiguous tensors (spmm.py:9-13)."""
    if nparts == 1:
        return [B.contiguous()]
    return [chunk.contiguous() for chunk in torch.chunk(B, nparts, dim)]


class SparseTensorCOO(SparseGroupBase):
    def _register(self, hidden_size, B_parts, fmt):
        self.free_group()  # (a group made earlier goes before its replacement is built, not after)
        self.format = fmt
        self.hidden_size = hidden_size
        self.dense_parts = B_parts
        self.max_B_parts_ncols = (hidden_size + B_parts - 1) / B_parts
        return split_widths(hidden_size, B_parts)

    def to_pim_group_csr(self, hidden_size, B_parts=4):
        h_size = self._register(hidden_size, B_parts, "CSR")
        if len(self.csr) != len(self.parts):
            self.build_csr()
        self.sp_info_ptr = torch.ops.pim_ops.spmm_csr_to_device_group(
            [p.crow_indices() for p in self.csr], [p.col_indices() for p in self.csr],
            [p.values() for p in self.csr], [p.size(0) for p in self.csr], [p.size(1) for p in self.csr],
            h_size, hidden_size)

    def to_pim_group_coo(self, hidden_size, B_parts=4):
        h_size = self._register(hidden_size, B_parts, "COO")
        if len(self.coo) != len(self.parts):
            self.build_coo()
        nrows, ncols = self._coo_arrays()
        self.sp_info_ptr = torch.ops.pim_ops.spmm_coo_to_device_group(
            self.row_indices, self.col_indices, self.values, nrows, ncols, h_size, hidden_size)

    def to_pim_group(self, hidden_size, B_parts=4):
        if self.format == "COO":
            self.to_pim_group_coo(hidden_size, B_parts)
        elif self.format == "CSR":
            self.to_pim_group_csr(hidden_size, B_parts)
        else:
            assert False

    def mul(self, B: torch.Tensor):
        assert self.hidden_size == B.size(1)
        blocks = dense_split(B, self.dense_parts)
        if self.format == "CSR":
            return torch.ops.pim_ops.spmm_csr_run_group(self.sp_info_ptr, blocks)
        if self.format == "COO":
            return torch.ops.pim_ops.spmm_coo_run_group(self.sp_info_ptr, blocks)
        return None


def prepare_pim_spmm(adj_t, args):
    A = SparseTensorCOO(adj_t, dtype=args.data_type, format=args.sp_format)
    A.col_split(args.sp_parts)
    A.to_pim_group(args.hidden_size, args.ds_parts)
    return A


def pim_spmm(x, adj_t: SparseTensorCOO):
    return adj_t.mul(x)
