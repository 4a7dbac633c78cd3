""""Grande" wrapper -- same public surface as the reference's backend_pim/grande.py.

Layout (grande.py:53-107): sparse column block i is held whole by every unit of
rank i; unit j of that rank gets feature window j (``dpus_per_rank[i]`` windows whose
widths differ by at most one, padded to an 8-byte multiple).  On MI355X a "unit" is
an XCD-sized feature window of the same launch (``dpu_init_ranks`` reports 8 per rank).
"""
import torch

from ._common import TORCH_TYPES, SparseGroupBase  # noqa: F401

# elements per 8 bytes (grande.py:11)
TYPES_MUL = {torch.int64: 1, torch.int32: 2, torch.int16: 4, torch.int8: 8, torch.float32: 2, torch.float64: 1}


def dense_split(B, ncols, dim=1):
    """Per-unit feature windows of one row block of X, each ``pad`` columns wide where
    pad = first width rounded up to 8 bytes; windows start at the running sum of the true
    widths, so neighbours overlap by the padding (grande.py:12-23)."""
    mul = TYPES_MUL[B.dtype]
    pad = (int(ncols[0]) + mul - 1) // mul * mul
    tail = int(ncols[-1]) % pad
    if tail != 0:
        B = torch.nn.functional.pad(B, (0, pad - tail))
    if len(ncols) == 1:
        return [B.contiguous()]
    out, start = [], 0
    for n in ncols:
        out.append(B[:, start:start + pad].contiguous())
        start += int(n)
    return out


class SparseTensorCOO(SparseGroupBase):
    def __init__(self, coo, dtype=torch.int32, dpus_per_rank=[], format=""):
        super().__init__(coo.int(), dtype=dtype, format="")
        self.dpus_per_rank = dpus_per_rank

    def to_pim_group_csr(self, hidden_size, B_parts=4):
        self.free_group()  # (a group made earlier goes before its replacement is built, not after)
        self.format = "CSR"
        self.hidden_size = hidden_size
        if len(self.csr) != len(self.parts):
            self.build_csr()
        self.dense_ncols = []
        for i in range(len(self.csr)):
            units = int(self.dpus_per_rank[i])
            base, extra = divmod(hidden_size, units)
            self.dense_ncols.append(torch.tensor([base + (1 if u < extra else 0) for u in range(units)],
                                                 dtype=torch.int32))
        self.sp_info_ptr = torch.ops.pim_ops.spmm_csr_to_device_group(
            [p.crow_indices() for p in self.csr], [p.col_indices() for p in self.csr],
            [p.values() for p in self.csr], [p.size(0) for p in self.csr], [p.size(1) for p in self.csr],
            self.dense_ncols, hidden_size)

    def mul(self, B: torch.Tensor):
        assert self.hidden_size == B.size(1)
        assert len(self.dpus_per_rank) == len(self.csr)
        row_blocks = torch.split(B, [p.size(1) for p in self.csr], dim=0)
        windows = []
        for i, block in enumerate(row_blocks):
            windows += dense_split(block, self.dense_ncols[i])
        if self.format == "CSR":
            return torch.ops.pim_ops.spmm_csr_run_group(self.sp_info_ptr, windows)
        if self.format == "COO":
            return torch.ops.pim_ops.spmm_coo_run_group(self.sp_info_ptr, windows)
        return None


def prepare_pim_spmm_grande(adj_t, args, dpus_per_rank):
    A = SparseTensorCOO(adj_t, dtype=args.data_type, dpus_per_rank=dpus_per_rank, format=args.sp_format)
    A.col_split(args.sp_parts)
    A.to_pim_group_csr(args.hidden_size)
    return A


def pim_spmm_grande(x, adj_t: SparseTensorCOO):
    return adj_t.mul(x)
