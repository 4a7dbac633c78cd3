"""SparseP-style SpMV wrapper -- same public surface as the reference's backend_pim/spmv.py.

An SpMM with h features runs as h SpMVs, ``groups`` (= ds_parts) vectors per backend
call (spmv.py:89-102).  Integer dtypes only (``torch.iinfo``, spmv.py:45); the matrix is
padded to a multiple of 64/bits rows and columns (spmv.py:45-51) and results are cut
back to the true row count (spmv.py:93).
"""
import torch

from ._common import TORCH_TYPES, SparseGroupBase, split_widths  # noqa: F401


def dense_split(B, nparts, dim=1):
    if nparts == 1:
        return [B.contiguous()]
    return [chunk.contiguous() for chunk in torch.chunk(B, nparts, dim)]


class SparseTensorCOO(SparseGroupBase):
    def __init__(self, coo, dtype=torch.int32, groups=32):
        super().__init__(coo.int(), dtype=dtype, format="")
        self.groups = groups
        self._wide = None  # (width, handle, generation): a second group planned for the whole [N, h] product, see _mul_device

    def _drop_wide(self):
        """free the cached second group if it is still alive (dpu_release frees every group and bumps the generation:
        a handle of an earlier generation must neither be used nor freed -- its address may belong to someone else)"""
        from .. import _lib

        wide, self._wide = self._wide, None
        if wide is not None and _lib.is_initialized() and wide[2] == _lib.generation():
            torch.ops.pim_ops.spmm_free_group(wide[1])

    def __del__(self):
        try:
            self._drop_wide()
        except Exception:  # interpreter shutdown: the library may be gone already
            pass
        super().__del__()

    def build_coo(self):
        quantum = 64 // torch.iinfo(self.dtype).bits
        self.coo = []
        for item in self.parts:
            extra = (-item.size(0)) % quantum
            self.coo.append(self._coalesced(item, (item.size(0) + extra, item.size(1) + extra)))

    def to_pim_group_coo(self, hidden_size, rank_pre_spmv=1):
        B_parts = hidden_size
        self._drop_wide()
        self.free_group()
        self.format = "COO"
        self.hidden_size = hidden_size
        self.dense_parts = B_parts
        self.max_B_parts_ncols = (hidden_size + B_parts - 1) / B_parts
        if len(self.coo) != len(self.parts):
            self.build_coo()
        nrows, ncols = self._coo_arrays()
        self.sp_info_ptr = torch.ops.pim_ops.spmv_coo_to_device_group(
            self.row_indices, self.col_indices, self.values, nrows, ncols, split_widths(hidden_size, B_parts),
            hidden_size, rank_pre_spmv)

    def _pad_rows(self, B):
        # the backend multiplies the PADDED matrix: give the vectors its column count
        # (the reference reads past the end of the unpadded vector here)
        want = self.coo[0].size(1)
        if B.size(0) == want:
            return B
        return torch.nn.functional.pad(B, (0, 0, 0, want - B.size(0)))

    def mul_single(self, B: torch.Tensor):
        assert self.hidden_size == B.size(1)
        vectors = dense_split(self._pad_rows(B), self.dense_parts)
        res = torch.ops.pim_ops.spmv_coo_run_group(self.sp_info_ptr, vectors)
        return res[:self.raw.size(0), ...]

    def mul(self, B: torch.Tensor):
        if B.dtype == self.dtype and len(self.coo) == 1 and B.dim() == 2 and B.size(1) % self.groups == 0:
            if B.is_cuda:
                return self._mul_device(B)
            if torch.cuda.is_available():
                # CPU tensors: one upload, one product, one download instead of h / groups staged backend calls (same sums)
                return self._mul_device(B.cuda()).cpu()
        panels = dense_split(B, B.size(1) // self.groups)
        return torch.cat([self.mul_single(panel) for panel in panels], dim=1)

    def _mul_device(self, B: torch.Tensor):
        """device-resident features: the h SpMVs of the loop above are ONE product over the whole [N, h] matrix (same
        sums, same order per output element) -- no per-group calls, no vector packing, 128-byte gathers"""
        from .. import _lib

        Bp = self._pad_rows(B).contiguous()
        out = torch.empty((self.coo[0].size(0), Bp.size(1)), dtype=self.dtype, device=B.device)
        handle = self.sp_info_ptr
        if 2 <= self.dense_parts <= 4 and Bp.size(1) > 4:
            # groups of 2..4 vectors are planned for the LDS-staged SpMV kernel (narrow column panels); the whole-matrix
            # product wants the sweep's panels: a second group over the same arrays, made once (10.1 -> 6.9 ms, Reddit-shaped)
            if self._wide is None or self._wide[0] != Bp.size(1) or self._wide[2] != _lib.generation():
                self._drop_wide()
                nrows, ncols = [c.size(0) for c in self.coo], [c.size(1) for c in self.coo]
                self._wide = (Bp.size(1), torch.ops.pim_ops.spmv_coo_to_device_group(
                    self.row_indices, self.col_indices, self.values, nrows, ncols, [1] * Bp.size(1), Bp.size(1), 1),
                    _lib.generation())
            handle = self._wide[1]
        _lib.block_run(handle, 0, Bp.data_ptr(), Bp.size(1), out.data_ptr(), Bp.size(1), Bp.size(1), False,
                       torch.cuda.current_stream(B.device).cuda_stream)
        return out[:self.raw.size(0), ...]

    def col_split(self, nparts=4):
        assert False


def prepare_pim_spmv(adj_t, args):
    assert args.sp_format == "COO"
    A = SparseTensorCOO(adj_t, dtype=args.data_type, groups=args.ds_parts)
    A.to_pim_group_coo(args.ds_parts, args.sp_parts)
    return A


def pim_spmv(x, adj_t: SparseTensorCOO):
    return adj_t.mul(x)
