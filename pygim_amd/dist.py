"""sp_parts / ds_parts across the GPUs of one node (one process per GPU, torch.distributed).

The reference spreads its partitions over UPMEM ranks and merges on the host
(spmm_mul_csr.c:481-551).  Here a partition is a GPU and the merge is an RCCL collective
over xGMI (backend "nccl"); on CPU the same code runs over "gloo" (tests).

  RowSplitSpMM      sp_parts = world as a ROW split of A (BASELINE config 4; the reference's
                    row_split is `assert False`, spmm.py:124-125).  X replicated, rank r owns an
                    nnz-balanced block of rows (partition_by_nnz walk) and produces C[rows_r, :].
                    Collective: all-gather of the row blocks (only when the full C is needed).
                    Bit-exact for every dtype: one GPU sums each output row in stored order.
  ColSplitSpMM      sp_parts = world as the reference's COLUMN split (spmm.py:127-136): rank r
                    holds A[:, cols_r] (local column ids) and X[cols_r, :], produces a full-size
                    partial C; collective: sum all-reduce (or reduce-scatter).  Integers exact,
                    floats reorder (within the 1e-5 bound).
  FeatureSplitSpMM  ds_parts = world (BASELINE config 5): A replicated, rank r owns the feature
                    block X[:, f_r] (widths as spmm.py:62-72) and produces C[:, f_r]; no collective
                    inside the product; all-gather along features re-assembles C.

  GridSpMM          row_parts x feat_parts grid (rank (i, j) = row block i x feature block j).
  RowShardAdj       config 4 end to end: activations row-sharded through the dense layers, the aggregation
                    all-gathers the QUANTISED row blocks (see the class).

The local product goes through an *engine*: HipEngine (the C ABI, device tensors) by default.
Tests inject an engine built on the CPU oracle to exercise the partition/collective logic with
world_size 2 over gloo.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist

from . import partition


class HipEngine:
    """Local block product on this rank's GPU through the C ABI."""

    def __init__(self):
        from . import _lib, pim_ops

        self._lib = _lib
        self._code = pim_ops.DTYPE_CODE
        if not _lib.is_initialized():
            _lib.init_ranks(dist.get_world_size() if dist.is_initialized() else 1)

    def create(self, rowptr, col, values, nrows, ncols, dtype, h):
        dev = torch.device("cuda", torch.cuda.current_device())
        self.keep = [rowptr.to(dev, torch.int32).contiguous(), col.to(dev, torch.int32).contiguous()]
        vptr = None
        if values is not None:
            self.keep.append(values.to(dev, dtype).contiguous())
            vptr = [self.keep[2].data_ptr()]
        self.dtype, self.nrows, self.h = dtype, nrows, h
        self.handle = self._lib.group_create(self._lib.CSR, self._code[dtype], [self.keep[0].data_ptr()],
                                             [self.keep[1].data_ptr()], vptr, [nrows], [ncols],
                                             [self.keep[1].numel()], [1], [h], h)
        return self

    def run(self, x, out=None):
        assert x.is_cuda and x.dtype == self.dtype and x.is_contiguous()
        if out is None:
            out = torch.empty((self.nrows, self.h), dtype=self.dtype, device=x.device)
        st = torch.cuda.current_stream(x.device).cuda_stream
        self._lib.spmm_run_group(self.handle, [x.data_ptr()], out.data_ptr(), st)
        return out

    def free(self):
        self._lib.group_free(self.handle)


def _world(group):
    if not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def _collect(world):
    """Do the collectives run?  Always on several ranks; on ONE rank only when a process group exists and
    PYGIM_FORCE_COLLECTIVES=1 (the first-run check of the RCCL path on a single GPU: every all-gather / all-reduce of the
    N > 1 code really executes at world size 1, tests/test_rccl_gpu.py)."""
    import os

    return world > 1 or (dist.is_initialized() and os.environ.get("PYGIM_FORCE_COLLECTIVES", "0") == "1")


class RowSplitSpMM:
    def __init__(self, rowptr, col, values, ncols, dtype, h, group=None, balance="nnz", engine_factory=HipEngine):
        self.group = group
        self.world, self.rank = _world(group)
        n = rowptr.numel() - 1
        self.nrows_total, self.h, self.dtype = n, h, dtype
        self.split = (partition.partition_by_nnz(rowptr, self.world) if balance == "nnz"
                      else partition.partition_by_row(n, self.world))
        r0, r1 = self.split[self.rank], self.split[self.rank + 1]
        lo, hi = int(rowptr[r0]), int(rowptr[r1])
        self.r0, self.r1 = r0, r1
        self.max_rows = max(self.split[i + 1] - self.split[i] for i in range(self.world))
        self.engine = engine_factory().create((rowptr[r0:r1 + 1] - lo), col[lo:hi],
                                              None if values is None else values[lo:hi], r1 - r0, ncols, dtype, h)

    def mul_local(self, x_full: torch.Tensor, out=None) -> torch.Tensor:
        """C[rows_r, :] for this rank (X replicated on every rank)."""
        return self.engine.run(x_full, out)

    def mul(self, x_full: torch.Tensor) -> torch.Tensor:
        """Full C on every rank: local product written straight into this rank's slot of the
        gather buffer, then one all-gather of equal-sized (padded) row blocks."""
        buf = torch.empty((self.world, self.max_rows, self.h), dtype=self.dtype, device=x_full.device)
        mine = buf[self.rank]
        self.engine.run(x_full, mine[: self.r1 - self.r0])
        if _collect(self.world):
            dist.all_gather_into_tensor(buf.view(-1), mine.reshape(-1), group=self.group)
        if all(self.split[i + 1] - self.split[i] == self.max_rows for i in range(self.world)):
            return buf.view(self.world * self.max_rows, self.h)
        return torch.cat([buf[i, : self.split[i + 1] - self.split[i]] for i in range(self.world)], dim=0)


class ColSplitSpMM:
    def __init__(self, local_rowptr, local_col, local_values, nrows, local_ncols, dtype, h, group=None,
                 engine_factory=HipEngine):
        self.group = group
        self.world, self.rank = _world(group)
        self.engine = engine_factory().create(local_rowptr, local_col, local_values, nrows, local_ncols, dtype, h)

    def mul(self, x_local_rows: torch.Tensor, reduce_scatter: bool = False) -> torch.Tensor:
        """partial = A[:, cols_r] . X[cols_r, :]; summed over ranks (spmm_mul_csr.c:491-502)."""
        part = self.engine.run(x_local_rows)
        if not _collect(self.world):
            return part
        if reduce_scatter and part.size(0) % self.world == 0:
            out = torch.empty((part.size(0) // self.world, part.size(1)), dtype=part.dtype, device=part.device)
            dist.reduce_scatter_tensor(out, part, op=dist.ReduceOp.SUM, group=self.group)
            return out
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
        return part


class FeatureSplitSpMM:
    def __init__(self, rowptr, col, values, ncols, dtype, h_total, group=None, engine_factory=HipEngine):
        self.group = group
        self.world, self.rank = _world(group)
        self.widths: List[int] = partition.split_widths(h_total, self.world)
        self.h_total = h_total
        self.f0 = sum(self.widths[: self.rank])
        self.nrows = rowptr.numel() - 1
        self.dtype = dtype
        self.engine = engine_factory().create(rowptr, col, values, self.nrows, ncols, dtype, self.widths[self.rank])

    def local_features(self, x_full: torch.Tensor) -> torch.Tensor:
        return x_full[:, self.f0:self.f0 + self.widths[self.rank]].contiguous()

    def mul_local(self, x_block: torch.Tensor) -> torch.Tensor:
        """C[:, f_r]: no collective (the reference concatenates ds_parts, spmm_mul_csr.c:483,504)."""
        return self.engine.run(x_block)

    def gather(self, c_block: torch.Tensor) -> torch.Tensor:
        """Full C on every rank: all-gather of the feature blocks (staged [world, N, wmax], then
        laid side by side)."""
        if not _collect(self.world):
            return c_block
        wmax = max(self.widths)
        stage = torch.zeros((self.world, self.nrows, wmax), dtype=c_block.dtype, device=c_block.device)
        mine = stage[self.rank]
        mine[:, : c_block.size(1)] = c_block
        dist.all_gather_into_tensor(stage.view(-1), mine.reshape(-1).clone(), group=self.group)
        return torch.cat([stage[i, :, : self.widths[i]] for i in range(self.world)], dim=1)


class GridSpMM:
    """2-D ``row_parts x feat_parts`` grid over the world (the reference's rank (i, j) = sparse part i x dense
    part j, spmm_mul_csr.c:344-345, with sp_parts as a ROW split): rank ``i * feat_parts + j`` owns the
    nnz-balanced row block i of A and the feature block j of X (all rows of it), and produces
    C[rows_i, f_j] with no collective inside the product.  What the grid buys over a pure feature split: the
    gathered row of X stays >= 128 bytes (a narrower row still moves a whole 128-byte line per entry,
    scripts/exp_rowbytes.py), e.g. papers100M h = 128 on 8 GPUs as 2 x 4 instead of 1 x 8."""

    def __init__(self, rowptr, col, values, ncols, dtype, h_total, row_parts, feat_parts, group=None,
                 engine_factory=HipEngine):
        self.group = group
        self.world, self.rank = _world(group)
        assert row_parts * feat_parts == self.world, "grid must cover the world"
        self.row_parts, self.feat_parts = row_parts, feat_parts
        self.i, self.j = divmod(self.rank, feat_parts)
        self.nrows, self.h_total, self.dtype = rowptr.numel() - 1, h_total, dtype
        self.split = partition.partition_by_nnz(rowptr, row_parts)
        self.widths: List[int] = partition.split_widths(h_total, feat_parts)
        self.r0, self.r1 = self.split[self.i], self.split[self.i + 1]
        self.f0 = sum(self.widths[: self.j])
        self.w = self.widths[self.j]
        lo, hi = int(rowptr[self.r0]), int(rowptr[self.r1])
        self.engine = engine_factory().create(rowptr[self.r0:self.r1 + 1] - lo, col[lo:hi],
                                              None if values is None else values[lo:hi], self.r1 - self.r0, ncols, dtype,
                                              self.w)

    def local_features(self, x_full: torch.Tensor) -> torch.Tensor:
        return x_full[:, self.f0:self.f0 + self.w].contiguous()

    def mul_local(self, x_block: torch.Tensor) -> torch.Tensor:
        """C[rows_i, f_j] of this rank"""
        return self.engine.run(x_block)

    def gather(self, c_block: torch.Tensor) -> torch.Tensor:
        """full C on every rank: one all-gather of padded [max_rows, max_width] tiles, then laid out"""
        if not _collect(self.world):
            return c_block
        mr = max(max(self.split[a + 1] - self.split[a] for a in range(self.row_parts)), 1)
        mw = max(self.widths)
        stage = torch.zeros((self.world, mr, mw), dtype=c_block.dtype, device=c_block.device)
        stage[self.rank, : c_block.size(0), : c_block.size(1)] = c_block
        dist.all_gather_into_tensor(stage.view(-1), stage[self.rank].reshape(-1).clone(), group=self.group)
        out = torch.empty((self.nrows, self.h_total), dtype=c_block.dtype, device=c_block.device)
        for a in range(self.row_parts):
            f = 0
            for b in range(self.feat_parts):
                out[self.split[a]:self.split[a + 1], f:f + self.widths[b]] = \
                    stage[a * self.feat_parts + b, : self.split[a + 1] - self.split[a], : self.widths[b]]
                f += self.widths[b]
        return out


class RowSplitAdj:
    """What a conv layer's ``adj_t`` is on N GPUs (BASELINE config 4): an nnz-balanced ROW block of A per
    rank, features replicated.  ``mul_quantized`` = quantise (over the full, replicated X) -> local rows
    of A . X_q -> dequantise, then one all-gather of the row blocks; ``mul`` = the plain product."""

    def __init__(self, rowptr, col, ncols, dtype, h, group=None):
        from . import _lib, pim_ops

        self._lib = _lib
        self.group = group
        self.world, self.rank = _world(group)
        self.dtype, self.hidden_size = dtype, h
        self.nrows = rowptr.numel() - 1
        if not _lib.is_initialized():
            _lib.init_ranks(self.world)
        self.split = partition.partition_by_nnz(rowptr, self.world)
        r0, r1 = self.split[self.rank], self.split[self.rank + 1]
        lo, hi = int(rowptr[r0]), int(rowptr[r1])
        dev = torch.device("cuda", torch.cuda.current_device())
        self._keep = [(rowptr[r0:r1 + 1] - lo).to(dev, torch.int32).contiguous(), col[lo:hi].to(dev, torch.int32).contiguous()]
        self.handle = _lib.group_create(_lib.CSR, pim_ops.DTYPE_CODE[dtype], [self._keep[0].data_ptr()],
                                        [self._keep[1].data_ptr()], None, [r1 - r0], [ncols], [hi - lo], [1], [h], h)
        self.my_rows = r1 - r0
        self.max_rows = max(self.split[i + 1] - self.split[i] for i in range(self.world))

    def _gather(self, buf):
        if _collect(self.world):
            dist.all_gather_into_tensor(buf.view(-1), buf[self.rank].reshape(-1), group=self.group)
        if self.world == 1 or all(self.split[i + 1] - self.split[i] == self.max_rows for i in range(self.world)):
            return buf.view(self.world * self.max_rows, -1)[: self.nrows]
        return torch.cat([buf[i, : self.split[i + 1] - self.split[i]] for i in range(self.world)], dim=0)

    def mul_quantized(self, x, post=None):
        assert x.is_cuda and x.dtype == torch.float32 and x.size(1) == self.hidden_size
        x = x.contiguous()
        buf = torch.empty((self.world, self.max_rows, self.hidden_size), dtype=torch.float32, device=x.device)
        scale = torch.empty((), dtype=torch.float32, device=x.device)
        mul_p = add_p = 0
        relu = False
        if post is not None:  # per-column epilogue in the sweep's last store (see SparseGroupBase.mul_quantized)
            col_mul, col_add, relu = post
            col_mul, col_add = col_mul.to(x.device, torch.float32).contiguous(), col_add.to(x.device, torch.float32).contiguous()
            mul_p, add_p = col_mul.data_ptr(), col_add.data_ptr()
        self._lib.quant_spmm_run(self.handle, x.data_ptr(), x.size(1), buf[self.rank].data_ptr(), scale.data_ptr(),
                                 torch.cuda.current_stream(x.device).cuda_stream, mul_p, add_p, relu)
        return self._gather(buf), scale

    def mul(self, x):
        assert x.is_cuda and x.dtype == self.dtype
        x = x.contiguous()
        buf = torch.empty((self.world, self.max_rows, self.hidden_size), dtype=self.dtype, device=x.device)
        self._lib.spmm_run_group(self.handle, [x.data_ptr()], buf[self.rank].data_ptr(),
                                 torch.cuda.current_stream(x.device).cuda_stream)
        return self._gather(buf)


class HipShardEngine:
    """Device steps of RowShardAdj through the C ABI: |max| bits, quantise, product, dequantise."""

    def __init__(self):
        from . import _lib, pim_ops

        self._lib = _lib
        self._code = pim_ops.DTYPE_CODE
        if not _lib.is_initialized():
            _lib.init_ranks(dist.get_world_size() if dist.is_initialized() else 1)

    def create(self, rowptr, col, nrows, ncols, dtype, h):
        dev = torch.device("cuda", torch.cuda.current_device())
        self.dev, self.dtype, self.nrows, self.h = dev, dtype, nrows, h
        self.code = self._code[dtype]
        self.keep = [rowptr.to(dev, torch.int32).contiguous(), col.to(dev, torch.int32).contiguous()]
        self.handle = self._lib.group_create(self._lib.CSR, self.code, [self.keep[0].data_ptr()], [self.keep[1].data_ptr()],
                                             None, [nrows], [ncols], [self.keep[1].numel()], [1], [h], h)
        return self

    def _st(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.dev)

    def absmax_bits(self, x):
        bits = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._lib.quant_absmax(x.data_ptr(), x.stride(0) if x.dim() == 2 and x.size(0) > 1 else x.size(-1), x.size(0), x.size(1),
                               bits.data_ptr(), self._st())
        return bits

    def quantize(self, x, bits, out):
        self._lib.quantize(self.code, x.data_ptr(), x.stride(0) if x.size(0) > 1 else x.size(1), x.size(0), x.size(1),
                           bits.data_ptr(), out.data_ptr(), 0, self._st())

    def product(self, xq_full, out):
        self._lib.spmm_run_group(self.handle, [xq_full.data_ptr()], out.data_ptr(), self._st())

    def dequantize(self, q, bits, out):
        self._lib.dequantize(self.code, q.data_ptr(), q.numel(), bits.data_ptr(), out.data_ptr(), self._st())

    def product_dequantized(self, xq_full, bits, out):
        """out = float(A . xq) * scale in one sweep (the last store of every row dequantises)"""
        self._lib.spmm_run_dequant(self.handle, xq_full.data_ptr(), xq_full.size(1), out.data_ptr(), bits.data_ptr(), self._st())

    def free(self):
        self._lib.group_free(self.handle)


class RowShardAdj:
    """``adj_t`` of a conv layer when the node features are ROW-SHARDED over the GPUs (BASELINE config 4).

    Rank r keeps the nnz-balanced row block rows_r of every activation through the dense layers (Linear, BN,
    ReLU are row-wise: no communication, 1/world of the work) and of A.  The aggregation
    ``dequantize(A . quantize(x))`` (pyg_gcn_conv.py:130-137, quantize.py:20-42) exchanges the QUANTISED block:

        bits   = max|x[rows_r]|  --MAX all-reduce, 4 bytes-->  the global scale (same value as on one GPU)
        x_q[rows_r] = round(x[rows_r] / scale) in the adjacency type (INT8: a quarter of the fp32 bytes)
        all-gather of the x_q row blocks --> X_q on every rank
        out[rows_r] = float(A[rows_r, :] . X_q) * scale

    The gathered blocks stay in their padded slots ([world, max_rows, h]); A's column ids are re-based once
    to that layout (owner * max_rows + offset inside the owner's block: monotonic, rows stay sorted), so no
    compaction pass is needed.  Every output row is summed by one GPU in stored order: integers bit-exact and
    floats bit-identical to the one-GPU result, whatever the world size."""

    row_sharded = True  # quantize.message_and_aggregate: x is a row block, the scale is global

    def __init__(self, rowptr, col, ncols, dtype, h, group=None, engine_factory=HipShardEngine):
        self.group = group
        self.world, self.rank = _world(group)
        self.dtype, self.hidden_size = dtype, h
        self.nrows = rowptr.numel() - 1
        assert ncols == self.nrows, "row-sharded activations need a square adjacency"
        self.split = partition.partition_by_nnz(rowptr, self.world)
        self.r0, self.r1 = self.split[self.rank], self.split[self.rank + 1]
        self.my_rows = self.r1 - self.r0
        self.max_rows = max(max(self.split[i + 1] - self.split[i] for i in range(self.world)), 1)
        lo, hi = int(rowptr[self.r0]), int(rowptr[self.r1])
        c = col[lo:hi].to(torch.int64)
        starts = torch.tensor(self.split[:-1], dtype=torch.int64, device=c.device)
        owner = torch.bucketize(c, torch.tensor(self.split[1:-1], dtype=torch.int64, device=c.device), right=True)
        c_padded = c + owner * self.max_rows - starts[owner]
        self.engine = engine_factory().create(rowptr[self.r0:self.r1 + 1] - lo, c_padded, self.my_rows,
                                              self.world * self.max_rows, dtype, h)
        self._xq = None

    def local_rows(self, t):
        """this rank's block of a node-indexed tensor"""
        return t[self.r0:self.r1]

    def _exchange(self, buf):
        if _collect(self.world):
            dist.all_gather_into_tensor(buf.view(-1), buf[self.rank].reshape(-1), group=self.group)
        return buf.view(self.world * self.max_rows, self.hidden_size)

    def _buffer(self, dtype):
        if self._xq is None or self._xq.dtype != dtype:
            self._xq = self.engine.empty((self.world, self.max_rows, self.hidden_size), dtype)
        return self._xq

    def mul_quantized(self, x_local):
        assert x_local.dtype == torch.float32 and tuple(x_local.shape) == (self.my_rows, self.hidden_size), \
            (tuple(x_local.shape), self.my_rows, self.hidden_size)
        e = self.engine
        x_local = x_local.contiguous()
        bits = e.absmax_bits(x_local)
        if _collect(self.world):
            dist.all_reduce(bits, op=dist.ReduceOp.MAX, group=self.group)
        buf = self._buffer(self.dtype)
        e.quantize(x_local, bits, buf[self.rank][: self.my_rows])
        xq = self._exchange(buf)
        out = e.empty((self.my_rows, self.hidden_size), torch.float32)
        if hasattr(e, "product_dequantized"):
            e.product_dequantized(xq, bits, out)
        else:  # engines without the fused sweep (the CPU test engine): product, then dequantise
            out_q = e.empty((self.my_rows, self.hidden_size), self.dtype)
            e.product(xq, out_q)
            e.dequantize(out_q, bits, out)
        return out, bits

    def mul(self, x_local):
        """plain product on row-sharded features of the adjacency type: [rows_r, h] -> [rows_r, h]"""
        assert x_local.dtype == self.dtype and tuple(x_local.shape) == (self.my_rows, self.hidden_size)
        buf = self._buffer(self.dtype)
        buf[self.rank][: self.my_rows].copy_(x_local)
        out = self.engine.empty((self.my_rows, self.hidden_size), self.dtype)
        self.engine.product(self._exchange(buf), out)
        return out
