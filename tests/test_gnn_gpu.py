"""End-to-end inference stacks on the GPU: the aggregation inside GCN / SAGE / GIN
goes through the HIP backend (fused quantise -> SpMM -> dequantise) and must equal the same network with
the aggregation written in dense torch (int64 matmul of the quantised features)."""
import types

import numpy as np
import pytest
import torch

from conftest import random_csr
from pygim_amd import gnn, pim_ops, quantize
from pygim_amd.backend_pim import spmm as spmm_mod
from pygim_amd.sparse_tensor import SparseTensorShim

pytestmark = pytest.mark.gpu


class DenseAdj:
    """reference aggregator: same quantiser, exact integer product with a dense adjacency"""

    def __init__(self, dense_int64, dtype):
        self.a, self.dtype = dense_int64, dtype

    def mul(self, x_q):
        return (self.a.double() @ x_q.double()).to(x_q.dtype)  # exact: |sums| < 2^53


@pytest.mark.parametrize("name", ["gcn", "sage", "gin"])
def test_inference_stack_matches_dense_reference(rng, name):
    n, fin, h, ncls = 700, 48, 64, 7
    rowptr, col = random_csr(rng, n, n, 12, long_rows=[(3, 2000)])
    adj = SparseTensorShim(rowptr=torch.from_numpy(rowptr), col=torch.from_numpy(col), sparse_sizes=(n, n))
    dense = adj.to_dense(torch.float64).to(torch.int64).cuda()
    pim_ops.load("spmm")
    torch.ops.pim_ops.dpu_init_ranks(2)
    try:
        A = spmm_mod.prepare_pim_spmm(adj.to("cuda"), types.SimpleNamespace(data_type=torch.int32, sp_format="CSR", sp_parts=2,
                                                                               ds_parts=1, hidden_size=h))
        torch.manual_seed(3)
        model = {"gcn": gnn.GCN, "sage": gnn.SAGE, "gin": gnn.GIN}[name](fin, h, ncls, num_layers=3).cuda().eval()
        x = torch.randn(n, fin, device="cuda")
        with torch.no_grad():
            out = model(x, A, None)
            ref = model(x, DenseAdj(dense, torch.int32), None)
        assert out.shape == (n, ncls)
        assert torch.equal(out, ref)
    finally:
        torch.ops.pim_ops.dpu_release()


def test_row_split_adj_single_rank(rng):
    """the multi-GPU adjacency object with world_size 1 (the collective is the identity)"""
    from pygim_amd.dist import RowSplitAdj

    n, h = 500, 64
    rowptr, col = random_csr(rng, n, n, 10)
    adj = RowSplitAdj(torch.from_numpy(rowptr), torch.from_numpy(col), n, torch.int32, h)
    x = torch.randn(n, h, device="cuda")
    out, scale = adj.mul_quantized(x)
    s_ref, xq = quantize.symmetric_quantize(x, torch.int32)
    dense = SparseTensorShim(rowptr=torch.from_numpy(rowptr), col=torch.from_numpy(col), sparse_sizes=(n, n)).to_dense(torch.float64)
    prod = (dense.cuda() @ xq.double()).to(torch.int32)
    ref = prod * (1.0 * s_ref)
    assert torch.equal(out, ref) and torch.equal(scale, s_ref)
    assert torch.equal(adj.mul(xq), prod)
    adj._lib.release()
