"""The reference's Python surface on the GPU: torch.ops.pim_ops (real HIP library) driven through
the backend_pim wrappers exactly as spmm_test.py drives the reference, CPU tensors (staged) and
device tensors (resident), checked against the oracle."""
import types

import numpy as np
import pytest
import torch

import oracle
from conftest import ALL_DTYPES, random_csr
from pygim_amd import pim_ops
from pygim_amd.backend_pim import grande as grande_mod
from pygim_amd.backend_pim import spmm as spmm_mod
from pygim_amd.backend_pim import spmv as spmv_mod
from pygim_amd.sparse_tensor import SparseTensorShim

pytestmark = pytest.mark.gpu
TORCH_OF = {"INT8": torch.int8, "INT16": torch.int16, "INT32": torch.int32, "INT64": torch.int64,
            "FLT32": torch.float32, "DBL64": torch.float64}


def make_adj(rng, n=600, deg=20, device="cpu"):
    rowptr, col = random_csr(rng, n, n, deg, long_rows=[(7, 5000)])
    adj = SparseTensorShim(rowptr=torch.from_numpy(rowptr), col=torch.from_numpy(col), sparse_sizes=(n, n))
    return (adj.to(device) if device != "cpu" else adj), rowptr, col


def ns(**kw):
    return types.SimpleNamespace(**kw)


@pytest.mark.parametrize("dt", ALL_DTYPES)
@pytest.mark.parametrize("fmt", ["CSR", "COO"])
def test_spmm_wrapper_cpu_and_device_tensors(rng, dt, fmt):
    pim_ops.load("spmm")
    tdt = TORCH_OF[dt]
    adj, rowptr, col = make_adj(rng)
    h = 96
    torch.ops.pim_ops.dpu_init_ranks(6)
    try:
        A = spmm_mod.prepare_pim_spmm(adj, ns(data_type=tdt, sp_format=fmt, sp_parts=3, ds_parts=2, hidden_size=h))
        x = torch.randint(-8, 4, (600, h)).to(tdt)
        ref = oracle.spmm_csr(rowptr, col, None, x.numpy())
        out = A.mul(x)
        assert out.device.type == "cpu" and np.array_equal(out.numpy(), ref)
        out_d = A.mul(x.cuda())
        assert out_d.is_cuda and np.array_equal(out_d.cpu().numpy(), ref)
        # adjacency resident on the device: arrays are used in place
        B = spmm_mod.prepare_pim_spmm(adj.to("cuda"), ns(data_type=tdt, sp_format=fmt, sp_parts=2, ds_parts=1, hidden_size=h))
        assert np.array_equal(B.mul(x.cuda()).cpu().numpy(), ref)
        torch.ops.pim_ops.spmm_free_group(B.sp_info_ptr)
    finally:
        torch.ops.pim_ops.dpu_release()


def test_grande_wrapper_gpu(rng):
    pim_ops.load("grande")
    adj, rowptr, col = make_adj(rng)
    units = torch.ops.pim_ops.dpu_init_ranks(2)
    try:
        for tdt, h in ((torch.int32, 256), (torch.int8, 100), (torch.float32, 21), (torch.float64, 64)):
            A = grande_mod.prepare_pim_spmm_grande(adj, ns(data_type=tdt, sp_format="CSR", sp_parts=2, hidden_size=h), units)
            x = torch.randint(-8, 4, (600, h)).to(tdt)
            ref = oracle.spmm_csr(rowptr, col, None, x.numpy())
            assert np.array_equal(grande_mod.pim_spmm_grande(x, A).numpy(), ref), (tdt, h)
            assert np.array_equal(A.mul(x.cuda()).cpu().numpy(), ref)
    finally:
        torch.ops.pim_ops.dpu_release()


def test_spmv_wrapper_gpu(rng):
    pim_ops.load("spmv")
    adj, rowptr, col = make_adj(rng, n=597)
    torch.ops.pim_ops.dpu_init_ranks(8)
    try:
        for tdt in (torch.int8, torch.int16, torch.int32, torch.int64):
            for groups in (1, 2, 4, 8):  # (2..4: the device path plans a second group for the whole-matrix product)
                A = spmv_mod.prepare_pim_spmv(adj, ns(data_type=tdt, sp_format="COO", sp_parts=1, ds_parts=groups))
                x = torch.randint(-8, 4, (597, 16)).to(tdt)
                ref = oracle.spmm_csr(rowptr, col, None, x.numpy())
                assert np.array_equal(spmv_mod.pim_spmv(x, A).numpy(), ref), (tdt, groups)
                # the reference's own loop: h / groups backend calls of `groups` SpMVs each (spmv.py:95-102)
                looped = torch.cat([A.mul_single(panel) for panel in spmv_mod.dense_split(x, 16 // groups)], dim=1)
                assert np.array_equal(looped.numpy(), ref), (tdt, groups, "mul_single loop")
                assert np.array_equal(A.mul(x.cuda()).cpu().numpy(), ref)
                assert np.array_equal(A.mul(x.cuda()).cpu().numpy(), ref)  # (second call: the cached wide group)
    finally:
        torch.ops.pim_ops.dpu_release()


def test_conv_layer_call_pattern(rng):
    """message_and_aggregate of the reference's conv layers (pyg_gcn_conv.py:130-137):
    quantise -> adj_t.mul(x_q) -> dequantise, with the quantiser's arithmetic (quantize.py:20-42)"""
    pim_ops.load("spmm")
    adj, rowptr, col = make_adj(rng)
    torch.ops.pim_ops.dpu_init_ranks(1)
    try:
        A = spmm_mod.prepare_pim_spmm(adj, ns(data_type=torch.int32, sp_format="CSR", sp_parts=1, ds_parts=1, hidden_size=64))
        x = torch.randn(600, 64)
        scale = x.abs().max() * 2 / pow(2, 20)
        x_q = torch.round(x / scale).to(torch.int32)
        out = A.mul(x_q.cuda()).cpu() * (1.0 * scale)
        ref = torch.from_numpy(oracle.spmm_csr(rowptr, col, None, x_q.numpy())) * (1.0 * scale)
        assert torch.equal(out, ref)
        dense = adj.to_dense(torch.float64) @ x.double()
        assert torch.allclose(out.double(), dense, atol=float(scale) * 6000)
    finally:
        torch.ops.pim_ops.dpu_release()


@pytest.mark.parametrize("dt", ["INT8", "INT16", "INT32", "FLT32"])
def test_fused_quantise_aggregate_dequantise(rng, dt):
    """one device call == the three reference steps (quantize.py:20-42 around adj_t.mul), bit for bit"""
    from pygim_amd import quantize as qz

    pim_ops.load("spmm")
    tdt = TORCH_OF[dt]
    adj, rowptr, col = make_adj(rng)
    torch.ops.pim_ops.dpu_init_ranks(2)
    try:
        A = spmm_mod.prepare_pim_spmm(adj.to("cuda"), ns(data_type=tdt, sp_format="CSR", sp_parts=2, ds_parts=1, hidden_size=64))
        x = torch.randn(600, 64)
        out, scale = A.mul_quantized(x.cuda())
        # oracle: numpy restatement of the quantiser + the SpMM oracle
        s_ref, xq_ref = oracle.symmetric_quantize(x.numpy(), np.dtype(np.float32) if dt == "FLT32" else NP_OF[dt])
        outq_ref = oracle.spmm_csr(rowptr, col, None, xq_ref)
        ref = oracle.symmetric_dequantize(outq_ref, 1.0, s_ref)
        assert np.float32(scale.item()) == s_ref
        if dt == "FLT32":
            # "quantised" floats are integers up to 2^19: sums of a 5000-entry row exceed 2^24, so the
            # float result depends on the summation order (long items are summed by 8 lane groups)
            mag = oracle.spmm_csr(rowptr, col, None, np.abs(xq_ref)) * s_ref
            assert np.all(np.abs(out.cpu().numpy() - ref) <= 1e-5 * mag + 1e-30)
        else:
            assert np.array_equal(out.cpu().numpy(), ref)
        # and the unfused torch path of this package gives the same tensor
        unfused = qz.message_and_aggregate(A, x.cuda(), fused=False)
        assert torch.equal(unfused.cpu(), out.cpu())  # same kernels, same order -> same bits
        assert torch.equal(qz.message_and_aggregate(A, x.cuda()).cpu(), out.cpu())
    finally:
        torch.ops.pim_ops.dpu_release()


NP_OF = {"INT8": np.int8, "INT16": np.int16, "INT32": np.int32}
