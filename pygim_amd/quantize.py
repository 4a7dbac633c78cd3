"""Symmetric quantisation around the aggregation, as the reference's conv layers do it.

Same names and arithmetic as the reference's models/quantize.py:20-42 (``symmetric_quantize``,
``symmetric_dequantize``) plus ``message_and_aggregate`` -- the body of
``*Conv.message_and_aggregate`` (pyg_gcn_conv.py:130-137, pyg_gin_conv.py:93-101,
pyg_sage_conv.py:147-155).  With a device-resident backend object the three steps run as ONE
device call (``SparseTensorCOO.mul_quantized`` -> ``pygim_quant_spmm_run``): no host round trip
between quantise, aggregate and dequantise, and the features never leave HBM.
"""
import torch

from .sparse_tensor import SparseTensor, matmul


def symmetric_quantize(v, dtype=torch.int32):
    abs_max = torch.max(v.abs())
    if dtype == torch.int8:
        scale = abs_max * 2 / pow(2, 5)
    elif dtype == torch.int16:
        scale = abs_max * 2 / pow(2, 10)
    elif dtype == torch.int32:
        scale = abs_max * 2 / pow(2, 20)
    else:  # any other "dtype" (including SparseTensor.dtype, a bound method on the cpu path)
        scale = abs_max * 2 / pow(2, 20)
        dtype = torch.float
    new_v = torch.round(v / scale)
    new_v = new_v.clone() if new_v.dtype == dtype else new_v.to(dtype)
    return scale, new_v


def symmetric_dequantize(out, scale_edge, scale_x):
    return out * (scale_edge * scale_x)


def message_and_aggregate(adj_t, x, fused=True):
    """quantise -> aggregate -> dequantise.  ``adj_t``: a SparseTensor (cpu path, torch_sparse.matmul
    semantics) or a backend_pim ``SparseTensorCOO``."""
    if isinstance(adj_t, SparseTensor):
        scale, x_q = symmetric_quantize(x, dtype=adj_t.dtype)
        return symmetric_dequantize(matmul(adj_t, x_q), 1., scale)
    if getattr(adj_t, "row_sharded", False):
        # row-sharded activations (pygim_amd.dist.RowShardAdj): x is this rank's row block, so the scale must be the
        # GLOBAL max|x| (MAX all-reduce inside mul_quantized) -- quantising the local block on its own would give every
        # rank a different scale.  Always the device quantiser, whatever `fused` says.
        if adj_t.dtype not in (torch.int8, torch.int16, torch.int32, torch.float32):
            raise RuntimeError(f"row-sharded aggregation needs an INT8/INT16/INT32/FLT32 adjacency, not {adj_t.dtype}")
        out, _ = adj_t.mul_quantized(x)
        return out
    if fused and x.is_cuda and x.dtype == torch.float32 and hasattr(adj_t, "mul_quantized") and \
            adj_t.dtype in (torch.int8, torch.int16, torch.int32, torch.float32):
        out, _ = adj_t.mul_quantized(x)
        return out
    scale, x_q = symmetric_quantize(x, dtype=adj_t.dtype)
    return symmetric_dequantize(adj_t.mul(x_q), 1., scale)
