"""Runs in a SUBPROCESS (one pim_ops registration per process, like the reference): loads a
libbackend_pim.so with torch.ops.load_library -- the reference's own loading call -- and drives it
through the backend_pim wrappers.  argv: variant  [gpu]"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
variant = sys.argv[1]
on_gpu = len(sys.argv) > 2 and sys.argv[2] == "gpu"
path = os.path.join(ROOT, "backend_pim", {"spmm": "spmm_default", "grande": "spmm_grande", "spmv": "spmv_sparseP"}[variant],
                    "build", "libbackend_pim.so")
torch.ops.load_library(path)  # spmm_test.py:111
ops = torch.ops.pim_ops
names = {"spmm": ["spmm_csr_to_device_group", "spmm_csr_run_group", "spmm_coo_to_device_group", "spmm_coo_run_group"],
         "grande": ["spmm_csr_to_device_group", "spmm_csr_run_group"],
         "spmv": ["spmv_coo_to_device_group", "spmv_coo_run_group"]}[variant]
for nme in names + ["dpu_init_ranks", "dpu_init_dpus", "dpu_release", "spmm_free_group"]:
    assert hasattr(ops, nme), nme
if not on_gpu:
    try:
        ops.dpu_init_ranks(1)
    except RuntimeError as e:
        assert "no HIP device" in str(e), e
        print("OK no-device")
        sys.exit(0)
    sys.exit("dpu_init_ranks succeeded without a device?")

import oracle  # noqa: E402
from conftest import random_csr  # noqa: E402
from pygim_amd.backend_pim import grande, spmm, spmv  # noqa: E402
from pygim_amd.sparse_tensor import SparseTensorShim  # noqa: E402

rng = np.random.default_rng(5)
n, h = 500, 64
rowptr, col = random_csr(rng, n, n, 15, long_rows=[(3, 4500)])
adj = SparseTensorShim(rowptr=torch.from_numpy(rowptr), col=torch.from_numpy(col), sparse_sizes=(n, n))
x = torch.randint(-8, 4, (n, h), dtype=torch.int32)
ref = oracle.spmm_csr(rowptr, col, None, x.numpy())
if variant == "spmm":
    ops.dpu_init_ranks(4)
    for fmt in ("CSR", "COO"):
        A = spmm.prepare_pim_spmm(adj, types.SimpleNamespace(data_type=torch.int32, sp_format=fmt, sp_parts=2, ds_parts=2, hidden_size=h))
        assert np.array_equal(A.mul(x).numpy(), ref) and np.array_equal(A.mul(x.cuda()).cpu().numpy(), ref)
elif variant == "grande":
    units = ops.dpu_init_ranks(2)
    assert list(units) == [8, 8]
    A = grande.prepare_pim_spmm_grande(adj, types.SimpleNamespace(data_type=torch.int32, sp_format="CSR", sp_parts=2, hidden_size=h), units)
    assert np.array_equal(A.mul(x).numpy(), ref) and np.array_equal(A.mul(x.cuda()).cpu().numpy(), ref)
else:
    ops.dpu_init_ranks(8)
    A = spmv.prepare_pim_spmv(adj, types.SimpleNamespace(data_type=torch.int32, sp_format="COO", sp_parts=1, ds_parts=8))
    assert np.array_equal(A.mul(x).numpy(), ref) and np.array_equal(A.mul(x.cuda()).cpu().numpy(), ref)
ops.dpu_release()
print("OK gpu")
