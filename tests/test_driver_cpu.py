"""BASELINE configs[0]: the reference driver's own CPU-runnable case -- ``spmm_test.py --version cpu --dataset Cora
--hidden_size 32 --data_type FLT32`` (reference spmm_test.py:24-37, 110: the torch_sparse.matmul leg only, no backend
loaded, no GPU touched).  Also: that leg's product equals the committed Cora golden vector's definition of A . X."""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spmm_test_version_cpu_cora_flt32():
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")  # no device even on a GPU box
    r = subprocess.run([sys.executable, os.path.join(ROOT, "spmm_test.py"), "--version", "cpu", "--dataset", "Cora",
                        "--hidden_size", "32", "--data_type", "FLT32", "--repeat", "2"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    out = r.stdout
    assert "Cora Dataset Info: Node(2708), Edge(10556)" in out
    times = [float(ln.split(":")[1]) for ln in out.splitlines() if ln.startswith("[DATA]torch_time(ms)")]
    assert len(times) == 2 and all(t >= 0 for t in times)
    assert out.count("-------------------- Model=spmm_test Repeat=") == 2
    # the cpu leg never reaches the backend: no pim timing, no equality line, no library load
    assert "[DATA]pim_time_spmm(ms)" not in out and "[DATA]outputs_equal" not in out
    assert "libpygim_hip" not in out + r.stderr


def test_cpu_leg_is_the_product_the_golden_vectors_define():
    """pygim_amd.sparse_tensor.matmul (the stand-in for torch_sparse.matmul on the version=cpu path) on the Cora fixture"""
    from pygim_amd.sparse_tensor import SparseTensor, matmul

    z = np.load(os.path.join(ROOT, "tests", "golden", "spmm_cora_csr_FLT32.npz"))
    n = len(z["rowptr"]) - 1
    adj = SparseTensor(rowptr=torch.from_numpy(z["rowptr"].astype(np.int64)), col=torch.from_numpy(z["col"].astype(np.int64)),
                       sparse_sizes=(n, n))
    y = matmul(adj, torch.from_numpy(z["x"]))
    assert np.array_equal(y.numpy(), z["y"])  # integer-valued driver features: exact in any summation order
