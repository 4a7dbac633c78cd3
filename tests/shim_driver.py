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
if variant in ("spmm", "grande"):
    # MatrixMarket debug ops of the default and the grande variant (utils.hpp:139-173; spmm_grande/pytorch_api.cpp:338-342): against the reference reader built in
    # place (oracle/_ref/libref_utils.so) when present, else against the stated semantics
    import tempfile

    import oracle as _oracle

    rng_m = np.random.default_rng(11)
    for nr_, nc_ in ((30, 20), (31, 17)):
        nnz_ = 6 * nr_
        rr, cc = rng_m.integers(0, nr_, nnz_), rng_m.integers(0, nc_, nnz_)
        with tempfile.NamedTemporaryFile("w", suffix=".mtx", delete=False) as f:
            f.write("%%MatrixMarket matrix coordinate real general\n% a comment\n")
            f.write(f"{nr_} {nc_} {nnz_}\n")
            for a_, b_ in zip(rr, cc):
                f.write(f"{a_ + 1} {b_ + 1} 2.5\n")
        order = np.argsort(rr, kind="stable")
        want_ptr = np.concatenate([[0], np.cumsum(np.bincount(rr, minlength=nr_ + nr_ % 2))])
        got = (ops.read_matrix_nrows(f.name), ops.read_matrix_ncols(f.name), ops.read_matrix_rowptr(f.name),
               ops.read_matrix_colind(f.name), ops.read_matrix_values(f.name))
        assert got[0] == nr_ + nr_ % 2 and got[1] == nc_ + nc_ % 2
        assert got[2].dtype == got[3].dtype == got[4].dtype == torch.int32
        assert got[2].tolist() == want_ptr.tolist() and got[3].tolist() == cc[order].tolist() and got[4].tolist() == [1] * nnz_
        if _oracle.have_ref_utils():
            rn, rc, rptr, rcol, rval = _oracle.ref_read_matrix_csr(f.name)
            assert (rn, rc) == got[:2] and rptr.tolist() == got[2].tolist() and rcol.tolist() == got[3].tolist()
            assert rval.tolist() == got[4].tolist()
        os.unlink(f.name)
    # ... and the committed outputs of the reference reader (tests/golden/mtx_ref.npz)
    zg = np.load(os.path.join(ROOT, "tests", "golden", "mtx_ref.npz"))
    for k in range(int(zg["n_cases"])):
        with tempfile.NamedTemporaryFile("wb", suffix=".mtx", delete=False) as f:
            f.write(zg[f"text_{k}"].tobytes())
        assert [ops.read_matrix_nrows(f.name), ops.read_matrix_ncols(f.name)] == zg[f"shape_{k}"].tolist()
        assert ops.read_matrix_rowptr(f.name).tolist() == zg[f"rowptr_{k}"].tolist()
        assert ops.read_matrix_colind(f.name).tolist() == zg[f"colind_{k}"].tolist()
        assert ops.read_matrix_values(f.name).tolist() == zg[f"values_{k}"].tolist()
        os.unlink(f.name)
    print("OK mtx")
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


# the shim validates what it hands to the C side (round-1 advisor finding: part count / shapes were unchecked there)
def must_raise(fn, what):
    try:
        fn()
    except RuntimeError as e:
        assert what in str(e), (what, str(e)[:300])
        return
    raise SystemExit(f"no error for: {what}")


xd = x.cuda()
if variant == "spmm":
    blocks = list(torch.chunk(xd, 2, dim=1))
    must_raise(lambda: ops.spmm_coo_run_group(A.sp_info_ptr, blocks[:1]), "expected 2 dense parts, got 1")
    must_raise(lambda: ops.spmm_coo_run_group(A.sp_info_ptr, [blocks[0], blocks[1][:-7]]), "dense part 1 has shape")
    must_raise(lambda: ops.spmm_coo_run_group(A.sp_info_ptr, [blocks[0], blocks[1][:, :-1]]), "dense part 1 has shape")
    must_raise(lambda: ops.spmm_coo_run_group(A.sp_info_ptr, [blocks[0], blocks[1].long()]), "expected scalar type")
elif variant == "grande":
    wins = []
    for i, block in enumerate(torch.split(xd, [p.size(1) for p in A.csr], dim=0)):
        wins += grande.dense_split(block, A.dense_ncols[i])
    assert np.array_equal(ops.spmm_csr_run_group(A.sp_info_ptr, wins).cpu().numpy(), ref)
    must_raise(lambda: ops.spmm_csr_run_group(A.sp_info_ptr, wins[:-1]), "dense parts, got")
    must_raise(lambda: ops.spmm_csr_run_group(A.sp_info_ptr, wins[:-1] + [wins[-1][:-3]]), "has shape")
else:
    want = A.coo[0].size(1)
    vecs = [torch.zeros(want, 1, dtype=torch.int32, device="cuda") for _ in range(8)]
    must_raise(lambda: ops.spmv_coo_run_group(A.sp_info_ptr, vecs[:5]), "expected 8 dense parts, got 5")
    must_raise(lambda: ops.spmv_coo_run_group(A.sp_info_ptr, vecs[:7] + [vecs[7][:-2]]), "elements, expected")
    out = ops.spmv_coo_run_group(A.sp_info_ptr, vecs)
    assert tuple(out.shape) == (A.coo[0].size(0), 8)
ops.dpu_release()
print("OK gpu")
