"""Runs under torch.distributed.run (gloo, every rank on the one visible GPU): the multi-GPU layer of pygim_amd/dist.py with
its real engines (HipEngine / HipShardEngine -> C ABI -> HIP kernels) against the CPU oracle.  A logic check of the N > 1
paths on a one-GPU box; prints "OK rank r"."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from conftest import driver_features, random_csr  # noqa: E402
from pygim_amd import dist as pd  # noqa: E402

backend = os.environ.get("PYGIM_BENCH_BACKEND", "gloo")
if backend == "nccl":  # RCCL: one rank per GPU (on a one-GPU box: one rank, collectives forced on by PYGIM_FORCE_COLLECTIVES=1)
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(lr)
    dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
else:
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
world, rank = dist.get_world_size(), dist.get_rank()
rng = np.random.default_rng(321)
n, h = 1500, 96
rowptr, col = random_csr(rng, n, n, 20, long_rows=[(7, 3000)])
x = driver_features(rng, n, h, np.int32)
ref = oracle.spmm_csr(rowptr, col, None, x)
rp_t, col_t = torch.from_numpy(rowptr), torch.from_numpy(col)
xd = torch.from_numpy(x).cuda()

rs = pd.RowSplitSpMM(rp_t, col_t, None, n, torch.int32, h)
assert np.array_equal(rs.mul(xd).cpu().numpy(), ref), "row split"
fs = pd.FeatureSplitSpMM(rp_t, col_t, None, n, torch.int32, h)
assert np.array_equal(fs.gather(fs.mul_local(fs.local_features(xd))).cpu().numpy(), ref), "feature split"
for rparts in range(1, world + 1):
    if world % rparts == 0:
        gr = pd.GridSpMM(rp_t, col_t, None, n, torch.int32, h, rparts, world // rparts)
        assert np.array_equal(gr.gather(gr.mul_local(gr.local_features(xd))).cpu().numpy(), ref), ("grid", rparts)
import scipy.sparse as sp  # noqa: E402

a = sp.csr_matrix((np.ones(len(col), dtype=np.int32), col.copy(), rowptr.copy()), shape=(n, n))
step = (n + world - 1) // world
c0, c1 = rank * step, min(n, (rank + 1) * step)
loc = a[:, c0:c1].tocsr()
loc.sort_indices()
cs = pd.ColSplitSpMM(torch.from_numpy(loc.indptr.astype(np.int32)), torch.from_numpy(loc.indices.astype(np.int32)),
                     torch.from_numpy(loc.data.astype(np.int32)), n, c1 - c0, torch.int32, h)
assert np.array_equal(cs.mul(xd[c0:c1].contiguous()).cpu().numpy(), ref), "column split"
if n % world == 0 and backend == "nccl":  # (gloo has no reduce_scatter_tensor)
    part = cs.mul(xd[c0:c1].contiguous(), reduce_scatter=True)  # reduce-scatter form of the same merge
    r0 = rank * (n // world)
    assert np.array_equal(part.cpu().numpy(), ref[r0:r0 + n // world]), "column split, reduce-scatter"
xf = rng.standard_normal((n, h)).astype(np.float32)
for tdt, npdt in ((torch.int8, np.int8), (torch.int32, np.int32)):
    s_ref, xq_ref = oracle.symmetric_quantize(xf, npdt)
    want = oracle.symmetric_dequantize(oracle.spmm_csr(rowptr, col, None, xq_ref), 1.0, s_ref)
    sh = pd.RowShardAdj(rp_t, col_t, n, tdt, h)
    got, _ = sh.mul_quantized(sh.local_rows(torch.from_numpy(xf)).contiguous().cuda())
    assert np.array_equal(got.cpu().numpy(), want[sh.r0:sh.r1]), ("row shard", tdt)
torch.cuda.synchronize()
dist.barrier()
print(f"OK rank {rank}", flush=True)
dist.destroy_process_group()
