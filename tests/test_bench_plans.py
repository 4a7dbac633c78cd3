"""CPU, world size 8 (and 2, 3) over gloo: every RCCL arrangement `bench.py --gpus N` can choose (pygim_amd/bench_plans.py: row
pieces, pipelined rows, feature pieces, pipelined features) assembles the oracle's C on every rank.  The product engine is the
CPU oracle behind the library's Python surface (tests/fake_abi.py), streams and events are inert stand-ins, the collectives
are real (gloo).  The push arrangements need HIP IPC and are exercised on the GPU box only (tests/test_bench_gpu.py).
Reference being replaced: the host-side N-way merge spmm_default/spmm_mul_csr.c:481-551, rank -> (sp, ds) block :344-345."""
import contextlib
import os
import socket
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _Stream:
    cuda_stream = 0

    def wait_event(self, e):
        pass

    def wait_stream(self, s):
        pass


class _Event:
    def record(self, s=None):
        pass


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle
    from conftest import random_csr
    from fake_abi import FakeLib
    from pygim_amd import bench_plans

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)       # the same graph on every rank
        n, h = 413, 64
        rowptr_np, col_np = random_csr(rng, n, n, 9, long_rows=[(7, 900)])
        x_np = rng.integers(-8, 4, size=(n, h)).astype(np.float32)
        ref = oracle.spmm_csr(rowptr_np, col_np, None, x_np)
        rowptr, col, x = torch.from_numpy(rowptr_np.astype(np.int32)), torch.from_numpy(col_np.astype(np.int32)), torch.from_numpy(x_np)
        lib = FakeLib()
        lib.init_ranks(world)
        main = _Stream()
        env = SimpleNamespace(world=world, rank=rank, multi=True, dev=torch.device("cpu"), n=n, nnz=len(col_np), h=h, x=x, rowptr=rowptr,
                              col=col, rowptr_cpu=rowptr, split=bench_plans.nnz_balanced_row_split(rowptr, world), main_stream=main,
                              stream=0, lib=lib, dist=dist, Stream=_Stream, Event=_Event,
                              stream_ctx=lambda s: contextlib.nullcontext(), synchronize=lambda: None)
        assert env.split[0] == 0 and env.split[-1] == n and len(env.split) == world + 1
        plans = bench_plans.build(env)
        cases = [("Pieces", 1), ("Pieces", 2), ("Pieces", 4), ("PipelinedRows", 1), ("FeaturePieces", 1), ("FeaturePieces", 2),
                 ("PipelinedFeatures", 1)]
        for name, k in cases:
            if "Feature" in name and h % world:
                continue
            pl = plans[name](k)
            for _ in range(3):      # odd count: both gather buffers of the pipelined forms get used
                pl.step()
            if hasattr(pl, "drain"):
                pl.drain()
            dist.barrier()
            full = pl.full_c().numpy()
            assert full.shape == (n, h) and np.array_equal(full, ref), (name, k, rank)
            # the products-only step (what bench.py times beside the exchange) must not touch the collectives
            pl.step(exchange=False)
            assert isinstance(pl.describe(), str) and pl.my_rows > 0
            pl.free()
            dist.barrier()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_bench_arrangements_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), res
