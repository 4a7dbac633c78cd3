"""The N-GPU arrangements `bench.py --gpus N` times (moved out of bench.py in round 3 so that the CPU suite can run them at world
size 8 over gloo with the oracle as the product engine, tests/test_bench_plans.py).

Every class is one way to end a step with the full C = A.X on every rank (what the next GCN layer needs); which of the
reference's knobs it realises is in its docstring: sp_parts as a ROW split (the reference's own row_split is `assert False`,
backend_pim/spmm.py:124-125; the walk is partition_by_nnz_csr, support/partition.c:51-99), ds_parts as a FEATURE split
(spmm.py:62-72).  rank -> block mapping and the merge that the exchange replaces: spmm_default/spmm_mul_csr.c:344-345, 481-551.

`env` carries what the classes close over: world, rank, multi, dev, n, nnz, h, rowptr, col, x, rowptr_cpu, split, main_stream,
stream (raw handle), lib (pygim_amd._lib or a test double with the same functions), dist (torch.distributed), and the stream
primitives Stream / Event / stream_ctx / synchronize (torch.cuda's on a GPU, inert stand-ins on the CPU).
"""
import sys

import torch


def nnz_balanced_row_split(rowptr_cpu, nparts):
    """Same greedy walk as the reference's partition_by_nnz_csr (support/partition.c:51-99):
    close a block once its running nnz reaches floor(nnz / nparts)."""
    n = rowptr_cpu.numel() - 1
    if nparts == 1:
        return [0, n]
    rp = rowptr_cpu.to(torch.int64)
    target = int(rp[-1]) // nparts
    split = [0]
    base = 0
    for _ in range(nparts - 1):
        # first row index r with rp[r] - rp[base] >= target
        r = int(torch.searchsorted(rp, rp[base] + target, right=False))
        r = min(max(r, base), n)
        split.append(r)
        base = r
    split.append(n)
    return split


def build(env):
    """the candidate classes, closed over `env`"""
    world, rank, multi, dev = env.world, env.rank, env.multi, env.dev
    n, nnz, h, x, rowptr, col, rowptr_cpu, split = env.n, env.nnz, env.h, env.x, env.rowptr, env.col, env.rowptr_cpu, env.split
    main_stream, stream, _lib, dist = env.main_stream, env.stream, env.lib, env.dist

    class Pieces:
        """This rank's row block cut into K nnz-balanced pieces.  On N > 1 GPUs every piece runs on its
        own stream and is all-gathered (RCCL) as soon as it is done, so the exchange of piece k overlaps
        the products of the other pieces; piece 0 also re-lays X slice-major, the others wait for it."""

        def __init__(self, K):
            self.K = K
            self.bounds = []
            for r in range(world):
                a0, b0 = split[r], split[r + 1]
                sub = nnz_balanced_row_split(rowptr_cpu[a0:b0 + 1] - rowptr_cpu[a0], K)
                self.bounds.append([a0 + v for v in sub])
            self.mine_b = self.bounds[rank]
            self.handles, self.outs, self.gathers, self.keep = [], [], [], []
            self.my_rows = self.my_nnz = 0
            for c in range(K):
                c0, c1 = self.mine_b[c], self.mine_b[c + 1]
                lo, hi = int(rowptr_cpu[c0]), int(rowptr_cpu[c1])
                rp_c = (rowptr[c0:c1 + 1] - lo).contiguous()
                col_c = col[lo:hi].contiguous()
                self.keep += [rp_c, col_c]
                self.handles.append(_lib.group_create(_lib.CSR, _lib.FLT32, [rp_c.data_ptr()], [col_c.data_ptr()], None,
                                                      [c1 - c0], [n], [hi - lo], [1], [h], h))
                pad_rows = max(self.bounds[r][c + 1] - self.bounds[r][c] for r in range(world))
                # gather buffer of piece c: world blocks of pad_rows rows; this rank's block is written in place
                g = torch.empty((world, max(pad_rows, 1), h), dtype=torch.float32, device=dev)
                self.gathers.append(g)
                self.outs.append(g[rank])
                self.my_rows += c1 - c0
                self.my_nnz += hi - lo
            self.side = [env.Stream() for _ in range(K - 1)] if multi else []

        def step(self, exchange=True):
            K, handles, outs, gathers, side = self.K, self.handles, self.outs, self.gathers, self.side
            if not multi:
                for c in range(K):
                    _lib.spmm_run_group(handles[c], [x.data_ptr()], outs[c].data_ptr(), stream, x_unchanged=c > 0)
                return
            pending = []
            _lib.spmm_run_group(handles[0], [x.data_ptr()], outs[0].data_ptr(), stream)
            ready = env.Event()
            ready.record(main_stream)
            if exchange:
                pending.append(dist.all_gather_into_tensor(gathers[0].view(-1), outs[0].reshape(-1), async_op=True))
            for c in range(1, K):  # the same X: piece 0's slice-major copy is shared (x_unchanged), ordered by `ready`
                s_c = side[c - 1]
                s_c.wait_event(ready)
                with env.stream_ctx(s_c):
                    _lib.spmm_run_group(handles[c], [x.data_ptr()], outs[c].data_ptr(), s_c.cuda_stream, x_unchanged=True)
                    if exchange:
                        pending.append(dist.all_gather_into_tensor(gathers[c].view(-1), outs[c].reshape(-1), async_op=True))
            for c, wk in enumerate(pending):
                if c == 0:
                    wk.wait()  # stream-level wait, the host does not block
                else:
                    with env.stream_ctx(side[c - 1]):
                        wk.wait()
            for s_c in side:
                main_stream.wait_stream(s_c)

        def free(self):
            for hd in self.handles:
                _lib.group_free(hd)
            self.handles = []

        def full_c(self):
            """the assembled [n, h] result as this rank holds it after a step"""
            return torch.cat([self.gathers[c][r, : self.bounds[r][c + 1] - self.bounds[r][c]]
                              for r in range(world) for c in range(self.K)])

        def describe(self):
            if not multi:
                return "single GPU" if self.K == 1 else f"single GPU, {self.K} row pieces"
            return (f"sp_parts={world} as an nnz-balanced row split, {self.K} piece(s) per rank on their own streams, each "
                    f"all-gathered (RCCL) as soon as it is done")

    class PipelinedRows:
        """sp_parts = world as an nnz-balanced row split, one product per rank and step, and the all-gather (RCCL) of
        step k overlapping the product of step k + 1: two gather buffers, the collective enqueued asynchronously right
        after the product (RCCL's stream waits for the product, the compute stream does not wait for RCCL until the
        buffer comes round again).  Every step's gathered C is complete before the closing fence (drain)."""

        def __init__(self, K):
            self.K = 1
            self.bounds = [split[r:r + 2] for r in range(world)]
            c0, c1 = split[rank], split[rank + 1]
            lo, hi = int(rowptr_cpu[c0]), int(rowptr_cpu[c1])
            rp_c = (rowptr[c0:c1 + 1] - lo).contiguous()
            col_c = col[lo:hi].contiguous()
            self.keep = [rp_c, col_c]
            self.handles = [_lib.group_create(_lib.CSR, _lib.FLT32, [rp_c.data_ptr()], [col_c.data_ptr()], None, [c1 - c0], [n],
                                              [hi - lo], [1], [h], h)]
            self.pad_rows = max(max(split[r + 1] - split[r] for r in range(world)), 1)
            self.bufs = [torch.empty((world, self.pad_rows, h), dtype=torch.float32, device=dev) for _ in range(2)]
            self.pending = [None, None]
            self.k = 0
            self.last = 0
            self.my_rows, self.my_nnz = c1 - c0, hi - lo

        def step(self, exchange=True):
            b = self.k & 1
            self.k += 1
            if self.pending[b] is not None:
                self.pending[b].wait()  # the compute stream waits for the gather that last used this buffer
                self.pending[b] = None
            g = self.bufs[b]
            _lib.spmm_run_group(self.handles[0], [x.data_ptr()], g[rank].data_ptr(), stream)
            if exchange and multi:
                self.pending[b] = dist.all_gather_into_tensor(g.view(-1), g[rank].reshape(-1), async_op=True)
            self.last = b

        def drain(self):
            for b in range(2):
                if self.pending[b] is not None:
                    self.pending[b].wait()
                    self.pending[b] = None

        def free(self):
            self.drain()
            for hd in self.handles:
                _lib.group_free(hd)
            self.handles = []

        def full_c(self):
            g = self.bufs[self.last]
            return torch.cat([g[r, : split[r + 1] - split[r]] for r in range(world)])

        def describe(self):
            return (f"sp_parts={world} as an nnz-balanced row split, one product per rank; the all-gather (RCCL) of step k "
                    f"overlaps the product of step k+1 (two gather buffers), all gathers complete inside the timed region")

    def check_peer(t):
        """a peer's buffer opened through HIP IPC must be directly reachable from this rank's device (xGMI / PCIe P2P);
        otherwise the push candidates are not offered (the exception drops them on every rank)"""
        other = t.device.index
        if other != dev.index and not torch.cuda.can_device_access_peer(dev.index, other):
            raise RuntimeError(f"device {dev.index} has no peer access to device {other}")

    def ipc_exchange(plan, local, reduce_tensor):
        """The local half of a push plan's set-up (its group, its result matrices), then the IPC handle exchange -- in which EVERY rank takes
        part whatever happened to its local half (a rank whose group_create raised used to skip the all_gather_object its peers
        sat in).  A rank that failed sends an error marker; every rank then frees what it made and raises the same error."""
        err, mine = None, None
        try:
            local()
            if multi and world > 1:
                mine = [reduce_tensor(c) for c in plan.C]  # (rebuild function, IPC handle + geometry) per buffer
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {str(e)[:200]}"
        if not (multi and world > 1):
            if err:
                raise RuntimeError(err)
            return
        everyone = [None] * world
        dist.all_gather_object(everyone, ("ERR", err) if err else ("OK", mine))
        bad = [(r, m[1]) for r, m in enumerate(everyone) if m[0] == "ERR"]
        if bad:
            for hd in plan.handles:
                _lib.group_free(hd)
            plan.handles, plan.C = [], []
            raise RuntimeError(f"push plan set-up failed on rank(s) {[r for r, _ in bad]}: {bad[0][1]}")
        for r in range(world):
            if r != rank:
                plan.peer[r] = [fn(*a) for fn, a in everyone[r][1]]
                check_peer(plan.peer[r][0])

    class PushRows:
        """sp_parts = world as an nnz-balanced row split whose exchange is a PUSH over xGMI by the copy engines: every rank
        writes its result block straight into its place in every peer's [N, h] matrix (the peers' buffers are opened through
        HIP IPC once), world - 1 peer copies on their own streams behind the product, and a 4-byte all-reduce as the arrival
        barrier.  The pushes of step k overlap the product of step k + 1 (two result matrices); no CU is spent on the
        exchange and the result needs no re-layout.  xGMI is point-to-point, one link per peer: the copies use all of them."""

        def __init__(self, K):
            from torch.multiprocessing.reductions import reduce_tensor

            self.K = 1
            c0, c1 = split[rank], split[rank + 1]
            self.c0, self.c1 = c0, c1
            lo, hi = int(rowptr_cpu[c0]), int(rowptr_cpu[c1])
            rp_c = (rowptr[c0:c1 + 1] - lo).contiguous()
            col_c = col[lo:hi].contiguous()
            self.keep = [rp_c, col_c]
            self.handles, self.C = [], []

            def local():   # (may raise on THIS rank only: out of memory, say)
                self.handles = [_lib.group_create(_lib.CSR, _lib.FLT32, [rp_c.data_ptr()], [col_c.data_ptr()], None, [c1 - c0], [n],
                                                  [hi - lo], [1], [h], h)]
                self.C = [torch.zeros((n, h), dtype=torch.float32, device=dev) for _ in range(2)]

            self.peer = [[None, None] for _ in range(world)]
            ipc_exchange(self, local, reduce_tensor)
            self.copy_streams = [env.Stream() for _ in range(world)]
            self.sync_stream = env.Stream()
            self.flags = [torch.zeros(1, dtype=torch.float32, device=dev) for _ in range(2)]
            self.pending = [None, None]
            self.k = 0
            self.last = 0
            self.my_rows, self.my_nnz = c1 - c0, hi - lo

        def step(self, exchange=True):
            b = self.k & 1
            self.k += 1
            if self.pending[b] is not None:
                self.pending[b].wait()  # everyone's pushes into this buffer (two steps ago) have landed, mine have left
                self.pending[b] = None
            mine = self.C[b][self.c0:self.c1]
            _lib.spmm_run_group(self.handles[0], [x.data_ptr()], mine.data_ptr(), stream)
            if exchange and multi:
                done = env.Event()
                done.record(main_stream)
                for r in range(world):
                    if r == rank or self.peer[r][b] is None:
                        continue
                    s_r = self.copy_streams[r]
                    s_r.wait_event(done)
                    with env.stream_ctx(s_r):
                        self.peer[r][b][self.c0:self.c1].copy_(mine, non_blocking=True)
                self.sync_stream.wait_event(done)
                for r in range(world):
                    if r != rank:
                        self.sync_stream.wait_stream(self.copy_streams[r])
                with env.stream_ctx(self.sync_stream):
                    self.pending[b] = dist.all_reduce(self.flags[b], async_op=True)
            self.last = b

        def drain(self):
            for b in range(2):
                if self.pending[b] is not None:
                    self.pending[b].wait()
                    self.pending[b] = None

        def free(self):
            self.drain()
            env.synchronize()
            self.peer = None  # close the peers' buffers before anyone frees theirs
            if multi:
                dist.barrier()
            for hd in self.handles:
                _lib.group_free(hd)
            self.handles = []

        def full_c(self):
            return self.C[self.last]

        def describe(self):
            return (f"sp_parts={world} as an nnz-balanced row split, one product per rank; every rank pushes its block into "
                    f"every peer's result matrix (HIP IPC, {max(world - 1, 0)} peer copies on the copy engines over xGMI) "
                    f"behind the next step's product, a 4-byte all-reduce as the arrival barrier")

    class FeaturePieces:
        """ds_parts = world: rank r owns the feature block X[:, r*h/world : (r+1)*h/world] and computes that
        block of C for ALL rows (A replicated), in K nnz-balanced row pieces on their own streams; each piece
        is all-gathered along the feature dimension and laid row-major as soon as it is done."""

        def __init__(self, K):
            self.K = K
            self.hw = h // world
            self.f0 = rank * self.hw
            self.b = nnz_balanced_row_split(rowptr_cpu, K)
            self.C = torch.empty((n, h), dtype=torch.float32, device=dev)
            self.handles, self.gathers, self.keep = [], [], []
            self.my_rows, self.my_nnz = n, nnz
            for c in range(K):
                c0, c1 = self.b[c], self.b[c + 1]
                lo, hi = int(rowptr_cpu[c0]), int(rowptr_cpu[c1])
                rp_c = (rowptr[c0:c1 + 1] - lo).contiguous()
                col_c = col[lo:hi].contiguous()
                self.keep += [rp_c, col_c]
                self.handles.append(_lib.group_create(_lib.CSR, _lib.FLT32, [rp_c.data_ptr()], [col_c.data_ptr()], None,
                                                      [c1 - c0], [n], [hi - lo], [1], [self.hw], self.hw))
                self.gathers.append(torch.empty((world, max(c1 - c0, 1), self.hw), dtype=torch.float32, device=dev))
            self.side = [env.Stream() for _ in range(K - 1)]

        def _piece(self, c, s, exchange=True, x_unchanged=False):
            c0, c1 = self.b[c], self.b[c + 1]
            g = self.gathers[c]
            mine = g[rank]
            # product on the strided feature window of X (row stride h): C_block[rows_c, hw]
            _lib.block_run(self.handles[c], 0, x.data_ptr() + 4 * self.f0, h, mine.data_ptr(), self.hw, self.hw, False,
                           s.cuda_stream, x_unchanged=x_unchanged)
            if not exchange:
                return
            if multi:
                wk = dist.all_gather_into_tensor(g.view(-1), mine.reshape(-1), async_op=True)
                wk.wait()
            if c1 > c0:
                self.C[c0:c1].view(c1 - c0, world, self.hw).copy_(g[:, : c1 - c0].permute(1, 0, 2))

        def step(self, exchange=True):
            self._piece(0, main_stream, exchange)
            if self.K > 1:
                ready = env.Event()
                ready.record(main_stream)
                for c in range(1, self.K):  # same feature window of the same X within this step
                    s_c = self.side[c - 1]
                    s_c.wait_event(ready)
                    with env.stream_ctx(s_c):
                        self._piece(c, s_c, exchange, x_unchanged=True)
                for s_c in self.side:
                    main_stream.wait_stream(s_c)

        def free(self):
            for hd in self.handles:
                _lib.group_free(hd)
            self.handles = []

        def full_c(self):
            return self.C

        def describe(self):
            return (f"ds_parts={world} as a feature split (A replicated, {self.hw} features per rank), {self.K} row piece(s) "
                    f"per rank on their own streams, each all-gathered (RCCL) along the features and laid row-major")

    class PipelinedFeatures:
        """ds_parts = world, pipelined: rank r computes C[:, r*h/world : (r+1)*h/world] for ALL rows in one product per step
        (A replicated; per-rank product 0.83 ms at world = 8 against 1.09 ms for a 1/8 row share: the L2 panels are re-used
        by all the rows); the all-gather (RCCL) of step k runs behind the product of step k + 1, and the gathered
        [world, N, h/world] blocks of step k are laid row-major right after that product.  Two gather buffers; every gather and
        re-layout is complete before the closing fence (drain)."""

        def __init__(self, K):
            self.K = 1
            self.hw = h // world
            self.f0 = rank * self.hw
            self.C = torch.empty((n, h), dtype=torch.float32, device=dev)
            self.handles = [_lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz],
                                              [1], [self.hw], self.hw)]
            self.bufs = [torch.empty((world, n, self.hw), dtype=torch.float32, device=dev) for _ in range(2)]
            self.pending = [None, None]
            self.k = 0
            self.my_rows, self.my_nnz = n, nnz

        def _finish(self, b):
            if self.pending[b] is not None:
                self.pending[b].wait()  # the compute stream waits for that gather, then lays its blocks row-major
                self.pending[b] = None
                self.C.view(n, world, self.hw).copy_(self.bufs[b].permute(1, 0, 2))

        def step(self, exchange=True):
            b = self.k & 1
            self.k += 1
            g = self.bufs[b]
            _lib.block_run(self.handles[0], 0, x.data_ptr() + 4 * self.f0, h, g[rank].data_ptr(), self.hw, self.hw, False, stream)
            if not exchange:
                return
            if multi:
                self.pending[b] = dist.all_gather_into_tensor(g.view(-1), g[rank].reshape(-1), async_op=True)
                self._finish(1 - b)  # the previous step's gather has had this whole product to complete
            else:
                self.C.view(n, world, self.hw).copy_(g.permute(1, 0, 2))

        def drain(self):
            last = (self.k - 1) & 1
            self._finish(1 - last)
            self._finish(last)

        def free(self):
            self.drain()
            for hd in self.handles:
                _lib.group_free(hd)
            self.handles = []

        def full_c(self):
            return self.C

        def describe(self):
            return (f"ds_parts={world} as a feature split (A replicated, {self.hw} features per rank), one product per rank and "
                    f"step; the all-gather (RCCL) of step k overlaps the product of step k+1, then its blocks are laid row-major")

    class PushFeatures:
        """ds_parts = world with the push exchange: rank r computes C[:, f_r] for all rows straight into its own [N, h]
        result matrix (column window, row stride h) and pushes that window into every peer's matrix with one strided (2-D)
        peer copy per peer on its own stream -- the result lands in its final row-major place on every rank, no gather
        buffer, no re-layout, no CU spent on the exchange; 4-byte all-reduce as the arrival barrier; the pushes of step k
        overlap the product of step k + 1."""

        def __init__(self, K):
            import ctypes

            from torch.multiprocessing.reductions import reduce_tensor

            self.K = 1
            self.hw = h // world
            self.f0 = rank * self.hw
            self.hip = ctypes.CDLL("libamdhip64.so")
            # hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, kind, stream)
            self.hip.hipMemcpy2DAsync.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                                  ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
            self.hip.hipMemcpy2DAsync.restype = ctypes.c_int
            self.handles, self.C = [], []

            def local():
                self.handles = [_lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz],
                                                  [1], [self.hw], self.hw)]
                self.C = [torch.zeros((n, h), dtype=torch.float32, device=dev) for _ in range(2)]

            self.peer = [[None, None] for _ in range(world)]
            ipc_exchange(self, local, reduce_tensor)
            self.copy_streams = [env.Stream() for _ in range(world)]
            self.sync_stream = env.Stream()
            self.flags = [torch.zeros(1, dtype=torch.float32, device=dev) for _ in range(2)]
            self.pending = [None, None]
            self.k = 0
            self.last = 0
            self.my_rows, self.my_nnz = n, nnz

        def step(self, exchange=True):
            import ctypes

            b = self.k & 1
            self.k += 1
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None
            win = self.C[b].data_ptr() + 4 * self.f0
            _lib.block_run(self.handles[0], 0, x.data_ptr() + 4 * self.f0, h, win, h, self.hw, False, stream)
            if exchange and multi:
                done = env.Event()
                done.record(main_stream)
                for r in range(world):
                    if r == rank or self.peer[r][b] is None:
                        continue
                    s_r = self.copy_streams[r]
                    s_r.wait_event(done)
                    rc = self.hip.hipMemcpy2DAsync(ctypes.c_void_p(self.peer[r][b].data_ptr() + 4 * self.f0), ctypes.c_size_t(4 * h),
                                                   ctypes.c_void_p(win), ctypes.c_size_t(4 * h), ctypes.c_size_t(4 * self.hw),
                                                   ctypes.c_size_t(n), ctypes.c_int(4), ctypes.c_void_p(s_r.cuda_stream))  # 4 = hipMemcpyDefault
                    if rc != 0:
                        raise RuntimeError(f"hipMemcpy2DAsync to rank {r}: error {rc}")
                self.sync_stream.wait_event(done)
                for r in range(world):
                    if r != rank:
                        self.sync_stream.wait_stream(self.copy_streams[r])
                with env.stream_ctx(self.sync_stream):
                    self.pending[b] = dist.all_reduce(self.flags[b], async_op=True)
            self.last = b

        def drain(self):
            for b in range(2):
                if self.pending[b] is not None:
                    self.pending[b].wait()
                    self.pending[b] = None

        def free(self):
            self.drain()
            env.synchronize()
            self.peer = None
            if multi:
                dist.barrier()
            for hd in self.handles:
                _lib.group_free(hd)
            self.handles = []

        def full_c(self):
            return self.C[self.last]

        def describe(self):
            return (f"ds_parts={world} as a feature split (A replicated, {self.hw} features per rank) computed into place; every "
                    f"rank pushes its column window into every peer's result matrix (HIP IPC, one strided peer copy per peer on "
                    f"the copy engines over xGMI) behind the next step's product, a 4-byte all-reduce as the arrival barrier")


    return {"Pieces": Pieces, "PipelinedRows": PipelinedRows, "PushRows": PushRows, "FeaturePieces": FeaturePieces,
            "PipelinedFeatures": PipelinedFeatures, "PushFeatures": PushFeatures}
