#!/usr/bin/env python3
"""bench.py -- the reference's headline benchmark on MI355X.

Metric (BASELINE.json): SpMM GFLOP/s (+ achieved algorithmic GB/s) on a Reddit-shaped
synthetic CSR (N = 232 965, nnz = 114 615 892, unit weights), h = 256, FLT32.
One "step" = one pass of the hot path: C = A . X through the C ABI
(pygim_spmm_run_group) with A, X and C resident in HBM.

N = 1 : whole graph on the one GPU.
N > 1 : strong scaling of the SAME graph with the full C on every rank at the end of every
        step (what the next GCN layer needs).  Candidates, timed on the node before the
        warm-up (--partition fixes one): row split (sp_parts: nnz-balanced row blocks, the
        reference's partition_by_nnz_csr walk, X replicated) in 1 / 2 / 4 pieces with the
        all-gather of each piece behind the products of the others; the same with the
        all-gather of step k behind the product of step k + 1 ("pipelined"); feature split
        (ds_parts: A replicated, h / N features per rank); and "push" forms of both, where a
        rank's block goes into every peer's result matrix by peer copies (HIP IPC, copy engines)
        with a 4-byte RCCL all-reduce as the arrival barrier (--partition push / push-feature, or
        PYGIM_BENCH_PUSH=1 to let auto time them too; the default auto keeps to RCCL all-gathers).
        All exchanges complete inside the timed region.

Launch: python bench.py --gpus N --steps 20 --warmup 5          (N > 1 without WORLD_SIZE: starts its own N ranks as children)
        python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3
L2_PEAK_TBS = 34.5  # MI355X_MICROARCH.md, L2 section (its measured rate for L2-resident ROW GATHERS is 16.8-18.8 TB/s)
LDS_READ_B32_TBS = 256 * 128 * 2.4e9 / 1e12  # ds_read_b32: 128 B/clk per CU (MI355X_MICROARCH.md, LDS table) x 256 CUs at 2.4 GHz = 78.6


from pygim_amd.bench_plans import nnz_balanced_row_split  # noqa: E402,F401  (tests import it from here)


def cpu_quota():
    """CPUs this process may really use: the cgroup quota when there is one (the pool's GPU boxes show 256 CPUs and grant 16), else
    the CPUs of its affinity mask"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // period)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(rowptr, col, x, args):
    """The oracle's row-parallel loop (kind 'port') on a bounded row sample of the same graph."""
    import oracle

    threads = max(1, min(oracle.max_threads(), cpu_quota()))   # as many threads as the container has CPUs of quota (more only contend)
    nrows = rowptr.numel() - 1 if args.cpu_rows <= 0 else min(args.cpu_rows, rowptr.numel() - 1)
    rp = rowptr[: nrows + 1].cpu().numpy().astype(np.uint32)
    nnz = int(rp[-1])
    cl = col[:nnz].cpu().numpy().astype(np.uint32)
    xh = x.cpu().numpy()
    out = np.zeros((nrows, xh.shape[1]), dtype=xh.dtype)
    oracle.spmm_csr_rowpar(rp, cl, None, xh, nthreads=threads, out=out)  # warm (page-in)
    out[:] = 0
    t0 = time.perf_counter()
    oracle.spmm_csr_rowpar(rp, cl, None, xh, nthreads=threads, out=out)
    dt = time.perf_counter() - t0
    gflops = 2.0 * nnz * xh.shape[1] / dt / 1e9
    # an independent library beside the port (SURVEY.md 8d ii): torch's CSR x dense kernel (MKL) on the same sample
    lib = None
    try:
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a = torch.sparse_csr_tensor(torch.from_numpy(rp.astype(np.int64)), torch.from_numpy(cl.astype(np.int64)),
                                        torch.ones(nnz, dtype=torch.float32), size=(nrows, xh.shape[0]))
        xt = torch.from_numpy(xh)
        y = a @ xt  # warm
        t1 = time.perf_counter()
        y = a @ xt
        dl = time.perf_counter() - t1
        lib = {"kind": "torch.sparse_csr @ dense (MKL)", "value": round(2.0 * nnz * xh.shape[1] / dl / 1e9, 3), "unit": "GFLOP/s",
               # (threads the pool was set to, and what the container's quota lets run at once: the second is what the rate reflects)
               "threads": torch.get_num_threads(), "threads_effective": min(torch.get_num_threads(), cpu_quota()), "seconds": round(dl, 3),
               "equal_to_port": bool(np.array_equal(y.numpy(), out))}
    except Exception as e:  # not every build has the CSR kernels
        lib = {"kind": "torch.sparse_csr @ dense (MKL)", "error": str(e)[:120]}
    # the REFERENCE's own host loop (spmm_grande/spmm_mul_csr.c:119-136, compiled in place into oracle/_ref, one thread: row -> feature ->
    # entry, as written) on the first rows -- the port's result there must be its bytes
    ref_leg = None
    try:
        if oracle.have_ref_host():
            nr_ref = int(min(nrows, 1500))
            t2 = time.perf_counter()
            y_ref = oracle.ref_spmm_host_csr(rp[: nr_ref + 1], cl[: int(rp[nr_ref])], None, xh, variant="grande")
            dr = time.perf_counter() - t2
            ref_leg = {"kind": "reference spmm_host_csr (spmm_grande/spmm_mul_csr.c:119-136), 1 thread", "rows": nr_ref,
                       "value": round(2.0 * int(rp[nr_ref]) * xh.shape[1] / dr / 1e9, 3), "unit": "GFLOP/s", "seconds": round(dr, 3),
                       "port_bit_identical": bool(y_ref.tobytes() == out[:nr_ref].tobytes())}
    except Exception as e:  # noqa: BLE001
        ref_leg = {"error": str(e)[:120]}
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # "cores" = the threads the loop ran with; what they share is the container's CPU quota, stated beside it
    return {"value": round(gflops, 3), "unit": "GFLOP/s", "cores": threads, "cpu_quota_cpus": cpu_quota(), "kind": "port", "cpu_model": model,
            "omp_num_threads": os.environ.get("OMP_NUM_THREADS", "unset"), "library": lib, "reference_1thread": ref_leg,
            "sample": f"rows [0,{nrows}) of the same graph ({nnz} nnz, h={xh.shape[1]}), "
                      f"{dt:.2f} s wall, oracle row-parallel CSR loop, os.cpu_count()={os.cpu_count()}"}, out


def self_launch(n_gpus):
    """`python bench.py --gpus N` as a plain command (pygim_amd/launch.py: the N ranks as CHILD processes of torch.distributed.run)"""
    from pygim_amd.launch import self_launch as _launch

    return _launch(os.path.abspath(__file__), n_gpus, sys.argv[1:], threads_per_rank=max(1, cpu_quota() // max(n_gpus, 1)), tag="bench")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # 1.3 s of GPU time: long enough for an outside observer (SMI) to see it
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--shape", default="reddit")
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--clustered", action="store_true", help="columns near the row id instead of uniform")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the CPU-baseline sample (0 = the whole graph, about 6 s on 128 threads)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the clustered-columns measurement beside the headline")
    ap.add_argument("--partition", default="auto", choices=["auto", "row", "feature", "pipelined", "pipelined-feature", "push", "push-feature"])
    ap.add_argument("--chunks", type=int, default=0, help="row pieces per rank (0 = 1 on one GPU; on N > 1 the fastest of 1 / 2 / 4, measured before the warm-up)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_latest.json"))
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    from pygim_amd import _lib, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        # a rank that hangs (a collective that never completes, a teardown that blocks) says where and ends instead of waiting for ever
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ.get("PYGIM_RANK_TIMEOUT", "1800")), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    # (PYGIM_BENCH_BACKEND=gloo with several ranks on one GPU is a logic check of the N>1 path only)
    backend = os.environ.get("PYGIM_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist

    # PYGIM_FORCE_COLLECTIVES=1 under torch.distributed.run with ONE rank: the process group is created and every
    # collective of the N > 1 path really executes (RCCL at world size 1) -- the first-run check of that path on one GPU
    force = os.environ.get("PYGIM_FORCE_COLLECTIVES", "0") == "1" and "MASTER_ADDR" in os.environ
    multi = world > 1 or force
    if multi:
        # a collective that never completes (a peer died, a peer is stuck in setup) fails after this many seconds instead of the
        # library default of ten minutes or more; the per-rank watchdog above is the second line
        import datetime

        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("PYGIM_COLLECTIVE_TIMEOUT", "600")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)
    rccl_world = dist.get_world_size() if multi else 0

    n, nnz, d_max = synth.SHAPES[args.shape]
    h = args.hidden
    # every rank builds the same seeded graph and features on its own device.  Ranks that SHARE a device (the N > 1 logic checks on one
    # GPU, scripts/check_multirank.sh: never a measurement) would run world x the 115 M-entry generation on it at once -- in round 4 that
    # alone starved the row / feature runs past their time-outs: there rank 0 generates, the others read its arrays from /dev/shm
    shared_dev = multi and torch.cuda.device_count() < world
    if shared_dev:
        shm = os.path.join("/dev/shm", f"pygim_bench_{os.environ.get('MASTER_PORT', '0')}_{args.shape}_{int(args.clustered)}")
        if rank == 0:
            rowptr, col = synth.make_csr(n, nnz, d_max, seed=0, device=dev, clustered=args.clustered)
            os.makedirs(shm, exist_ok=True)
            np.save(os.path.join(shm, "rowptr.npy"), rowptr.cpu().numpy())
            np.save(os.path.join(shm, "col.npy"), col.cpu().numpy())
        dist.barrier()
        if rank != 0:
            rowptr = torch.from_numpy(np.load(os.path.join(shm, "rowptr.npy"))).to(dev)
            col = torch.from_numpy(np.load(os.path.join(shm, "col.npy"))).to(dev)
        dist.barrier()
        if rank == 0:
            import shutil

            shutil.rmtree(shm, ignore_errors=True)
    else:
        rowptr, col = synth.make_csr(n, nnz, d_max, seed=0, device=dev, clustered=args.clustered)
    x = synth.features(n, h, torch.float32, seed=0, device=dev)
    torch.cuda.synchronize()

    _lib.init_ranks(world)
    if world > 1:
        # every rank builds its plans with its share of the host's cores: the schedule builder starts one thread per core by default,
        # which on an 8-rank node would be 8 x all cores at once, each rank holding its blobs
        _lib.set_tunable("lds_threads", max(1, cpu_quota() // world))

    class TimedLib:
        """pygim_amd._lib with group creation timed (what a rank paid for its plans, printed per rank below)"""

        def __init__(self, real):
            self._real, self.create_ms, self.created = real, 0.0, 0

        def __getattr__(self, name):
            return getattr(self._real, name)

        def group_create(self, *a, **kw):
            torch.cuda.synchronize()
            t_c0 = time.perf_counter()
            hd = self._real.group_create(*a, **kw)
            torch.cuda.synchronize()
            self.create_ms += (time.perf_counter() - t_c0) * 1e3
            self.created += 1
            return hd

    tlib = TimedLib(_lib)
    rowptr_cpu = rowptr.cpu()
    split = nnz_balanced_row_split(rowptr_cpu, world)
    main_stream = torch.cuda.current_stream()
    stream = main_stream.cuda_stream

    from types import SimpleNamespace

    from pygim_amd import bench_plans

    env = SimpleNamespace(world=world, rank=rank, multi=multi, dev=dev, n=n, nnz=nnz, h=h, x=x, rowptr=rowptr, col=col,
                          rowptr_cpu=rowptr_cpu, split=split, main_stream=main_stream, stream=stream, lib=tlib, dist=dist,
                          Stream=lambda: torch.cuda.Stream(device=dev), Event=torch.cuda.Event, stream_ctx=torch.cuda.stream,
                          synchronize=torch.cuda.synchronize)
    plans = bench_plans.build(env)
    Pieces, PipelinedRows, PushRows = plans["Pieces"], plans["PipelinedRows"], plans["PushRows"]
    FeaturePieces, PipelinedFeatures, PushFeatures = plans["FeaturePieces"], plans["PipelinedFeatures"], plans["PushFeatures"]

    live = []  # plans whose asynchronous exchanges must be complete at a fence

    def fence():
        for pl in live:
            if hasattr(pl, "drain"):
                pl.drain()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # partition and number of pieces: fixed by --partition / --chunks, else a single piece on one GPU and,
    # on N > 1, whichever candidate is fastest on THIS node (measured before the warm-up; all ranks agree
    # through a max-reduce): row split (sp_parts) in 1 / 2 / 4 pieces, feature split (ds_parts) in 1 / 2
    feat_ok = h % world == 0 and (h // world) * 4 >= 32
    from pygim_amd import autotune

    if world > 1:
        # a rank's row share is a few tall tiles: let the LDS-staged plan split them into column ranges (S x the workgroups, each
        # landing 1 / S of X; a 1/8 share of this graph: 1.08 -> 0.52 ms, profiles/r03_exp_colsplit.txt).  For FLT32 a row's sum is
        # then the sum of its ranges' sums (the norm-wise contract); the N = 1 line never splits.
        _lib.set_tunable("lds_col_split_f32", 1)
        autotune.LDS_COL_SPLIT = True

    prior, table = autotune.choose(n, n, nnz, h, 4, max(world, 1))
    prior_1, _ = autotune.choose(n, n, nnz, h, 4, 1)
    cands = []
    if args.partition == "pipelined":
        cands.append((PipelinedRows, 1))
    if args.partition == "pipelined-feature" and feat_ok:
        cands.append((PipelinedFeatures, 1))
    if args.partition == "push":
        cands.append((PushRows, 1))
    if args.partition == "push-feature" and feat_ok:
        cands.append((PushFeatures, 1))
    if args.partition == "row":
        cands += [(Pieces, k) for k in ((args.chunks,) if args.chunks > 0 else ((1,) if not multi else (1, 2, 4)))]
        if multi and args.chunks == 0:
            cands.append((PipelinedRows, 1))
    if args.partition == "feature" and feat_ok:
        cands += [(FeaturePieces, k) for k in ((args.chunks,) if args.chunks > 0 else (1, 2))]
    if args.partition == "auto":
        if not multi:
            cands.append((Pieces, max(args.chunks, 1)))
        else:
            # the chooser's model (pygim_amd/autotune.py, validated against measured per-rank shares in
            # tests/test_autotune_gpu.py) ranks the two families; the two best arrangements of the preferred family and the
            # best of the other are timed on the node (3 candidates instead of 6)
            kk = (args.chunks,) if args.chunks > 0 else (1,)
            row_first = prior.feat_parts == 1 or not feat_ok or world == 1
            fam_row = [(PipelinedRows, 1), (PushRows, 1)] if args.chunks == 0 else [(Pieces, kk[0])]
            fam_feat = [(PushFeatures, 1), (PipelinedFeatures, 1)] if feat_ok and world > 1 else []
            if os.environ.get("PYGIM_BENCH_PUSH", "0") != "1":
                # default: RCCL collectives only.  The push exchange (HIP IPC peer copies) has run between processes on ONE GPU
                # only -- no multi-GPU box in the build -- so it is offered by --partition push / push-feature or PYGIM_BENCH_PUSH=1,
                # not timed blind inside the driver's scaling run
                fam_row = [(PipelinedRows, 1), (Pieces, 2)] if args.chunks == 0 else fam_row
                fam_feat = [(PipelinedFeatures, 1), (FeaturePieces, 2)] if fam_feat else []
                cands = (fam_row + fam_feat[:1]) if row_first else (fam_feat + fam_row[:1])
            else:
                cands = (fam_row + fam_feat[:1]) if row_first else (fam_feat + fam_row[1:2])
    assert cands, "no admissible partition"
    timed = {}
    fail_rank = int(os.environ.get("PYGIM_BENCH_FAIL_RANK", "-1"))   # (test hook: this rank raises while setting a candidate up)
    if True:
        best = None
        for cls, kk in cands:
            # a candidate that cannot be set up on some rank (e.g. IPC refused, out of memory) is dropped on ALL ranks: every rank
            # reports a status word and the MIN decides -- no rank sits in a fence while another has raised.  Constructors WITH a
            # collective inside (the push plans' IPC handle exchange) take part in it whatever their local half did and raise the
            # same error on every rank (bench_plans.ipc_exchange).  The only candidate failing ends every rank with the same non-zero exit
            try:
                if rank == fail_rank:
                    raise RuntimeError("PYGIM_BENCH_FAIL_RANK: forced set-up failure on this rank")
                pl = cls(kk)
                okf = 1
            except Exception as e:  # noqa: BLE001
                pl, okf = None, 0
                print(f"[bench] rank {rank}: candidate {cls.__name__} not available: {str(e)[:200]}", file=sys.stderr)
            if multi:
                agree = torch.tensor([okf], dtype=torch.int32, device=dev)
                dist.all_reduce(agree, op=dist.ReduceOp.MIN)
                okf = int(agree.item())
            if not okf:
                if pl is not None:
                    pl.free()
                timed[f"{cls.__name__}:{kk}"] = None
                continue
            live[:] = [pl]
            # a candidate whose exchange fails at run time on every rank alike (a refused peer copy, say) is dropped as well
            try:
                for _ in range(2):
                    pl.step()
                fence()
                t_c = time.perf_counter()
                for _ in range(4):
                    pl.step()
                fence()
                t_c = time.perf_counter() - t_c
            except RuntimeError as e:
                t_c = float("inf")
                print(f"[bench] rank {rank}: candidate {cls.__name__} failed while running: {str(e)[:200]}", file=sys.stderr)
            tt = torch.tensor([t_c], dtype=torch.float64, device=dev)
            if multi:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            if not math.isfinite(float(tt.item())):
                live[:] = []
                try:
                    pl.free()
                except RuntimeError:
                    pass
                timed[f"{cls.__name__}:{kk}"] = None
                continue
            if len(cands) > 1:
                timed[f"{cls.__name__}:{kk}"] = round(float(tt.item()) / 4 * 1e3, 4)
            if best is None or float(tt.item()) < best[0]:
                if best is not None:
                    best[1].free()
                best = (float(tt.item()), pl)
            else:
                pl.free()
            del pl
        if best is None:
            if multi:
                dist.destroy_process_group()
            raise SystemExit(f"[bench] rank {rank}: no candidate could be set up on every rank ({[c.__name__ for c, _ in cands]})")
        plan = best[1]
    live[:] = [plan]
    K, step = plan.K, plan.step
    handles = plan.handles

    def kernel_family(hd):
        """which kernel family a group's products take, as the library reports it (never re-derived here)"""
        lp, lc = _lib.group_lds_plan(hd), _lib.group_lds_code(hd)
        if lp["tiles"] > 0 and lc["active"]:
            g = _lib.group_lds_geometry(hd)
            return {"kernel": "k_lds_code8_f32" if g["waves"] == 8 else "k_lds_code_f32", "code_bytes": lc["code_bytes"], "tiles": lp["tiles"],
                    "col_splits": g["col_splits"], "ring": f"{g['buffers']} x {g['chunk_cols']}", "note": _lib.group_lds_note(hd)}
        if lp["tiles"] > 0:
            return {"kernel": "k_lds_spmm_f32 (token form)", "code_bytes": 0, "tiles": lp["tiles"], "col_splits": _lib.group_lds_geometry(hd)["col_splits"],
                    "note": _lib.group_lds_note(hd)}
        return {"kernel": "k_csr_panel (L2 sweep)", "code_bytes": 0, "tiles": 0, "col_splits": 1, "note": _lib.group_lds_note(hd)}

    fams = [kernel_family(hd) for hd in handles]
    mine = {"rank": rank, "groups_created": tlib.created, "group_create_ms": round(tlib.create_ms, 1), "plan_threads": max(1, cpu_quota() // world) if world > 1 else f"all ({cpu_quota()} CPUs of quota)",
            "groups": fams}
    print(f"[bench] rank {rank}: {json.dumps(mine)}", file=sys.stderr, flush=True)
    per_rank = [mine]
    if multi:
        per_rank = [None] * dist.get_world_size()
        dist.all_gather_object(per_rank, mine)
    split_tiles = any(f["col_splits"] > 1 for pr in per_rank for f in pr["groups"])
    headline_warning = None
    if world == 1 and not force and (args.shape, h) == ("reddit", 256) and not os.environ.get("PYGIM_TUNE") and not fams[0]["kernel"].startswith("k_lds_code"):
        # the group fell down the ladder (pygim_group_lds_note says why): the line below names the kernel that really ran (the roofline
        # object takes the family from the library), says so in config.headline_kernel_warning, and PYGIM_BENCH_STRICT=1 makes it fatal
        headline_warning = f"the headline group is NOT on the code-stream kernel but on {fams[0]['kernel']}: {fams[0]['note']}"
        print(f"[bench] WARNING: {headline_warning}", file=sys.stderr, flush=True)
        if os.environ.get("PYGIM_BENCH_STRICT", "0") == "1":
            raise SystemExit(f"[bench] {headline_warning}")
    for hd in handles:
        _lib.group_kernel_events(hd, True)  # HIP events around the dominant kernel of every product, on its launch stream
    my_rows, my_nnz = plan.my_rows, plan.my_nnz
    my_h = getattr(plan, "hw", h)

    for _ in range(args.warmup):
        step()
    fence()
    for hd in handles:
        _lib.group_kernel_ms(hd, reset=True)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record(main_stream)
        step()
    marks[args.steps].record(main_stream)
    fence()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if step_ms else 0.0
    k_ms_sum, k_count = 0.0, 0
    for hd in handles:
        ms_c, cnt_c = _lib.group_kernel_ms(hd, reset=True)
        k_ms_sum += ms_c
        k_count = max(k_count, cnt_c)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    # for reading the scaling: the same step without the exchange (each rank's products only), outside the timed region
    products_only_ms = None
    if multi:
        fence()
        t1 = time.perf_counter()
        for _ in range(3):
            step(exchange=False)
        fence()
        tp = torch.tensor([(time.perf_counter() - t1) / 3 * 1e3], dtype=torch.float64, device=dev)
        dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        products_only_ms = round(float(tp.item()), 4)
        step()  # leave the gathered C of a full step behind for the check below
        fence()
        for hd in handles:
            _lib.group_kernel_ms(hd, reset=True)
    total_flops = synth.flops(nnz, h)
    gflops = total_flops / (ms_per_step * 1e-3) / 1e9

    # roofline of the dominant kernel on THIS rank's block: HIP events (on the launch stream) bracket
    # the panel launches of each product; kernel_ms = their sum per product
    alg_bytes = synth.algorithmic_bytes(my_rows, n, my_nnz, my_h, 4, "CSR", with_values=True)
    k_ms = k_ms_sum / max(k_count, 1)
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    # HBM traffic per product: NOT measured in this run (PMC passes need rocprofv3); it is replayed from the committed
    # summary of the last counter run of this same command, and labelled as such
    traffic, traffic_source = None, None
    if world == 1 and K == 1 and os.path.exists(args.traffic_json):
        try:
            tj = json.load(open(args.traffic_json))
            traffic = tj.get("hbm_bytes_per_product")
            lds_now = _lib.group_lds_plan(handles[0])["tiles"] > 0
            if lds_now != ("k_lds_spmm" in str(tj.get("kernel")) or "k_lds_code" in str(tj.get("kernel"))):
                traffic = None  # the committed counter run is of the other kernel family: nothing to replay
            traffic_source = {"replayed_from": os.path.relpath(args.traffic_json, ROOT), "collected": tj.get("collected"),
                              "box": tj.get("box"), "command": tj.get("command"), "kernel_ms_then": tj.get("kernel_ms")}
            # a counter run of another kernel than the one just timed (its time differs by more than 10 %) is not replayed
            then = tj.get("kernel_ms")
            if traffic is not None and then and k_ms and abs(k_ms - then) / then > 0.10:
                traffic_source["not_replayed"] = f"the counter run's kernel took {then:.3f} ms, this run's {k_ms:.3f} ms: more than 10 % apart"
                traffic = None
        except Exception:
            traffic = None
    plan_info = _lib.group_plan(handles[0])
    lds_info = _lib.group_lds_plan(handles[0])
    n_panels = int(plan_info["n_panels"])
    gather = synth.gather_bytes(my_rows, my_nnz, my_h, 4)
    if lds_info["tiles"] > 0 and my_h >= 33:
        # LDS-staged product (pygim_amd/csrc/lds_kernel_gen.hpp): one launch; every stored entry reads its 256-byte row slice
        # of X from LDS (ds_read_b32), the chunks of X are staged L2 -> LDS once per (tile, slice)
        nsl = (my_h + 63) // 64
        # (plans with long slots -- >= 128 tokens per wave and chunk -- take the 16-token-batch form of the kernel, pygim_hip.hip lds_long_slots)
        long_slots = lds_info["chunk_fills"] > 0 and lds_info["tokens"] / (lds_info["chunk_fills"] * 16) >= 128
        code_info = _lib.group_lds_code(handles[0])
        geo = _lib.group_lds_geometry(handles[0])    # waves, accumulators per wave, chunk columns, ring buffers ... as the library planned them
        chunk_bytes = geo["chunk_cols"] * 256
        if code_info["active"]:   # the schedule compiled into machine code (k_lds_code_* / k_lds_code8_*)
            kfam = "k_lds_code8_f32" if geo["waves"] == 8 else "k_lds_code_f32"
            kname = (f"k_slice_pack<float,4,4> + {kfam} (ONE launch per product: {lds_info['tiles']} row tiles x {nsl} slices, {geo['waves']} waves x "
                     f"{geo['acc_per_wave']} accumulators, LDS ring {geo['buffers']} x {geo['chunk_cols']} columns; "
                     f"{code_info['code_bytes'] / 1e9:.2f} GB of generated gfx950 code per graph)")
        else:
            kname = (f"k_slice_pack<float,4,4> + k_lds_spmm_f32_w16{'b' if long_slots else ''} (ONE launch per product: {lds_info['tiles']} row tiles x "
                     f"{nsl} slices)")
        launches = 1
        staged = lds_info["chunk_fills"] * chunk_bytes * nsl
        # the LDS array is what the code-stream kernel keeps busiest: 2 cycles per 256-byte ds_read (one per stored entry that does not
        # share a neighbour's read) + the DMA's writes at ~128 B/clk (scripts/micro/fillrate.hip), per CU at the clock below
        lds_reads = (lds_info["nnz"] - geo["shared_entries"]) * nsl
        lds_cycles = lds_reads * 2 + staged / 128
        on_chip = {"level": "LDS (one 256-byte row slice per stored entry and slice: ds_read_b32, or ds_read2st64_b32 for two entries)", "gather_bytes": gather,
                   "achieved_TBs": round(gather / (k_ms * 1e-3) / 1e12, 2) if k_ms else None, "peak_TBs": round(LDS_READ_B32_TBS, 1),
                   "frac": round(gather / (k_ms * 1e-3) / 1e12 / LDS_READ_B32_TBS, 4) if k_ms else None,
                   "geometry": geo,
                   "staged_L2_to_LDS_bytes": staged,
                   "staged_TBs": round(staged / (k_ms * 1e-3) / 1e12, 2) if k_ms else None,
                   "lds_reads": lds_reads, "entries_sharing_a_read": geo["shared_entries"] * nsl,
                   "lds_array_cycles_per_cu": round(lds_cycles / 256),
                   "lds_array_busy_frac_at_2.4GHz": round(lds_cycles / 256 / (k_ms * 1e-3 * 2.4e9), 3) if k_ms else None,
                   "tokens_incl_padding": lds_info["tokens"],
                   **({"code_stream": code_info, "instruction_fetch_bytes": code_info["code_bytes"] * nsl,
                       # bytes the L2 hands to the CUs (chunks into LDS + the code, each fetched once per slice) against what 256 CUs land
                       # with a DMA request stream that never dries up (scripts/micro/fillrate.hip, round 4: 113 GB/s per CU = 29 TB/s;
                       # round 3's two-buffer ring drained at every slot boundary: 81 GB/s per CU alone, 17.7 TB/s under the product)
                       "l2_read_TBs": round((staged + code_info["code_bytes"] * nsl) / (k_ms * 1e-3) / 1e12, 2) if k_ms else None,
                       "l2_read_ceiling_measured_TBs": 29.0,
                       "l2_read_frac": round((staged + code_info["code_bytes"] * nsl) / (k_ms * 1e-3) / 1e12 / 29.0, 3) if k_ms else None}
                      if code_info["active"] else {})}
    else:
        # template arguments: <T, VEC, LOG_LPR, AMODE, HAS_VALS, DEQ>; AMODE 3 = 128-byte slice-major rows + 16-bit panel-local ids
        amode = 3 if plan_info["col16"] else 2
        kname = (f"k_slice_pack + k_csr_panel<float,4,3,{amode},false,false> x {n_panels} panel launches per product" if n_panels
                 else "k_csr_wide<float,4>")
        launches = max(n_panels, 1)
        # the sweep pulls one 128-byte line per (stored entry, slice) out of the L2: a RANDOM-LINE GATHER rate, to be read against
        # the guide's measured 16.8-18.8 TB/s for L2-resident row gathers, not against the L2's streaming peak
        on_chip = {"level": "L2 -> L1 random-line gathers", "gather_bytes": gather,
                   "achieved_TBs": round(gather / (k_ms * 1e-3) / 1e12, 2) if k_ms else None,
                   "peak_TBs": L2_PEAK_TBS, "frac": round(gather / (k_ms * 1e-3) / 1e12 / L2_PEAK_TBS, 4) if k_ms else None,
                   "measured_gather_ceiling_TBs": "16.8-18.8 (MI355X_MICROARCH.md, indexed rows from the XCD's L2)"}
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                "kernel": kname, "kernel_ms": round(k_ms, 4), "products_timed": k_count,
                "launches_per_product": launches,
                "algorithmic_bytes": alg_bytes,
                "gather_model_GBs": round(gather / (k_ms * 1e-3) / 1e9, 1) if k_ms else None,
                "fp32_frac": round(synth.flops(my_nnz, my_h) / (k_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 5) if k_ms else None,
                "on_chip": on_chip}

    result = {
        "metric": ("SpMM GFLOP/s, Reddit-shaped CSR h=256 fp32" if (args.shape, h) == ("reddit", 256) else
                   f"SpMM GFLOP/s, {args.shape}-shaped CSR h={h} fp32"), "value": round(gflops, 2), "unit": "GFLOP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "ms_per_step_median": round(median_ms, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.shape}-shaped CSR SpMM" + (" (configs[1])" if (args.shape, h) == ("reddit", 256) else " (not the headline configuration)"), "N": n, "nnz": nnz, "h": h,
                   "columns": "clustered" if args.clustered else "uniform",
                   "partition": plan.describe(), "candidate": f"{type(plan).__name__}:{plan.K}",
                   "candidates_timed_ms": timed, "rccl_world": rccl_world, "backend": backend if multi else None,
                   # the chooser's forecast beside the measurement (pygim_amd/autotune.py: a rank's product + the exchange that re-assembles
                   # C over xGMI; its link terms have never met real xGMI): the candidate it prefers, every (row parts x feature parts)
                   # grid it priced, and the speed-up over its own 1-GPU price -- a measured SCALE run compares with THIS at once
                   "model_prior": {"row_parts": prior.row_parts, "feat_parts": prior.feat_parts,
                                   "predicted_ms": round(prior.seconds * 1e3, 4),
                                   "predicted_product_ms": round(prior.product_s * 1e3, 4), "predicted_collective_ms": round(prior.collective_s * 1e3, 4),
                                   "predicted_1gpu_ms": round(prior_1.seconds * 1e3, 4),
                                   "predicted_speedup_vs_1gpu": round(prior_1.seconds / prior.seconds, 3) if prior.seconds > 0 else None,
                                   # the ranks' products alone (what one GPU can measure, profiles/r06_exp_shard.txt); the rest of the step is the exchange over xGMI
                                   "predicted_products_only_speedup": round(prior_1.product_s / prior.product_s, 3) if prior.product_s > 0 else None,
                                   "grids_priced_ms": {f"{c.row_parts}x{c.feat_parts}": round(c.seconds * 1e3, 4) for c in table}},
                   "per_rank": per_rank,
                   **({"headline_kernel_warning": headline_warning} if headline_warning else {}),
                   **({"ms_per_step_products_only": products_only_ms} if products_only_ms is not None else {})},
        "roofline": roofline,
    }

    e2e_host = None
    if rank == 0 and world == 1 and not args.clustered and not args.no_extra and (args.shape, h) == ("reddit", 256):
        # the reference driver's default call (CPU tensors in and out), FIRST among the extras: what spmm_test.py does is this call in a process that has done
        # little else, and the DMA engines the runtime hands the call's copy streams depend on what the process did before (profiles/r06_exp_host_call.txt:
        # 8.2 ms in a fresh process; after the BASELINE-config legs below the pitched downloads run at 21 GB/s and the group turns to direct stores: 9.0)
        from pygim_amd import bench_configs as _bc

        try:
            e2e_host = _bc.end_to_end_cpu_tensors(rowptr, col, n, h)
        except Exception as e:  # noqa: BLE001  (an extra: never the reason the line is missing)
            e2e_host = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    if rank == 0 and world == 1 and not args.clustered and not args.no_extra:
        # SURVEY.md 8(d) asks for both column shapes: the same product on the community-like variant of the graph (columns within
        # ~1 % of the row id), measured here OUTSIDE the timed region and reported beside the headline
        rp_c, col_c = synth.make_csr(n, nnz, d_max, seed=0, device=dev, clustered=True)
        torch.cuda.synchronize()
        t_create = time.perf_counter()
        hd_c = _lib.group_create(_lib.CSR, _lib.FLT32, [rp_c.data_ptr()], [col_c.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
        torch.cuda.synchronize()
        t_create = time.perf_counter() - t_create   # one-time: sweep + LDS-staged plans of a graph of this size (device-resident CSR in)
        out_c = torch.empty((n, h), dtype=torch.float32, device=dev)
        for _ in range(3):
            _lib.spmm_run_group(hd_c, [x.data_ptr()], out_c.data_ptr(), stream)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
        for i in range(10):
            ev[i].record(main_stream)
            _lib.spmm_run_group(hd_c, [x.data_ptr()], out_c.data_ptr(), stream)
        ev[10].record(main_stream)
        torch.cuda.synchronize()
        ts_c = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10))
        cc = torch.bincount(col_c.long(), minlength=n).double()
        # creation cost of the HEADLINE group (the reference's to_device / prepare step, paid once per graph; spmm_default/spmm_mul_csr.c:118-330)
        # and how many products it takes to earn it back against the next forms down the ladder -- MEASURED here on the same graph
        # (ADVICE r04): the token kernels (lds_code = 0: no code generation, no executable memory) and the L2 sweep (lds_mode = 2)
        create_ms = mine["group_create_ms"]
        # the same group once more in this process (the first creation of a process also loads the library's code objects and grows the allocator's
        # pools: what a long-lived server pays per graph is this second figure)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hd_again = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
        torch.cuda.synchronize()
        create_again_ms = round((time.perf_counter() - t0) * 1e3, 1)
        _lib.group_free(hd_again)
        alts = {}
        for name, knob, val in (("token_kernels", "lds_code", 0), ("sweep", "lds_mode", 2)):
            prev = _lib.set_tunable(knob, val)
            try:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                hd_a = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
                torch.cuda.synchronize()
                t_a = (time.perf_counter() - t0) * 1e3
            finally:
                _lib.set_tunable(knob, prev)
            out_a = torch.empty((n, h), dtype=torch.float32, device=dev)
            for _ in range(2):
                _lib.spmm_run_group(hd_a, [x.data_ptr()], out_a.data_ptr(), stream)
            torch.cuda.synchronize()
            ea = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
            for i in range(5):
                ea[i].record(main_stream)
                _lib.spmm_run_group(hd_a, [x.data_ptr()], out_a.data_ptr(), stream)
            ea[5].record(main_stream)
            torch.cuda.synchronize()
            ms_a = sorted(ea[i].elapsed_time(ea[i + 1]) for i in range(5))[2]
            gain = ms_a - ms_per_step
            alts[name] = {"ms_per_product": round(ms_a, 3), "group_create_ms": round(t_a, 1), "lds_note": _lib.group_lds_note(hd_a),
                          "break_even_products": (max(0, math.ceil((create_ms - t_a) / gain)) if gain > 0 else None)}
            _lib.group_free(hd_a)
            del out_a
        result["extra_headline"] = {
            "group_create_ms": create_ms, "group_create_ms_again": create_again_ms, "code_bytes": fams[0]["code_bytes"], "lds_note": fams[0]["note"],
            "code_generated_on_device": bool(_lib.group_lds_code(handles[0]).get("device_generated")),
            "alternatives_measured": alts,
            "note": "group_create_ms = pygim_group_create of the timed group (device-resident CSR in): validation, the sweep's plan beside it, and the code stream "
                    "(round 5: generated on the device from the resident CSR, byte-identical to the host encoder); break_even_products = products after which "
                    "this group's creation + products cost less than the alternative's (0 = from the first product)"}
        result["extra"] = {"clustered_ms_per_step": round(ts_c[len(ts_c) // 2], 4),
                           "clustered_GFLOPs": round(total_flops / (ts_c[len(ts_c) // 2] * 1e-3) / 1e9, 1),
                           "clustered_lds_plan": _lib.group_lds_plan(hd_c),
                           "group_create_ms": round(t_create * 1e3, 1),
                           "clustered_check": "column-count checksum exact" if torch.equal(out_c.double().sum(0), cc @ x.double()) else "MISMATCH",
                           "note": "same N / nnz / degrees, columns within ~1 % of the row id (synth.make_csr clustered=True); 10 steps, median, outside the timed region; group_create_ms = the one-time plan build (the reference's to_device / prepare step) for it"}
        _lib.group_free(hd_c)
        del rp_c, col_c, out_c
        # a graph WITH structure and arbitrary node ids beside the two above (round 5, profiles/r05_structured.txt): a stochastic block model of the
        # same shape (50 communities, 80 % of a row's entries inside its own), ids shuffled -- the library finds the communities by label
        # propagation and builds its tiles from them (lds_tile_order, automatic)
        try:
            rp_s, col_s = synth.make_shape(args.shape, seed=0, device=dev, kind="sbm")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hd_s = _lib.group_create(_lib.CSR, _lib.FLT32, [rp_s.data_ptr()], [col_s.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
            torch.cuda.synchronize()
            t_create_s = time.perf_counter() - t0
            out_s = torch.empty((n, h), dtype=torch.float32, device=dev)
            for _ in range(2):
                _lib.spmm_run_group(hd_s, [x.data_ptr()], out_s.data_ptr(), stream)
            torch.cuda.synchronize()
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
            for i in range(10):
                evs[i].record(main_stream)
                _lib.spmm_run_group(hd_s, [x.data_ptr()], out_s.data_ptr(), stream)
            evs[10].record(main_stream)
            torch.cuda.synchronize()
            ts_s = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(10))
            cs = torch.bincount(col_s.long(), minlength=n).double()
            geo_s = _lib.group_lds_geometry(hd_s)
            result["extra"]["sbm_shuffled_ids"] = {
                "ms_per_step": round(ts_s[len(ts_s) // 2], 4), "tiles": _lib.group_lds_tiles(hd_s), "lds_plan": _lib.group_lds_plan(hd_s),
                "entries_sharing_a_read_frac": round(geo_s["shared_entries"] / max(nnz, 1), 3), "group_create_ms": round(t_create_s * 1e3, 1),
                "check": "column-count checksum exact" if torch.equal(out_s.double().sum(0), cs @ x.double()) else "MISMATCH",
                "note": "stochastic block model of the same N / nnz / degrees with shuffled node ids (pygim_amd/synth.py make_sbm); 10 steps, median, outside the timed region"}
            _lib.group_free(hd_s)
            del rp_s, col_s, out_s
        except Exception as e:  # noqa: BLE001  (an extra: never the reason the line is missing)
            result["extra"]["sbm_shuffled_ids"] = {"error": str(e)[:160]}
        # ... and BASELINE config 3's shape with the same kind of structure (products-shaped SBM, INT32, shuffled ids): too sparse for the LDS-staged
        # kernel as a whole, so the part is split by density (csrc/lds_hybrid_dev.hpp) -- the cells inside communities through the LDS-staged kernel
        # over X staged in the propagation's order, the rest added by the sweep; lds_hybrid = 0 beside it is the part's own plan
        if args.shape == "reddit":
            hy_prev = None
            try:
                n3, nnz3, _ = synth.SHAPES["ogbn-products"]
                rp3, col3 = synth.make_shape("ogbn-products", seed=0, device=dev, kind="sbm")
                x3 = synth.features(n3, h, torch.int32, seed=0, device=dev)
                out3 = torch.empty((n3, h), dtype=torch.int32, device=dev)
                ref3, info3, hy_prev = None, {}, None
                for hy in (0, 1):
                    prev = _lib.set_tunable("lds_hybrid", hy)
                    hy_prev = prev if hy_prev is None else hy_prev
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    hd3 = _lib.group_create(_lib.CSR, _lib.INT32, [rp3.data_ptr()], [col3.data_ptr()], None, [n3], [n3], [nnz3], [1], [h], h)
                    torch.cuda.synchronize()
                    t_c3 = time.perf_counter() - t0
                    for _ in range(2):
                        _lib.spmm_run_group(hd3, [x3.data_ptr()], out3.data_ptr(), stream)
                    torch.cuda.synchronize()
                    evs = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
                    for i in range(6):
                        evs[i].record(main_stream)
                        _lib.spmm_run_group(hd3, [x3.data_ptr()], out3.data_ptr(), stream)
                    evs[6].record(main_stream)
                    torch.cuda.synchronize()
                    t3 = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(6))
                    key = "density_split" if hy else "own_plan"
                    info3[key] = {"ms_per_step": round(t3[len(t3) // 2], 4), "group_create_ms": round(t_c3 * 1e3, 1), "lds_runs": _lib.group_lds_runs(hd3),
                                  "note": _lib.group_lds_note(hd3)[-200:]}
                    if ref3 is None:
                        ref3 = out3.clone()
                        info3[key]["check"] = "column-count checksum exact" if torch.equal(
                            out3.double().sum(0), torch.bincount(col3.long(), minlength=n3).double() @ x3.double()) else "MISMATCH"
                    else:
                        info3[key]["check"] = "equal to the part's own plan, element by element" if torch.equal(out3, ref3) else "MISMATCH"
                    _lib.group_free(hd3)
                _lib.set_tunable("lds_hybrid", hy_prev)
                info3["note"] = ("products-shaped stochastic block model (N = 2 449 029, nnz = 123 718 280, 1 200 communities, 80 % of a row's entries inside its own), ids "
                                 f"shuffled, INT32 h = {h}; 6 steps, median, outside the timed region")
                result["extra"]["products_sbm_shuffled_ids"] = info3
                del rp3, col3, x3, out3, ref3
            except Exception as e:  # noqa: BLE001
                if hy_prev is not None:
                    _lib.set_tunable("lds_hybrid", hy_prev)
                result["extra"]["products_sbm_shuffled_ids"] = {"error": str(e)[:160]}
    if rank == 0 and world == 1 and not args.no_extra and not args.clustered and (args.shape, h) == ("reddit", 256):
        # BASELINE.json's OTHER configurations beside the headline, each with its own roofline object and check, measured
        # outside the timed region (pygim_amd/bench_configs.py); the SBM / clustered extras above are graphs of this build's own making
        from pygim_amd import bench_configs

        result["extra"]["note_on_structured_graphs"] = "clustered_*, sbm_shuffled_ids and products_sbm_shuffled_ids are NOT BASELINE configurations (locality studies of this build)"
        cfgs = {}
        for key, fn in (("configs[2]_products_coo_i32", lambda: bench_configs.config3_products_coo(dev, stream, h)),
                        ("configs[3]_reddit_gcn_3_layers_one_gpu", lambda: bench_configs.config4_gcn_one_gpu(dev, h)),
                        ("configs[4]_papers100m_per_gpu", lambda: bench_configs.config5_papers_slices(dev, stream))):
            t0 = time.perf_counter()
            try:
                cfgs[key] = fn()
            except Exception as e:  # noqa: BLE001  (an extra: never the reason the line is missing)
                cfgs[key] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
            if isinstance(cfgs[key], dict):
                cfgs[key]["wall_s"] = round(time.perf_counter() - t0, 1)
            torch.cuda.empty_cache()
        result["baseline_configs"] = cfgs
        if e2e_host is not None:
            result["extra"]["end_to_end_cpu_tensors"] = e2e_host
            if "ms_per_mul" in e2e_host:
                result["extra"]["end_to_end_cpu_tensors_ms"] = e2e_host["ms_per_mul"]
            try:   # ... and once more HERE, after every other leg of this process (the same call, whatever the runtime's engines do by now)
                again = bench_configs.end_to_end_cpu_tensors(rowptr, col, n, h)
                result["extra"]["end_to_end_cpu_tensors"]["after_the_other_legs"] = {k: again[k] for k in ("ms_per_mul", "direct_stores", "ms_each_call", "ms_per_mul_serial", "check") if k in again}
            except Exception as e:  # noqa: BLE001
                result["extra"]["end_to_end_cpu_tensors"]["after_the_other_legs"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, cpu_out = cpu_baseline(rowptr, col, x, args)
        result["cpu_baseline"] = base
        if not args.no_check:
            got = plan.full_c()[: cpu_out.shape[0]].cpu().numpy()
            result["check"] = ((f"bit-exact vs oracle on all {cpu_out.shape[0]} rows" if cpu_out.shape[0] == n else
                                f"bit-exact vs oracle on rows [0, {cpu_out.shape[0]})") if np.array_equal(got, cpu_out) else "MISMATCH")
            # real-valued features, whole graph (exercises the floating-point bar of north_star: 1e-5 relative): the same
            # product on X ~ U(-1, 1) against the oracle's row-parallel loop; error relative to |result| and to |A|.|x|
            import oracle

            xr = synth.features(n, h, torch.float32, seed=1, device=dev, kind="uniform")
            cr = torch.empty((my_rows, h), dtype=torch.float32, device=dev)
            ca = torch.empty_like(cr)
            _lib.spmm_run_group(handles[0], [xr.data_ptr()], cr.data_ptr(), stream)
            xa = xr.abs()
            _lib.spmm_run_group(handles[0], [xa.data_ptr()], ca.data_ptr(), stream)
            torch.cuda.synchronize()
            if K == 1 and my_rows == n:
                nr = cpu_out.shape[0]
                ref = np.zeros((nr, h), dtype=np.float32)
                oracle.spmm_csr_rowpar(rowptr[: nr + 1].cpu().numpy().astype(np.uint32), col[: int(rowptr[nr])].cpu().numpy().astype(np.uint32),
                                       None, xr.cpu().numpy(), nthreads=max(1, min(oracle.max_threads(), cpu_quota())), out=ref)
                got_r = cr[:nr].cpu().numpy().astype(np.float64)
                err = np.abs(got_r - ref.astype(np.float64))
                mag = np.abs(ref.astype(np.float64))
                bound = ca[:nr].cpu().numpy().astype(np.float64)
                nz = mag > 0
                result["check_uniform_f32"] = {
                    "rows": int(nr), "bit_identical_elements": float(np.mean(got_r == ref)),
                    "max_err_rel_result": float(np.max(err[nz] / mag[nz])) if nz.any() else 0.0,
                    "max_err_rel_absAx": float(np.max(err / np.maximum(bound, 1e-30))),
                    "elements_over_1e-5_of_result": int(np.sum(err[nz] > 1e-5 * mag[nz])),
                    "bar": "1e-5 relative (BASELINE.json north_star); |A||x| is the backward-stable scale of a 500-term sum"}
                if result["check_uniform_f32"]["max_err_rel_absAx"] > 1e-5:
                    result["check"] = "MISMATCH"
    if multi:
        # every rank now holds every block: column-count checksum of the assembled C (exact: small integers)
        full = plan.full_c()
        colcount = torch.bincount(col.long(), minlength=n).double()
        ok = full.shape[0] == n and torch.equal(full.double().sum(0), colcount @ x.double())
        flag = torch.tensor([0 if ok else 1], device=dev)
        dist.all_reduce(flag)
        contract = ("every row summed by one wave in stored order: FLT32 bit-identical to the CPU loop for any features"
                    if not split_tiles else
                    "column-split row tiles (lds_col_split_f32): a row's FLT32 sum is the sum of its column ranges' sequential sums -- the norm-wise "
                    "contract (<= 1e-5 |A||x|, INTEGRATION.md section 4), not the bit-identical one; with the driver's small-integer features every "
                    "partial sum is exact, so this check is exact all the same")
        result["check"] = (f"column-count checksum of the gathered C exact on every rank; contract verified: {contract}"
                           if int(flag.item()) == 0 else "MISMATCH")
    for hd in handles:
        _lib.group_free(hd)
    if rank == 0:
        print(f"[DATA]pim_time_spmm(ms):  {ms_per_step}", file=sys.stderr)
        print(f"[DATA]kernel_time(ms):  {k_ms}", file=sys.stderr)
        print(json.dumps(result), flush=True)
    if multi:
        dist.destroy_process_group()
    if result.get("check") == "MISMATCH":
        raise SystemExit("bench result differs from the oracle")


if __name__ == "__main__":
    main()
