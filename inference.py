#!/usr/bin/env python3
"""Counterpart of the reference's inference.py: end-to-end GNN inference (GCN / GIN / SAGE, random
weights, eval mode) with the aggregation on the MI355X backend, timed as ``[DATA]infer_time(ms)``.

Same flags as the reference (inference.py:96-124) plus ``--device`` and multi-GPU through ``--gpus N`` (the plain command starts its
ranks itself) or ``python -m torch.distributed.run --nproc-per-node N inference.py --version spmm ...`` (row split of A,
RCCL all-gather between layers: BASELINE config 4).  Datasets are seeded synthetic graphs with the
named dataset's node / edge counts (no network); labels are random, so the accuracy print only keeps
the reference's log shape (its model is untrained too, inference.py:154-163).
"""
import argparse
import datetime
import os

import torch

from pygim_amd import gnn, pim_ops
from pygim_amd.backend_pim.grande import prepare_pim_spmm_grande
from pygim_amd.backend_pim.spmm import prepare_pim_spmm
from pygim_amd.backend_pim.spmv import prepare_pim_spmv
from pygim_amd.sparse_tensor import SparseTensor
from spmm_test import DATASETS, TORCH_TYPES, load_adjacency, load_ops


@torch.no_grad()
def test(args, model, data):
    """one timed forward pass; under torch.distributed every rank holds its row block of the nodes (features,
    labels, logits): the time is the slowest rank's, checksum and accuracy are summed over the ranks"""
    import torch.distributed as dist

    multi = dist.is_available() and dist.is_initialized()
    model.eval()
    graph = None
    if getattr(args, "graph", 0) and data["x"].is_cuda and not multi:
        graph = data.get("_graph")
        if graph is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):  # scratch buffers and plans are sized on first use
                    model(data["x"], data["adj_t"], data["edge_attr"])
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):  # the warm-up stream: the library keys its scratch buffers by stream
                out = model(data["x"], data["adj_t"], data["edge_attr"])
            graph = data["_graph"] = (g, out)
    if data["x"].is_cuda:
        torch.cuda.synchronize()
    if multi:
        dist.barrier()
    st = datetime.datetime.now()
    if graph is not None:
        graph[0].replay()
        y_pred = graph[1]
    else:
        y_pred = model(data["x"], data["adj_t"], data["edge_attr"])
    if y_pred.is_cuda:
        torch.cuda.synchronize()
    ms = (datetime.datetime.now() - st).total_seconds() * 1000
    stats = torch.stack([y_pred.double().abs().sum(), y_pred.argmax(dim=-1).eq(data["y"]).sum().double(),
                         torch.tensor(float(y_pred.size(0)), dtype=torch.float64, device=y_pred.device)])
    if multi:
        t = torch.tensor([ms], dtype=torch.float64, device=y_pred.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        ms = float(t.item())
    if not multi or dist.get_rank() == 0:
        print("[DATA]infer_time(ms): ", ms, flush=True)
        print("[DATA]logits_checksum: ", float(stats[0]), flush=True)
    return float(stats[1] / stats[2])


def get_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", type=str, default="PubMed", choices=sorted(DATASETS))
    ap.add_argument("--datadir", type=str, default="./data")
    ap.add_argument("--version", type=str, default="grande", choices=["spmm", "grande", "spmv", "cpu"])
    ap.add_argument("--lib_path", type=str, default="./backend_pim/spmm_grande/build/libbackend_pim.so")
    ap.add_argument("--model", type=str, default="gcn", choices=["gcn", "gin", "sage"])
    ap.add_argument("--num_layers", type=int, default=3)
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--hidden_size", type=int, default=256)
    ap.add_argument("--in_features", type=int, default=128)
    ap.add_argument("--num_classes", type=int, default=41)
    ap.add_argument("--data_type", type=str, default="INT32", choices=sorted(TORCH_TYPES))
    ap.add_argument("--sp_format", type=str, default="CSR", choices=["CSR", "COO"])
    ap.add_argument("--sp_parts", type=int, default=1)
    ap.add_argument("--ds_parts", type=int, default=1)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--nr_dpus", type=int, default=0)
    ap.add_argument("--group_per_rank", type=int, default=1)  # what the harness passes to the multigroup backend (experiment.py:434); no meaning here
    ap.add_argument("--device", type=str, default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--gpus", type=int, default=1, help="N > 1 as a plain command: the N ranks are started as child processes of "
                    "torch.distributed.run (pygim_amd/launch.py), one per GPU, row split of A with RCCL all-gathers between the layers")
    ap.add_argument("--graph", type=int, default=0, help="1 = capture the forward pass into a HIP graph after two warm-up runs and "
                    "replay it (one launch per inference: pays on small graphs, where a forward pass is ~40 short kernels)")
    ap.add_argument("--fuse_post", type=int, default=0, help="1 = GCN: bias + BatchNorm(eval) + ReLU folded into the "
                    "aggregation's last store (same mathematics, one rounding sequence; not in the reference)")
    args = ap.parse_args()
    print(args, flush=True)
    args.data_type = TORCH_TYPES[args.data_type]
    return args


def main(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # PYGIM_FORCE_COLLECTIVES=1 under torch.distributed.run with ONE rank: the row-sharded multi-GPU path with every
    # collective executing at world size 1 (first-run check of the RCCL path on one GPU)
    sharded = world > 1 or (os.environ.get("PYGIM_FORCE_COLLECTIVES", "0") == "1" and "MASTER_ADDR" in os.environ
                            and args.version != "cpu")
    if sharded:
        import torch.distributed as dist

        backend = os.environ.get("PYGIM_BENCH_BACKEND", "nccl")  # gloo = logic check with several ranks on one GPU
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        idx = lr if backend == "nccl" else lr % torch.cuda.device_count()
        torch.cuda.set_device(idx)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", idx))
        else:
            dist.init_process_group(backend)
        args.device = "cuda"
        if rank == 0:
            from pygim_amd import autotune

            n0, nnz0, _ = DATASETS[args.dataset]
            best, _tab = autotune.choose(n0, n0, nnz0, args.hidden_size, 4, world)
            print(f"[DATA]rccl_world: {dist.get_world_size()}", flush=True)
            print(f"[DATA]partition_prior(row x feature): {best.row_parts} x {best.feat_parts}  (used: row-sharded "
                  f"{world} x 1, the arrangement the dense layers need)", flush=True)
    rowptr, col, n = load_adjacency(args)  # --datadir when the raw files are there, else the seeded synthetic graph
    torch.manual_seed(0)
    x = torch.randn(n, args.in_features)
    y = torch.randint(0, args.num_classes, (n,))
    data = {"x": x.to(args.device), "y": y.to(args.device), "edge_attr": None}
    adj_t = SparseTensor(rowptr=rowptr, col=col, sparse_sizes=(n, n))
    if args.version == "cpu":
        data["adj_t"] = adj_t
        data["x"], data["y"] = x, y
    elif sharded:
        # sp_parts = world as a row split with row-SHARDED activations (pygim_amd/dist.py RowShardAdj)
        from pygim_amd.dist import RowShardAdj

        shard = RowShardAdj(rowptr, col, n, args.data_type, args.hidden_size)
        data["adj_t"] = shard
        data["x"], data["y"] = shard.local_rows(data["x"]).contiguous(), shard.local_rows(data["y"]).contiguous()
    else:
        load_ops(args)
        if args.version == "grande":
            units = torch.ops.pim_ops.dpu_init_ranks(args.sp_parts)
            data["adj_t"] = prepare_pim_spmm_grande(adj_t, args, units)
        else:
            torch.ops.pim_ops.dpu_init_ranks(args.sp_parts * args.ds_parts)
            data["adj_t"] = (prepare_pim_spmm if args.version == "spmm" else prepare_pim_spmv)(adj_t, args)
    Model = {"gcn": gnn.GCN, "gin": gnn.GIN, "sage": gnn.SAGE}[args.model]
    torch.manual_seed(1)
    model = Model(args.in_features, args.hidden_size, args.num_classes, args.num_layers).to(data["x"].device)
    model.fuse_post = bool(args.fuse_post)
    for i in range(args.repeat):
        if rank == 0:
            print("-------------------- Model={} nrl={} Repeat={}--------------------".format(args.model, args.num_layers, i), flush=True)
        acc = test(args, model, data)
        if rank == 0:
            print(f"Test_acc: {acc:.4f}")
    if args.version != "cpu" and not sharded:
        torch.ops.pim_ops.dpu_release()
    if sharded:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    os.environ.setdefault("PYGIM_DATA_LOG", "1")  # the reference's per-run [DATA] timer lines
    _args = get_args()
    if _args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import sys

        from pygim_amd.launch import self_launch

        raise SystemExit(self_launch(os.path.abspath(__file__), _args.gpus, sys.argv[1:], threads_per_rank=max(1, (os.cpu_count() or 8) // _args.gpus), tag="inference"))
    main(_args)
