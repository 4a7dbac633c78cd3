#!/usr/bin/env python3
"""Counterpart of the reference's spmm_test.py for the MI355X backend: same flags, same call
pattern (load the op library -> dpu_init_* -> prepare_pim_* -> repeat { cpu matmul ; pim mul } ->
dpu_release), same ``[DATA]key: value`` log grammar (parsed by the reference's utils/experiment.py).

Differences forced by the environment: datasets cannot be downloaded, so ``--dataset`` selects a
seeded synthetic graph with that dataset's node / edge counts (pygim_amd/synth.py); ``--device``
(new, default cpu like the reference) lets the features live on the GPU; results of the two paths
are compared (the reference has the comparison commented out, spmm_test.py:36-37).
"""
import argparse
import datetime
import os
import sys

import torch

from pygim_amd import pim_ops, synth
from pygim_amd.backend_pim.grande import prepare_pim_spmm_grande
from pygim_amd.backend_pim.spmm import prepare_pim_spmm
from pygim_amd.backend_pim.spmv import prepare_pim_spmv
from pygim_amd.sparse_tensor import SparseTensor, matmul

TORCH_TYPES = {"INT64": torch.int64, "INT32": torch.int32, "INT16": torch.int16, "INT8": torch.int8,
               "FLT32": torch.float32, "DBL64": torch.float64}
DATASETS = synth.DATASETS  # (nodes, edges, max degree) of the datasets the reference driver knows (spmm_test.py:42-53)


def ms_since(t0):
    return (datetime.datetime.now() - t0).total_seconds() * 1000


def load_adjacency(args):
    """adj_t as (rowptr, col, n) on the CPU: read from --datadir when the dataset's raw files are there (the layout the
    reference's load_datasets roots produce, spmm_test.py:40-50; pygim_amd/datasets.py), else the seeded synthetic graph
    with the dataset's node / edge counts"""
    from pygim_amd import datasets

    got = datasets.load_adjacency(args.datadir, args.dataset)
    if got is not None:
        rowptr, col, n = got
        print(f"[INFO] {args.dataset}: adj_t read from {os.path.join(args.datadir, args.dataset)} ({n} nodes, {len(col)} edges)",
              flush=True)
        return torch.from_numpy(rowptr), torch.from_numpy(col), n
    n, nnz, dmax = DATASETS[args.dataset]
    gen_dev = "cuda" if torch.cuda.is_available() else "cpu"
    rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=gen_dev)
    return rowptr.cpu().long(), col.cpu().long(), n


def load_graph(args):
    rowptr, col, n = load_adjacency(args)
    adj_t = SparseTensor(rowptr=rowptr, col=col, sparse_sizes=(n, n))
    x = torch.randint(-2 ^ 6, 2 ^ 6, (n, args.hidden_size), dtype=args.data_type)  # the reference's generator, verbatim
    return adj_t, x


def run_once(adj_t, pim_adj_t, x, args):
    print("{} Dataset Info: Node({}), Edge({})".format(args.dataset, adj_t.size(1), adj_t.nnz()))
    t0 = datetime.datetime.now()
    res_torch = matmul(adj_t, x)
    print("[DATA]torch_time(ms): ", ms_since(t0), flush=True)
    if args.version == "cpu":
        return
    x_pim = x.type(args.data_type).to(args.device)
    if x_pim.is_cuda:
        torch.cuda.synchronize()
    t0 = datetime.datetime.now()
    res_lib = pim_adj_t.mul(x_pim)
    if res_lib.is_cuda:
        torch.cuda.synchronize()
    print("[DATA]pim_time_spmm(ms): ", ms_since(t0), flush=True)
    same = torch.equal(res_lib.cpu(), res_torch) if not x.is_floating_point() else \
        torch.allclose(res_lib.cpu().double(), res_torch.double(), rtol=1e-5, atol=1e-5)
    print("[DATA]outputs_equal: ", int(same), flush=True)
    if not same:
        sys.exit("[ERROR] Outputs differ!")


def get_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", type=str, default="PubMed", choices=sorted(DATASETS))
    ap.add_argument("--datadir", type=str, default="./data")
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--version", type=str, default="spmm", choices=["spmm", "grande", "spmv", "cpu"])
    ap.add_argument("--tune", type=bool, default=True)
    ap.add_argument("--lib_path", type=str, default="./backend_pim/spmm_default/build/libbackend_pim.so")
    ap.add_argument("--hidden_size", type=int, default=256)
    ap.add_argument("--data_type", type=str, default="INT32", choices=sorted(TORCH_TYPES))
    ap.add_argument("--sp_format", type=str, default="COO", choices=["CSR", "COO"])
    ap.add_argument("--sp_parts", type=int, default=32)
    ap.add_argument("--ds_parts", type=int, default=1)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--nr_dpus", type=int, default=0)
    ap.add_argument("--group_per_rank", type=int, default=1)  # what the harness passes to the multigroup backend (experiment.py:434); no meaning here
    ap.add_argument("--device", type=str, default="cpu", choices=["cpu", "cuda"])
    args = ap.parse_args()
    print(args, flush=True)
    args.data_type = TORCH_TYPES[args.data_type]
    return args


def load_ops(args):
    """``torch.ops.load_library(args.lib_path)`` (reference spmm_test.py:111) when --lib_path names a built
    library of the variant --version asks for; otherwise the same ops are registered from Python over the
    same C ABI (the flag defaults name the spmm_default library whatever --version says)."""
    path = args.lib_path.strip('"')
    if os.path.isfile(path) and pim_ops.variant_of(path) == args.version:
        pim_ops.load_library(path)
    else:
        pim_ops.load(args.version)


def main(args):
    adj_t, x = load_graph(args)
    pim_adj_t = None
    if args.version != "cpu":
        load_ops(args)
        dpus_per_rank = None
        if args.nr_dpus == 0:
            if args.version == "grande":
                dpus_per_rank = torch.ops.pim_ops.dpu_init_ranks(args.sp_parts)
            else:
                torch.ops.pim_ops.dpu_init_ranks(args.sp_parts * args.ds_parts)
        else:
            dpus_per_rank = torch.ops.pim_ops.dpu_init_dpus(args.nr_dpus)
        if args.version == "spmm":
            pim_adj_t = prepare_pim_spmm(adj_t, args)
        elif args.version == "spmv":
            pim_adj_t = prepare_pim_spmv(adj_t, args)
        else:
            pim_adj_t = prepare_pim_spmm_grande(adj_t, args, dpus_per_rank)
    for i in range(args.repeat):
        print("-------------------- Model=spmm_test Repeat={}--------------------".format(i), flush=True)
        run_once(adj_t, pim_adj_t, x, args)
    if args.version != "cpu":
        torch.ops.pim_ops.dpu_release()


if __name__ == "__main__":
    os.environ.setdefault("PYGIM_DATA_LOG", "1")  # the reference's per-run [DATA] timer lines
    main(get_args())
