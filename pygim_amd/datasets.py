"""On-disk graph loaders for the drivers' ``--datadir``.

The reference's ``load_datasets`` (spmm_test.py:40-71, inference.py:44-93) goes through torch_geometric / ogb dataset
classes (which download, then cache ``processed/*.pt`` pickles of PyG objects) and ``T.ToSparseTensor``.  Neither
package is installable here, so this module reads the RAW files those classes keep beside the processed ones -- the
formats are plain numpy / scipy / csv / pickle -- and rebuilds the same ``adj_t``:

    Planetoid (PubMed, Cora, CiteSeer)   <datadir>/<Name>/<Name>/raw/ind.<name>.graph   pickled {node: [neighbours]}
                                         -> edges both ways as listed, self loops removed, coalesced
                                         (torch_geometric/io/planetoid.py edge_index_from_dict)
    Reddit                               <datadir>/Reddit/raw/reddit_graph.npz         scipy sparse matrix -> coalesced
    OGB node property (ogbn-arxiv, ...)  <datadir>/<name>/<name_>/raw/edge.csv.gz + num-node-list.csv.gz
                                         (+ the inverse edges where the dataset's meta says add_inverse_edge)
    generic                              <datadir>/<name>/adj.npz                       scipy.sparse.save_npz of the adjacency

``adj_t`` is the TRANSPOSED adjacency in CSR (row = target, col = source, columns sorted), which is what
``T.ToSparseTensor`` stores.  Returns None when the files are absent; the drivers then fall back to the seeded
synthetic graph of the dataset's shape (no network on the GPU box).
"""
from __future__ import annotations

import gzip
import os
import pickle

import numpy as np

# ogb master.csv: datasets whose raw edge list holds each undirected edge once
OGB_ADD_INVERSE = {"ogbn-proteins": True, "ogbn-products": True, "ogbn-arxiv": False, "ogbn-papers100M": False,
                   "ogbn-mag": False}


def _csr_t(src, dst, n, coalesce=True):
    """adj_t = SparseTensor(row=dst, col=src): CSR over targets, sources sorted inside a row"""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    key = dst * n + src
    key = np.unique(key) if coalesce else np.sort(key, kind="stable")
    row, col = key // n, key % n
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(row, minlength=n), out=rowptr[1:])
    return rowptr, col, n


def _planetoid(root, name):
    path = os.path.join(root, name, "raw", f"ind.{name.lower()}.graph")
    if not os.path.isfile(path):
        return None
    with open(path, "rb") as f:
        graph = pickle.load(f, encoding="latin1")
    rows, cols = [], []
    for key, value in graph.items():
        rows += [int(key)] * len(value)
        cols += [int(v) for v in value]
    rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
    n = int(max(rows.max(initial=-1), cols.max(initial=-1))) + 1
    keep = rows != cols  # remove_self_loops
    return _csr_t(rows[keep], cols[keep], n)


def _reddit(root):
    path = os.path.join(root, "raw", "reddit_graph.npz")
    if not os.path.isfile(path):
        return None
    import scipy.sparse as sp

    a = sp.load_npz(path).tocoo()
    return _csr_t(a.row, a.col, int(a.shape[0]))


def _read_csv_gz(path):
    with gzip.open(path, "rt") as f:
        return np.loadtxt(f, delimiter=",", dtype=np.int64, ndmin=2)


def _ogb(root, name):
    raw = os.path.join(root, name.replace("-", "_"), "raw")
    edge, nodes = os.path.join(raw, "edge.csv.gz"), os.path.join(raw, "num-node-list.csv.gz")
    if not (os.path.isfile(edge) and os.path.isfile(nodes)):
        return None
    e = _read_csv_gz(edge)
    n = int(_read_csv_gz(nodes).sum())
    src, dst = e[:, 0], e[:, 1]
    if OGB_ADD_INVERSE.get(name, False):
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    # PygNodePropPredDataset keeps the edge list as read (no coalesce); ToSparseTensor sorts it
    return _csr_t(src, dst, n, coalesce=False)


def _generic(root):
    path = os.path.join(root, "adj.npz")
    if not os.path.isfile(path):
        return None
    import scipy.sparse as sp

    a = sp.load_npz(path).tocoo()
    return _csr_t(a.row, a.col, int(a.shape[0]), coalesce=False)


def load_adjacency(datadir, name):
    """(rowptr int64 [n+1], col int64 [nnz], n) of ``adj_t`` for dataset ``name`` under ``datadir`` (the directory layout
    the reference's ``osp.join(args.datadir, args.dataset)`` roots produce), or None when nothing usable is there."""
    root = os.path.join(datadir, name)
    if not os.path.isdir(root):
        return None
    if name in ("PubMed", "Cora", "CiteSeer"):
        got = _planetoid(root, name)
    elif name == "Reddit":
        got = _reddit(root)
    elif name.startswith("ogbn-"):
        got = _ogb(root, name)
    else:
        got = None
    return got if got is not None else _generic(root)


__all__ = ["load_adjacency", "OGB_ADD_INVERSE"]
