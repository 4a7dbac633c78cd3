"""pygim_amd/datasets.py: adj_t from the RAW files torch_geometric / ogb keep under --datadir (the reference's
load_datasets, spmm_test.py:40-71), on tiny fabricated directories of each format; and the driver picking them up."""
import collections
import gzip
import os
import pickle
import subprocess
import sys

import numpy as np
import scipy.sparse as sp

from pygim_amd import datasets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _csr_with_dups(src, dst, n):
    order = np.lexsort((src, dst))
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(dst, minlength=n), out=rowptr[1:])
    return rowptr, np.asarray(src)[order]


def test_planetoid_raw(tmp_path):
    rng = np.random.default_rng(3)
    n = 40
    graph = collections.defaultdict(list)
    for v in range(n):
        for u in rng.integers(0, n, size=int(rng.integers(0, 6))):
            graph[v].append(int(u))  # includes self loops and duplicates, as the real files do
    raw = tmp_path / "PubMed" / "PubMed" / "raw"
    raw.mkdir(parents=True)
    with open(raw / "ind.pubmed.graph", "wb") as f:
        pickle.dump(graph, f, protocol=2)
    rowptr, col, nn = datasets.load_adjacency(str(tmp_path), "PubMed")
    src = np.array([k for k, v in graph.items() for _ in v])
    dst = np.array([u for v in graph.values() for u in v])
    keep = src != dst
    # edge (row=key, col=neighbour) -> adj_t row = neighbour, col = key
    a = sp.coo_matrix((np.ones(keep.sum()), (dst[keep], src[keep])), shape=(nn, nn)).tocsr()
    a.sort_indices()
    assert nn == int(max(src.max(), dst.max())) + 1
    assert np.array_equal(rowptr, a.indptr) and np.array_equal(col, a.indices)


def test_reddit_raw(tmp_path):
    rng = np.random.default_rng(4)
    n = 30
    a = sp.random(n, n, density=0.2, random_state=5, format="coo")
    raw = tmp_path / "Reddit" / "raw"
    raw.mkdir(parents=True)
    sp.save_npz(str(raw / "reddit_graph.npz"), a)
    rowptr, col, nn = datasets.load_adjacency(str(tmp_path), "Reddit")
    t = sp.coo_matrix((np.ones(a.nnz), (a.col, a.row)), shape=(n, n)).tocsr()
    t.sort_indices()
    assert nn == n and np.array_equal(rowptr, t.indptr) and np.array_equal(col, t.indices)
    assert rng is not None


def test_ogb_raw_with_and_without_inverse_edges(tmp_path):
    rng = np.random.default_rng(6)
    n, m = 25, 80
    e = rng.integers(0, n, size=(m, 2))
    for name, inverse in (("ogbn-arxiv", False), ("ogbn-proteins", True)):
        raw = tmp_path / name / name.replace("-", "_") / "raw"
        raw.mkdir(parents=True)
        with gzip.open(raw / "edge.csv.gz", "wt") as f:
            for a, b in e:
                f.write(f"{a},{b}\n")
        with gzip.open(raw / "num-node-list.csv.gz", "wt") as f:
            f.write(f"{n}\n")
        rowptr, col, nn = datasets.load_adjacency(str(tmp_path), name)
        src, dst = e[:, 0], e[:, 1]
        if inverse:
            src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
        rp, cl = _csr_with_dups(src, dst, n)
        assert nn == n and len(col) == len(src)  # duplicates are kept (ToSparseTensor only sorts)
        assert np.array_equal(rowptr, rp) and np.array_equal(col, cl)


def test_absent_dataset_gives_none(tmp_path):
    assert datasets.load_adjacency(str(tmp_path), "PubMed") is None
    (tmp_path / "PubMed").mkdir()
    assert datasets.load_adjacency(str(tmp_path), "PubMed") is None


def test_driver_reads_datadir(tmp_path):
    """spmm_test.py --version cpu with --datadir pointing at a fabricated Cora: the graph on disk is used, not the synthetic one"""
    graph = {v: [(v + 1) % 12, (v + 5) % 12] for v in range(12)}
    raw = tmp_path / "Cora" / "Cora" / "raw"
    raw.mkdir(parents=True)
    with open(raw / "ind.cora.graph", "wb") as f:
        pickle.dump(graph, f, protocol=2)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "spmm_test.py"), "--version", "cpu", "--dataset", "Cora",
                        f"--datadir={tmp_path}", "--hidden_size", "8", "--data_type", "INT32", "--repeat", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert "Cora Dataset Info: Node(12), Edge(24)" in r.stdout and "[INFO] Cora: adj_t read from" in r.stdout
