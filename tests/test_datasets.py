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


# ---------------------------------------------------------------------------
# processed/*.pt of the PyG / OGB dataset classes, read without those packages
# ---------------------------------------------------------------------------
def _fake_pyg_modules():
    """minimal stand-ins for the classes whose pickles the processed files hold -- only to WRITE the test files; they are
    removed from sys.modules again before the loader runs"""
    import types

    mods = {}
    for name in ("torch_geometric", "torch_geometric.data", "torch_geometric.data.data", "torch_geometric.data.storage"):
        mods[name] = types.ModuleType(name)

    class Data:          # PyG 1.x: attributes in __dict__; PyG 2.0-2.3: __dict__ = {'_store': GlobalStorage}
        pass

    class GlobalStorage:
        pass

    Data.__module__ = "torch_geometric.data.data"
    Data.__qualname__ = "Data"
    GlobalStorage.__module__ = "torch_geometric.data.storage"
    GlobalStorage.__qualname__ = "GlobalStorage"
    mods["torch_geometric.data.data"].Data = Data
    mods["torch_geometric.data.storage"].GlobalStorage = GlobalStorage
    return mods, Data, GlobalStorage


def _write_processed(path, style, edge_index, n, with_num_nodes):
    import torch

    mods, Data, GlobalStorage = _fake_pyg_modules()
    x = torch.zeros(n, 3)
    fields = {"x": x, "edge_index": edge_index, "y": torch.zeros(n, dtype=torch.long)}
    if with_num_nodes:
        fields["num_nodes"] = n
        del fields["x"]
    slices = {k: torch.tensor([0, v.shape[-1] if k == "edge_index" else v.shape[0]]) for k, v in fields.items() if torch.is_tensor(v)}
    sys.modules.update(mods)
    try:
        if style == "v1":
            d = Data()
            d.__dict__.update(fields)
            torch.save((d, slices), path)
        elif style == "v2":
            d = Data()
            st = GlobalStorage()
            st.__dict__["_mapping"] = fields
            st.__dict__["_parent"] = d          # (a cycle, as the real pickles have)
            d.__dict__["_store"] = st
            torch.save((d, slices), path)
        else:                                   # PyG >= 2.4: (data.to_dict(), slices, data.__class__)
            torch.save((fields, slices, Data), path)
    finally:
        for k in mods:
            sys.modules.pop(k, None)


import pytest  # noqa: E402


@pytest.mark.parametrize("style", ["v1", "v2", "v3"])
@pytest.mark.parametrize("with_num_nodes", [False, True])
def test_processed_pt_without_torch_geometric(tmp_path, style, with_num_nodes):
    import torch

    rng = np.random.default_rng(11)
    n, e = 37, 300
    src, dst = rng.integers(0, n - 2, size=e), rng.integers(0, n - 2, size=e)   # the last two nodes are isolated
    ei = torch.from_numpy(np.stack([src, dst])).long()
    proc = tmp_path / "Reddit" / "processed"
    proc.mkdir(parents=True)
    _write_processed(str(proc / "data.pt"), style, ei, n, with_num_nodes)
    assert "torch_geometric" not in sys.modules
    rowptr, col, nn = datasets.load_adjacency(str(tmp_path), "Reddit")
    want_ptr, want_col = _csr_with_dups(src, dst, n)
    assert nn == n and np.array_equal(rowptr, want_ptr) and np.array_equal(col, want_col)


def test_processed_pt_ogb_layout_and_stored_adj_t(tmp_path):
    """OGB keeps <root>/<name_>/processed/geometric_data_processed.pt; a cache made with pre_transform=ToSparseTensor holds
    adj_t (torch_sparse.SparseTensor -> SparseStorage with _rowptr / _col) instead of edge_index"""
    import types
    import torch

    rng = np.random.default_rng(12)
    n, e = 25, 120
    src, dst = rng.integers(0, n, size=e), rng.integers(0, n, size=e)
    want_ptr, want_col = _csr_with_dups(src, dst, n)
    mod = types.ModuleType("torch_sparse")
    sub = types.ModuleType("torch_sparse.tensor")
    sub2 = types.ModuleType("torch_sparse.storage")

    class SparseTensor:
        pass

    class SparseStorage:
        pass

    SparseTensor.__module__, SparseTensor.__qualname__ = "torch_sparse.tensor", "SparseTensor"
    SparseStorage.__module__, SparseStorage.__qualname__ = "torch_sparse.storage", "SparseStorage"
    sub.SparseTensor, sub2.SparseStorage = SparseTensor, SparseStorage
    st = SparseStorage()
    st.__dict__.update({"_row": None, "_rowptr": torch.from_numpy(want_ptr), "_col": torch.from_numpy(want_col), "_value": None,
                        "_sparse_sizes": (n, n)})
    adj = SparseTensor()
    adj.__dict__["storage"] = st
    proc = tmp_path / "ogbn-arxiv" / "ogbn_arxiv" / "processed"
    proc.mkdir(parents=True)
    mods = {"torch_sparse": mod, "torch_sparse.tensor": sub, "torch_sparse.storage": sub2}
    sys.modules.update(mods)
    try:
        torch.save(({"adj_t": adj, "num_nodes": n}, None, None), str(proc / "geometric_data_processed.pt"))
    finally:
        for k in mods:
            sys.modules.pop(k, None)
    rowptr, col, nn = datasets.load_adjacency(str(tmp_path), "ogbn-arxiv")
    assert nn == n and np.array_equal(rowptr, want_ptr) and np.array_equal(col, want_col)


def test_processed_pt_refuses_foreign_code(tmp_path):
    """the reader is not a general unpickler: a file that names anything but tensors, containers and PyG objects is refused"""
    import torch

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))

    path = tmp_path / "data.pt"
    torch.save((Evil(), None), str(path))
    with pytest.raises(pickle.UnpicklingError):
        datasets.load_processed(str(path))


@pytest.mark.parametrize("module,name", [
    ("torch._utils", "_rebuild_tensor.__globals__.__class__.get"),   # protocol-4 dotted name: a getattr chain into dict.get
    ("torch._utils", "_rebuild_device_tensor_from_numpy.__globals__"),
    ("torch._utils", "_rebuild_nonexistent"),                          # (a prefix is not enough)
    ("torch", "EvilStorage"),                                          # (nor a suffix)
    ("torch.storage", "_load_from_bytes"),                             # (would re-enter an unrestricted torch.load)
    ("os", "system"), ("builtins", "eval"), ("builtins", "getattr"),
])
def test_restricted_unpickler_refuses_dotted_and_unlisted_names(module, name):
    """names are matched exactly, and a dotted name (pickle protocol 4 resolves it attribute by attribute) is
    refused before any lookup -- the harmless payload below used to come back as dict.get"""
    import io

    up = datasets._PygUnpickler(io.BytesIO(b""))
    with pytest.raises(pickle.UnpicklingError):
        up.find_class(module, name)
    # the same through a real protocol-4 stream: GLOBAL by STACK_GLOBAL with the dotted name
    payload = pickle.PROTO + bytes([4]) + pickle.SHORT_BINUNICODE + bytes([len(module)]) + module.encode() + \
        pickle.SHORT_BINUNICODE + bytes([len(name)]) + name.encode() + pickle.STACK_GLOBAL + pickle.STOP
    with pytest.raises(pickle.UnpicklingError):
        datasets._PygUnpickler(io.BytesIO(payload)).load()
