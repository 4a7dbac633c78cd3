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

    processed caches                     <root>/processed/data.pt | geometric_data_processed.pt: what the PyG / OGB dataset
                                         classes torch.save after their first run -- ``(Data, slices)`` pickles of
                                         torch_geometric objects (PyG 1.x / 2.0-2.3) or ``(dict, slices, class)``
                                         (PyG >= 2.4).  Read WITHOUT torch_geometric: a restricted unpickler rebuilds
                                         tensors and plain containers and turns every torch_geometric / torch_sparse /
                                         ogb class into an inert attribute bag (``load_processed``); anything else
                                         (``os.system`` ...) is refused.  Used when the raw files are gone.

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


# ---------------------------------------------------------------------------
# processed/*.pt of the PyG / OGB dataset classes, without those packages
# ---------------------------------------------------------------------------
_STUB_PREFIXES = ("torch_geometric", "torch_sparse", "ogb")
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"), ("builtins", "dict"), ("builtins", "list"),
    ("builtins", "tuple"), ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "int"), ("builtins", "float"),
    ("builtins", "bool"), ("builtins", "str"), ("builtins", "bytes"), ("builtins", "complex"), ("builtins", "slice"),
    ("builtins", "object"), ("_codecs", "encode"), ("copyreg", "_reconstructor"),
    ("numpy", "dtype"), ("numpy", "ndarray"), ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
}


class _Bag:
    """stands in for any torch_geometric / torch_sparse / ogb object: keeps the pickled state, runs none of its code"""

    def __init__(self, *args, **kwargs):
        self.__dict__["_args"] = args
        self.__dict__.update(kwargs)

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):  # (dict state, slots state)
            state = {**(state[0] or {}), **state[1]}
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_state"] = state


_bag_classes: dict = {}


def _bag_class(module, name):
    key = (module, name)
    if key not in _bag_classes:
        _bag_classes[key] = type(name, (_Bag,), {"__module__": "pygim_amd.datasets", "_origin": f"{module}.{name}"})
    return _bag_classes[key]


# exact names of torch that a tensor pickle may name -- rebuilders, storages, dtypes, Size; nothing is matched by prefix or suffix
_TORCH_GLOBALS = {
    ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._utils", "_rebuild_sparse_tensor"),
    ("torch._utils", "_rebuild_sparse_csr_tensor"), ("torch._utils", "_rebuild_qtensor"),
    ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch", "Size"), ("torch", "Tensor"), ("torch", "device"), ("torch", "dtype"),
    ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"),
    ("torch.serialization", "_get_layout"),
}
_TORCH_STORAGES = {"DoubleStorage", "FloatStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage", "ShortStorage",
                   "CharStorage", "ByteStorage", "BoolStorage", "UntypedStorage", "TypedStorage"}


class _PygUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        # protocol 4 resolves dotted names through getattr chains ("_rebuild_tensor.__globals__..." reaches builtins): a global is a
        # plain identifier in a plain module path, or it is refused before anything is looked up
        if not isinstance(module, str) or not isinstance(name, str) or "." in name or not name.isidentifier() or \
                not all(part.isidentifier() for part in module.split(".")):
            raise pickle.UnpicklingError(f"processed dataset file names {module!r}.{name!r}: not a plain global")
        if module.split(".")[0] in _STUB_PREFIXES:
            return _bag_class(module, name)
        if (module, name) in _TORCH_GLOBALS or (module, name) in _SAFE_GLOBALS:
            return super().find_class(module, name)
        if module == "torch" and (name in _TORCH_STORAGES or isinstance(getattr(__import__("torch"), name, None), __import__("torch").dtype)):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"processed dataset file refers to {module}.{name}: not a tensor, a container or a PyG object")


class _PickleShim:
    """what torch.load expects of ``pickle_module``"""
    __name__ = "pickle"
    Unpickler = _PygUnpickler

    @staticmethod
    def load(f, **kwargs):
        return _PygUnpickler(f, **kwargs).load()


def _walk(obj, found, depth=0):
    """collect named tensors / sizes from dicts, attribute bags and sequences (first hit of a name wins)"""
    import torch

    if depth > 6 or obj is None:
        return
    if isinstance(obj, _Bag):
        obj = obj.__dict__
    if isinstance(obj, dict):
        for k, v in obj.items():
            if isinstance(k, str) and (torch.is_tensor(v) or isinstance(v, (int, tuple, list)) and k in ("num_nodes", "_num_nodes", "_sparse_sizes")):
                found.setdefault(k.lstrip("_") if k in ("_num_nodes",) else k, v)
        for k, v in obj.items():
            if isinstance(v, (dict, _Bag)):
                _walk(v, found, depth + 1)
    elif isinstance(obj, (tuple, list)):
        for v in obj[:4]:
            if isinstance(v, (dict, _Bag)):
                _walk(v, found, depth + 1)


def load_processed(path):
    """(rowptr, col, n) of ``adj_t`` from one processed ``.pt`` of a single-graph PyG dataset (Planetoid, Reddit,
    PygNodePropPredDataset ...), as ``T.ToSparseTensor`` would build it from ``data.edge_index`` (row = target, col = source,
    sorted, duplicates kept) -- or from a stored ``adj_t`` when the cache was made with ``pre_transform=ToSparseTensor``."""
    import torch

    obj = torch.load(path, map_location="cpu", pickle_module=_PickleShim, weights_only=False)
    data = obj[0] if isinstance(obj, (tuple, list)) and obj else obj   # (data, slices[, class]): one graph -> slices are trivial
    found: dict = {}
    _walk(data, found)
    n = found.get("num_nodes")
    if isinstance(n, (tuple, list)):
        n = n[0] if n else None
    if torch.is_tensor(n):
        n = int(n.reshape(-1)[0]) if n.numel() else None
    if "edge_index" in found and found["edge_index"].dim() == 2 and found["edge_index"].shape[0] == 2:
        ei = found["edge_index"].numpy().astype(np.int64)
        if n is None:
            n = int(found["x"].shape[0]) if "x" in found and found["x"].dim() >= 1 else int(ei.max(initial=-1)) + 1
        return _csr_t(ei[0], ei[1], int(n), coalesce=False)
    if "_col" in found and ("_rowptr" in found or "_row" in found):  # torch_sparse.SparseStorage of adj_t
        col = found["_col"].numpy().astype(np.int64)
        sizes = found.get("_sparse_sizes")
        if "_rowptr" in found and found["_rowptr"] is not None:
            rowptr = found["_rowptr"].numpy().astype(np.int64)
            return rowptr, col, int(sizes[0]) if sizes else len(rowptr) - 1
        row = found["_row"].numpy().astype(np.int64)
        n = int(sizes[0]) if sizes else int(max(row.max(initial=-1), col.max(initial=-1))) + 1
        return _csr_t(col, row, n, coalesce=False)
    return None


def _processed(root, name):
    # Planetoid(root, name) keeps <root>/<name>/processed, OGB <root>/<name_with_underscores>/processed, Reddit <root>/processed
    inner = name.replace("-", "_") if name.startswith("ogbn-") else name
    for d in (os.path.join(root, inner), root):
        for fname in ("geometric_data_processed.pt", "data.pt"):
            path = os.path.join(d, "processed", fname)
            if os.path.isfile(path):
                got = load_processed(path)
                if got is not None:
                    return got
    return None


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
    if got is None:
        got = _processed(root, name)
    return got if got is not None else _generic(root)


__all__ = ["load_adjacency", "load_processed", "OGB_ADD_INVERSE"]
