"""Generates the golden vectors under tests/golden/ (run in the BUILD container).

Sources of truth, in order:
  * partition_*.npz  -- outputs of the REFERENCE's own support/partition.c compiled in place
                        into oracle/_ref (oracle/Makefile target `ref`); needs /root/reference.
  * spmm_*.npz       -- the reference holds no vectors for the arithmetic.
                        Expected outputs are those of the REFERENCE'S OWN host loops: spmm_host_coo
                        (spmm_default/spmm_mul_coo.c:40-51), the valued spmm_host_csr
                        (spmm_grande/spmm_mul_csr.c:119-136), the unit-weight spmm_host_csr
                        (spmm_default/spmm_mul_csr.c:100-113) and spmv's spmm_host
                        (spmv_sparseP/spmv_mul_coo.c:92-103), cut out of the reference files by name
                        and compiled in place with the reference's own support headers
                        (oracle/build_ref_host.sh -> oracle/_ref/libref_host_*).  A vector is accepted
                        only when the oracle restatement gives the SAME BITS (all six types, floats
                        included) and torch.sparse.mm / scipy agree (bit-exact for integers, 1e-5
                        for the real-valued float cases).  `pinned_by` records it.
  * group_*.npz      -- sp_parts x ds_parts group products from the reference's own group drivers
                        (spmm_host_csr_group / spmm_host_coo_group, spmm_default/ops.hpp:42-62,
                        97-118) on the reference's own structs (oracle/ref_host_glue.inc).
Inputs follow the reference driver: graph shapes of SURVEY.md section 8c, features
torch.randint(-8, 4) under torch.manual_seed (spmm_test.py:70).
Usage: python tests/golden/make_golden.py [quant]
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from conftest import NP_DTYPES, coalesce, random_csr  # noqa: E402
from pygim_amd import synth  # noqa: E402

TORCH_OF = {"INT8": torch.int8, "INT16": torch.int16, "INT32": torch.int32, "INT64": torch.int64,
            "FLT32": torch.float32, "DBL64": torch.float64}


def cross_check(fmt, rowptr, row, col, vals, x, y, exact):
    n = y.shape[0]
    wide = np.int64 if np.issubdtype(x.dtype, np.integer) else np.float64
    v = np.ones(len(col), dtype=wide) if vals is None else vals.astype(wide)
    if fmt == "CSR":
        a = sp.csr_matrix((v, col.copy(), rowptr.copy()), shape=(n, x.shape[0]))  # abs() below de-duplicates in place
    else:
        a = sp.coo_matrix((v, (row, col)), shape=(n, x.shape[0])).tocsr()
    ref = a @ x.astype(wide)
    # floating-point bar of BASELINE.json: 1e-5 relative -- measured against the magnitude
    # of the sum, |A| . |x| (a bound relative to the rounded result itself is meaningless
    # where terms cancel)
    scale = abs(a) @ np.abs(x.astype(wide))
    if exact:
        assert np.array_equal(y, ref.astype(x.dtype)), "oracle != scipy"
    else:
        assert np.all(np.abs(y - ref) <= 1e-5 * scale + 1e-30), "oracle vs scipy beyond 1e-5 relative"
    if fmt == "COO":
        t = torch.sparse_coo_tensor(torch.tensor(np.stack([row, col]).astype(np.int64)),
                                    torch.from_numpy(v.astype(x.dtype)), (n, x.shape[0]))
        yt = torch.sparse.mm(t, torch.from_numpy(x)).numpy()
        if exact:
            assert np.array_equal(y, yt), "oracle != torch.sparse.mm"
        else:
            assert np.all(np.abs(y - yt) <= 1e-5 * scale + 1e-30)


def torch_sparse_pin(fmt, rowptr, row, col, vals, x, y, exact):
    """The reference's real version=cpu call (spmm_test.py:25: ``torch_sparse.matmul(adj_t, x)``), when that package can be
    imported in the build container: returns "torch_sparse <version>" after checking the vector against it, else None
    (then the fixture records pinned_by = "unpinned: ..." and DESIGN.md section 2 keeps saying so)."""
    try:
        import torch_sparse  # noqa: F401  (un-vendored, version-unpinned dependency of the reference, Libs/install_libs.sh:13)
    except Exception:
        return None
    n = y.shape[0]
    if fmt == "CSR":
        r = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr.astype(np.int64)))
    else:
        r = row.astype(np.int64)
    v = None if vals is None else torch.from_numpy(vals)
    a = torch_sparse.SparseTensor(row=torch.from_numpy(r), col=torch.from_numpy(col.astype(np.int64)), value=v,
                                  sparse_sizes=(n, x.shape[0]))
    yt = torch_sparse.matmul(a, torch.from_numpy(x)).numpy()
    if exact:
        assert np.array_equal(y, yt.astype(y.dtype)), "oracle != torch_sparse.matmul"
    else:
        assert np.allclose(y, yt, rtol=1e-5, atol=0), "oracle vs torch_sparse.matmul beyond 1e-5 relative"
    return f"torch_sparse {getattr(torch_sparse, '__version__', '?')}"


def reference_pin(fmt, rowptr, row, col, vals, x, y):
    """The reference's own host loops, compiled in place (oracle/build_ref_host.sh): the vector is accepted only when
    they produce the SAME BITS as the oracle restatement -- integers and floats alike (same loop order, products and sums
    rounded separately, no reassociation at -O2 without -ffast-math)."""
    if not oracle.have_ref_host():
        return None
    same = lambda a: a.tobytes() == y.tobytes() and a.shape == y.shape
    if fmt == "CSR":
        assert same(oracle.ref_spmm_host_csr(rowptr, col, vals, x, variant="grande")), "oracle != reference spmm_host_csr (grande)"
        if vals is None:  # the default variant's loop reads the values and ignores them: unit weights only
            assert same(oracle.ref_spmm_host_csr(rowptr, col, None, x, variant="default")), "oracle != reference spmm_host_csr"
    else:
        assert same(oracle.ref_spmm_host_coo(row, col, vals, x, y.shape[0], variant="default")), "oracle != reference spmm_host_coo"
        assert same(oracle.ref_spmm_host_coo(row, col, vals, x, y.shape[0], variant="spmv")), "oracle != reference spmm_host (spmv)"
    return oracle.REF_HOST_PINNED_BY


QUANT_PINNED_BY = ("reference symmetric_quantize / symmetric_dequantize (models/quantize.py:20-42) and GCNConv.message_and_aggregate "
                   "(models/pyg_gcn_conv.py:130-137) cut out by name with ast and executed with torch only; the aggregation inside is the "
                   "reference's spmm_host_csr compiled in place (oracle/_ref/libref_host_grande_*)")


def reference_quantiser(root="/root/reference"):
    """{'symmetric_quantize', 'symmetric_dequantize', 'message_and_aggregate'}: the reference's own function bodies, compiled from
    the FunctionDef nodes of models/quantize.py and models/pyg_gcn_conv.py (class GCNConv) -- or None without /root/reference."""
    import ast

    qpath, cpath = os.path.join(root, "models", "quantize.py"), os.path.join(root, "models", "pyg_gcn_conv.py")
    if not (os.path.exists(qpath) and os.path.exists(cpath)):
        return None

    def cut(path, names, cls=None):
        tree = ast.parse(open(path).read(), filename=path)
        body = tree.body
        if cls is not None:
            body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
        nodes = [n for n in body if isinstance(n, ast.FunctionDef) and n.name in names]
        assert sorted(n.name for n in nodes) == sorted(names), (path, names)
        for n in nodes:  # (annotations name typing / torch_sparse symbols the cut leaves behind)
            n.returns = None
            for a in n.args.args + n.args.kwonlyargs:
                a.annotation = None
        return ast.fix_missing_locations(ast.Module(body=nodes, type_ignores=[]))

    class _NoSparseTensor:  # `isinstance(adj_t, SparseTensor)` in the layer: our adjacency object is not one, so `adj_t.mul` runs
        pass

    ns = {"torch": torch, "SparseTensor": _NoSparseTensor, "matmul": None}
    exec(compile(cut(qpath, ["symmetric_quantize", "symmetric_dequantize"]), qpath, "exec"), ns)
    exec(compile(cut(cpath, ["message_and_aggregate"], cls="GCNConv"), cpath, "exec"), ns)
    return {k: ns[k] for k in ("symmetric_quantize", "symmetric_dequantize", "message_and_aggregate")}


class RefAdjacency:
    """what the conv layer is handed as ``adj_t`` when the backend is on (inference.py -> prepare_pim_spmm): `.dtype` and
    `.mul(x_q)`; the product is the reference's own host loop (unit weights) on the quantised features"""

    def __init__(self, rowptr, col, dtype):
        self.rowptr, self.col, self.dtype = rowptr, col, dtype
        self.seen_xq = self.out_q = None

    def mul(self, x_q):
        self.seen_xq = x_q.numpy().copy()
        self.out_q = oracle.ref_spmm_host_csr(self.rowptr, self.col, None, self.seen_xq, variant="grande")
        return torch.from_numpy(self.out_q)


PINNED_BY = {"last": None}


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    if name.startswith("spmm_") or name.startswith("group_"):
        kw["pinned_by"] = PINNED_BY["last"] or ("unpinned: neither oracle/_ref/libref_host_* nor torch_sparse available in the build "
                                                 "container; accepted on agreement of the oracle with scipy.sparse and torch.sparse.mm")
    np.savez_compressed(path, **kw)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def emit(name, fmt, rowptr, col, x, vals=None, exact=True):
    npdt = x.dtype
    if fmt == "CSR":
        y = oracle.spmm_csr(rowptr, col, vals, x)
        cross_check("CSR", rowptr, None, col, vals, x, y, exact)
        PINNED_BY["last"] = " + ".join(filter(None, [reference_pin("CSR", rowptr, None, col, vals, x, y),
                                                      torch_sparse_pin("CSR", rowptr, None, col, vals, x, y, exact)])) or None
        kw = dict(fmt="CSR", rowptr=rowptr, col=col, x=x, y=y)
        if vals is not None:
            kw["vals"] = vals
    else:
        r, c, v = coalesce(rowptr, col, npdt)
        if vals is not None:  # weighted: keep given weights on the de-duplicated pattern
            v = (v * 0 + 1).astype(npdt) * vals[: len(v)]
        y = oracle.spmm_coo(r, c, v, x, len(rowptr) - 1)
        cross_check("COO", None, r, c, v, x, y, exact)
        PINNED_BY["last"] = " + ".join(filter(None, [reference_pin("COO", None, r, c, v, x, y),
                                                      torch_sparse_pin("COO", None, r, c, v, x, y, exact)])) or None
        kw = dict(fmt="COO", row=r, col=c, vals=v, x=x, y=y, nrows=len(rowptr) - 1)
    save(name, **kw)


def group_vectors(rng):
    """sp_parts x ds_parts (spmm.py:9-13,57-72,127-136) through spmm_host_csr_group / spmm_host_coo_group of the reference
    (spmm_default/ops.hpp:42-62,97-118), non-divisible splits included; the oracle's group() must give the same bits."""
    if not oracle.have_ref_host():
        print("oracle/_ref/libref_host_* missing: group vectors NOT regenerated")
        return
    n, h = 180, 27
    rowptr, col = random_csr(rng, n, n, 8, long_rows=[(3, 700)])
    a = sp.csr_matrix((np.ones(len(col), dtype=np.int64), col.copy(), rowptr.copy()), shape=(n, n))
    for fmt in ("CSR", "COO"):
        for name, sp_parts, ds_parts in (("INT32", 1, 1), ("INT32", 2, 1), ("INT32", 3, 4), ("INT8", 8, 3), ("FLT32", 3, 8),
                                         ("INT64", 2, 2), ("DBL64", 8, 1), ("INT16", 1, 4)):
            npdt = NP_DTYPES[name]
            x = rng.integers(-8, 4, size=(n, h)).astype(npdt) if np.issubdtype(npdt, np.integer) else \
                (rng.random((n, h)) * 2 - 1).astype(npdt)
            step = (n + sp_parts - 1) // sp_parts
            idx0, cols, vals, ncols = [], [], [], []
            for i in range(sp_parts):
                blk = a[:, i * step:min(n, (i + 1) * step)].tocsr()
                blk.sum_duplicates()
                blk.sort_indices()
                idx0.append((blk.indptr if fmt == "CSR" else blk.tocoo().row).astype(np.int32))
                cols.append(blk.indices.astype(np.int32))
                # CSR: the default variant's spmm_host_csr ignores the stored values (spmm_mul_csr.c:108-109): unit weights
                vals.append(np.ones(blk.nnz, dtype=npdt) if fmt == "CSR" else blk.data.astype(npdt))
                ncols.append(blk.shape[1])
            xs = [np.ascontiguousarray(c) for c in np.array_split(x, ds_parts, axis=1) if c.shape[1] > 0]
            y = oracle.ref_group(fmt == "COO", idx0, cols, vals, [n] * sp_parts, ncols, xs, h, variant="default")
            mine = oracle.group(fmt == "COO", idx0, cols, vals, [n] * sp_parts, ncols, xs, h)
            assert mine.tobytes() == y.tobytes(), "oracle.group != reference group driver"
            if fmt == "COO":
                assert oracle.ref_group(True, idx0, cols, vals, [n] * sp_parts, ncols, xs, h, variant="spmv").tobytes() == y.tobytes()
            PINNED_BY["last"] = "reference spmm_host_csr_group / spmm_host_coo_group (spmm_default/ops.hpp:42-62,97-118) compiled in place"
            kw = dict(fmt=fmt, x=x, y=y, n_parts=sp_parts, ds_parts=ds_parts, ncols=np.array(ncols, dtype=np.int32))
            for i in range(sp_parts):
                kw[f"idx0_{i}"], kw[f"col_{i}"], kw[f"vals_{i}"] = idx0[i], cols[i], vals[i]
            save(f"group_{fmt.lower()}_{name}_sp{sp_parts}_ds{ds_parts}", **kw)


def quant_vectors():
    # (6) quantise -> aggregate -> dequantise of the conv layers (pyg_gcn_conv.py:130-137, quantize.py:20-42): outputs of the
    # REFERENCE'S OWN functions.  models/quantize.py cannot be imported (line 5: `from torch_sparse import ...`, an un-vendored
    # package), but symmetric_quantize / symmetric_dequantize need torch alone, and GCNConv.message_and_aggregate needs those two
    # plus an adjacency object with `.dtype` and `.mul`: the three FunctionDef nodes are cut out of the reference files BY NAME with
    # `ast` (as oracle/build_ref_host.sh cuts the C loops), compiled in memory and executed here -- nothing of the reference is
    # written to disk.  The adjacency's `.mul` is the reference's own host loop compiled in place (oracle/_ref/libref_host_*).
    # The repo's restatements (oracle.symmetric_*, pygim_amd.quantize) are then CHECKED against these outputs, not the other way round.
    ref_fns = reference_quantiser()
    rng6 = np.random.default_rng(66)
    rowptr, col = random_csr(rng6, 256, 256, 14, empty_frac=0.1, long_rows=[(5, 1200)])
    x = (rng6.standard_normal((256, 40)) * 2.5).astype(np.float32)
    for name in ("INT8", "INT16", "INT32", "FLT32"):
        npdt = NP_DTYPES[name]
        if ref_fns is None or not oracle.have_ref_host():
            print(f"quant_gcn_{name}: /root/reference or oracle/_ref missing -- NOT regenerated")
            continue
        adj = RefAdjacency(rowptr, col, TORCH_OF[name])
        out_t = ref_fns["message_and_aggregate"](None, adj, torch.from_numpy(x))   # the layer's own call order, pyg_gcn_conv.py:130-137
        s_t, xq_t = ref_fns["symmetric_quantize"](torch.from_numpy(x), dtype=TORCH_OF[name])
        assert np.array_equal(adj.seen_xq, xq_t.numpy()) and adj.seen_xq.dtype == npdt
        scale, xq, out_q, out = np.float32(s_t.item()), xq_t.numpy(), adj.out_q, out_t.numpy()
        assert out.dtype == np.float32
        # the repo's statements of the same lines against the reference's output
        s_o, xq_o = oracle.symmetric_quantize(x, npdt)
        assert np.float32(s_o) == scale and xq_o.tobytes() == xq.tobytes(), "oracle.symmetric_quantize != reference"
        assert oracle.spmm_csr(rowptr, col, None, xq).tobytes() == out_q.tobytes()
        assert oracle.symmetric_dequantize(out_q, 1.0, s_o).tobytes() == out.tobytes(), "oracle.symmetric_dequantize != reference"
        from pygim_amd import quantize as qz

        s_p, xq_p = qz.symmetric_quantize(torch.from_numpy(x), TORCH_OF[name])
        assert np.float32(s_p.item()) == scale and np.array_equal(xq_p.numpy(), xq), "pygim_amd.quantize != reference"
        assert np.array_equal(qz.symmetric_dequantize(torch.from_numpy(out_q), 1.0, s_p).numpy(), out)
        save(f"quant_gcn_{name}", rowptr=rowptr, col=col, x=x, scale=scale, xq=xq, out_q=out_q, out=out,
             pinned_by=QUANT_PINNED_BY)


def main():
    for f in os.listdir(HERE):
        if f.endswith(".npz"):
            os.remove(os.path.join(HERE, f))
    # (1) Cora-shaped, h = 32, every dtype, driver features under torch.manual_seed
    rowptr_t, col_t = synth.make_shape("cora", seed=0)
    rowptr, col = rowptr_t.numpy(), col_t.numpy()
    n = len(rowptr) - 1
    for name, tdt in TORCH_OF.items():
        torch.manual_seed(2708)
        x = torch.randint(-8, 4, (n, 32)).to(tdt).numpy()  # == torch.randint(-2^6, 2^6, ...) of the driver
        emit(f"spmm_cora_csr_{name}", "CSR", rowptr, col, x)
        emit(f"spmm_cora_coo_{name}", "COO", rowptr, col, x)
    rng = np.random.default_rng(7)
    # (2) overflow: rows of degree >= 300 so int8 / int16 sums wrap
    rowptr, col = random_csr(rng, 64, 512, 40, long_rows=[(3, 900), (40, 5000)])
    for name in ("INT8", "INT16"):
        x = rng.integers(-128 if name == "INT8" else -20000, 127 if name == "INT8" else 20000,
                         size=(512, 24)).astype(NP_DTYPES[name])
        emit(f"spmm_wrap_csr_{name}", "CSR", rowptr, col, x)
        emit(f"spmm_wrap_coo_{name}", "COO", rowptr, col, x)
    # (3) duplicate-heavy multigraph: COO values > 1 after coalesce, CSR keeps duplicates
    rowptr, col = random_csr(rng, 200, 16, 12)
    x = rng.integers(-8, 4, size=(16, 9)).astype(np.int32)
    emit("spmm_dups_csr_INT32", "CSR", rowptr, col, x)
    emit("spmm_dups_coo_INT32", "COO", rowptr, col, x)
    # (4) ragged: empty rows, one 5000-entry row, odd widths
    rowptr, col = random_csr(rng, 300, 300, 6, empty_frac=0.4, long_rows=[(17, 5000)])
    for h in (1, 9, 32, 100, 256):
        x = rng.integers(-8, 4, size=(300, h)).astype(np.int32)
        emit(f"spmm_ragged_csr_h{h}_INT32", "CSR", rowptr, col, x)
    x = rng.integers(-8, 4, size=(300, 100)).astype(np.int64)
    emit("spmm_ragged_coo_h100_INT64", "COO", rowptr, col, x)
    # (5) non-integer floats and real-valued weights (tolerance cases)
    rowptr, col = random_csr(rng, 256, 256, 20, long_rows=[(9, 3000)])
    for name in ("FLT32", "DBL64"):
        npdt = NP_DTYPES[name]
        x = (rng.random((256, 48)) * 2 - 1).astype(npdt)
        vals = (rng.random(len(col)) * 2 - 1).astype(npdt)
        emit(f"spmm_real_csr_{name}", "CSR", rowptr, col, x, vals=vals, exact=False)
    # (5b) valued entries for every type (grande multiplies by the stored value, spmm_grande/spmm_mul_csr.c:131; integer
    # products wrap at the element width), CSR and COO
    rowptr, col = random_csr(rng, 400, 300, 10, long_rows=[(11, 900)])
    for name, npdt in NP_DTYPES.items():
        if np.issubdtype(npdt, np.integer):
            lo, hi = (-128, 127) if name == "INT8" else (-3000, 3000)
            vals = rng.integers(lo, hi, size=len(col)).astype(npdt)
            x = rng.integers(lo, hi, size=(300, 20)).astype(npdt)
        else:
            vals = (rng.random(len(col)) * 2 - 1).astype(npdt)
            x = (rng.random((300, 20)) * 2 - 1).astype(npdt)
        emit(f"spmm_valued_csr_{name}", "CSR", rowptr, col, x, vals=vals, exact=np.issubdtype(npdt, np.integer))
    # (5c) group products: sparse column blocks x dense feature blocks, outputs of the reference's OWN group drivers
    group_vectors(rng)
    # partition vectors from the reference's own partition.c
    if oracle.have_ref():
        kw, k = {}, 0
        for trial in range(12):
            nrows = int(rng.integers(1, 500))
            rp, _ = random_csr(rng, nrows, 64, float(rng.uniform(0.3, 30)), empty_frac=0.3)
            for nparts in (2, 8, 64):
                kw[f"rowptr_{k}"] = rp
                kw[f"nparts_{k}"] = nparts
                kw[f"by_nnz_{k}"] = oracle.ref_partition_by_nnz_csr(rp, nparts)
                kw[f"by_row_{k}"] = oracle.ref_partition_by_row_csr(nrows, nparts)
                k += 1
        save("partition_ref", n_cases=k, **kw)
    else:
        print("oracle/_ref missing: partition vectors NOT regenerated")
    quant_vectors()
    # (7) MatrixMarket reader: outputs of the REFERENCE's own readCOOMatrix + coo2csr (utils.hpp:15-127, compiled in place
    # into oracle/_ref/libref_utils.so) on small files; the file text travels in the fixture
    if oracle.have_ref_utils():
        import tempfile

        rng7 = np.random.default_rng(77)
        kw = {}
        for k, (nr_, nc_) in enumerate(((12, 9), (33, 40), (7, 7), (50, 3))):
            nnz_ = 4 * nr_
            rr, cc = rng7.integers(0, nr_, nnz_), rng7.integers(0, nc_, nnz_)
            text = "%%MatrixMarket matrix coordinate real general\n% generated by tests/golden/make_golden.py\n"
            text += f"{nr_} {nc_} {nnz_}\n" + "".join(f"{a + 1} {b + 1} {rng7.integers(1, 9)}.5\n" for a, b in zip(rr, cc))
            with tempfile.NamedTemporaryFile("w", suffix=".mtx", delete=False) as f:
                f.write(text)
            n_, m_, rp_, ci_, va_ = oracle.ref_read_matrix_csr(f.name)
            os.unlink(f.name)
            kw[f"text_{k}"] = np.frombuffer(text.encode(), dtype=np.uint8)
            kw[f"shape_{k}"] = np.array([n_, m_], dtype=np.int64)
            kw[f"rowptr_{k}"], kw[f"colind_{k}"], kw[f"values_{k}"] = rp_.astype(np.int32), ci_.astype(np.int32), va_
        save("mtx_ref", n_cases=4, **kw)
    else:
        print("oracle/_ref/libref_utils.so missing: mtx vectors NOT regenerated")


if __name__ == "__main__":
    if sys.argv[1:] == ["quant"]:   # only the conv layers' quantiser vectors
        quant_vectors()
    else:
        main()
