"""``torch.ops.pim_ops`` -- the reference's custom-op namespace, backed by the HIP C ABI.

The reference registers these ops from C++ (``TORCH_LIBRARY(pim_ops, m)``:
spmm_default/pytorch_api.cpp:372-389, spmm_grande/pytorch_api.cpp:326-343,
spmv_sparseP/pytorch_api.cpp:271-287) and selects the variant by which
``libbackend_pim.so`` ``--lib_path`` points at.  Here one HIP library serves all
three; ``load(variant)`` (or ``load_library(path)`` with a reference-style path)
registers that variant's schemas.  Same names, same argument meaning:

    dpu_init_ranks(int) -> () | int[]      dpu_init_dpus(int) -> () | int[]
    dpu_release() -> ()                    spmm_free_group(int) -> ()
    spmm_csr_to_device_group(Tensor[], Tensor[], Tensor[], int[], int[], int[]|Tensor[], int) -> int
    spmm_csr_run_group(int, Tensor[]) -> Tensor          (+ the _coo_ twins, default only)
    spmv_coo_to_device_group(..., int h_size, int ranks_per_spmv=1) -> int
    spmv_coo_run_group(int, Tensor[]) -> Tensor

Tensors may live on the CPU (the reference's only mode; data is staged through HBM
per call) or on the HIP device (no copies; the result stays on the device).
"""
from __future__ import annotations

import types

import torch

from . import _lib

DTYPE_CODE = {
    torch.int8: _lib.INT8, torch.int16: _lib.INT16, torch.int32: _lib.INT32,
    torch.int64: _lib.INT64, torch.float32: _lib.FLT32, torch.float64: _lib.DBL64,
}
VARIANTS = ("spmm", "grande", "spmv")

_library = None
_variant = None
_groups = {}  # handle -> dict(keepalive tensors, dtype, shapes)


def _log_timers(handle, on_host: bool) -> None:
    """the reference prints its Timer buckets after every run (spmm_mul_csr.c:563-580, parsed by
    utils/experiment.py:468-491); same keys, in milliseconds, when PYGIM_DATA_LOG=1 (the driver scripts set it)"""
    import os

    if not on_host or os.environ.get("PYGIM_DATA_LOG", "0") != "1":
        return
    t = _lib.group_timers(handle)
    print(f"[DATA]load_sparse_time: {t[4]:.3f}\n[DATA]load_dense_time: {t[0]:.3f}\n[DATA]kernel_time: {t[1]:.3f}\n"
          f"[DATA]retrieve_result_time: {t[2]:.3f}\n[DATA]alignment_time: {t[3]:.3f}", flush=True)


def _new_out(shape, dtype, dev):
    """the result tensor of a run: on the operands' device; for host operands page-locked host memory (from torch's caching
    host allocator), so the device-to-host copy of the result runs at PCIe speed instead of faulting fresh pages in
    (Reddit h = 256 f32: 18 -> 5 ms).  PYGIM_PINNED_OUT=0 returns ordinary pageable memory."""
    import os

    if dev.type == "cpu" and torch.cuda.is_available() and os.environ.get("PYGIM_PINNED_OUT", "1") != "0":
        return torch.empty(shape, dtype=dtype, pin_memory=True)
    return torch.empty(shape, dtype=dtype, device=dev)


def _stream_of(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


def _as_index(t: torch.Tensor, what: str) -> torch.Tensor:
    # the reference reinterprets int32 tensors as uint32_t* (pytorch_api.cpp:230-231) and
    # throws a c10::Error on any other dtype
    if t.dtype != torch.int32:
        raise RuntimeError(f"expected scalar type Int but found {t.dtype} for {what}")
    return t.contiguous()


def _to_device_group(fmt, idx0, colind, values, nrows, ncols, n_dense, dense_cols, h_size):
    if not (len(idx0) == len(colind) == len(values) == len(nrows) == len(ncols)):
        raise RuntimeError("part lists differ in length")
    dt = values[0].dtype
    if dt not in DTYPE_CODE:
        raise RuntimeError(f"unsupported value dtype {dt}")
    idx0 = [_as_index(t, "row indices") for t in idx0]
    colind = [_as_index(t, "col indices") for t in colind]
    vals = []
    for v in values:
        if v.dtype != dt:
            raise RuntimeError(f"expected scalar type {dt} but found {v.dtype}")
        vals.append(v.contiguous())
    dev = vals[0].device
    for t in idx0 + colind + vals:
        if t.device != dev:
            raise RuntimeError("all sparse arrays of a group must live on the same device")
    handle = _lib.group_create(
        fmt, DTYPE_CODE[dt], [t.data_ptr() for t in idx0], [t.data_ptr() for t in colind],
        [t.data_ptr() for t in vals], nrows, ncols, [v.numel() for v in vals], n_dense, dense_cols, h_size)
    # device arrays are used in place by the library: hold them (the reference relies on the
    # Python wrapper's self.csr / self.row_indices for the same purpose, spmm.py:52, 89-91)
    _groups[handle] = dict(keep=(idx0, colind, vals) if dev.type != "cpu" else (), dtype=dt,
                           rows=int(nrows[0]), cols=int(sum(ncols)), h=int(h_size),
                           n_dense=[int(x) for x in n_dense], dense_cols=[int(x) for x in dense_cols],
                           part_cols=[int(x) for x in ncols])
    return handle


def _group(handle):
    g = _groups.get(int(handle))
    if g is None:
        raise RuntimeError(f"unknown sparse group handle {handle}")
    return g


def _check_dense(g, parts, count):
    if len(parts) != count:
        raise RuntimeError(f"expected {count} dense parts, got {len(parts)}")
    for b in parts:
        if b.dtype != g["dtype"]:
            raise RuntimeError(f"expected scalar type {g['dtype']} but found {b.dtype}")
    dev = parts[0].device
    if any(b.device != dev for b in parts):
        raise RuntimeError("dense parts must live on one device")
    return dev


# ---- op bodies -----------------------------------------------------------------
def _dpu_init_ranks_void(nr_ranks: int) -> None:
    _lib.init_ranks(nr_ranks)


def _dpu_init_ranks_list(nr_ranks: int):
    return _lib.init_ranks(nr_ranks, want_units=True)


def _dpu_init_dpus_void(nr_dpus: int) -> None:
    _lib.init_units(nr_dpus)


def _dpu_init_dpus_list(nr_dpus: int):
    return _lib.init_units(nr_dpus)


def _dpu_release() -> None:
    _groups.clear()
    _lib.release()


def _spmm_free_group(handle: int) -> None:
    _groups.pop(int(handle), None)
    _lib.group_free(handle)


def _spmm_csr_to_device_group(row_indices, col_indices, values, nrows, ncols, dense_cols, h_size):
    n = len(ncols)
    return _to_device_group(_lib.CSR, row_indices, col_indices, values, nrows, ncols, [len(dense_cols)] * n,
                            list(dense_cols) * n, h_size)


def _spmm_coo_to_device_group(row_indices, col_indices, values, nrows, ncols, dense_cols, h_size):
    n = len(ncols)
    return _to_device_group(_lib.COO, row_indices, col_indices, values, nrows, ncols, [len(dense_cols)] * n,
                            list(dense_cols) * n, h_size)


def _spmm_run_group(handle, B_parts):
    g = _group(handle)
    if len(set(g["n_dense"])) != 1:
        raise RuntimeError("group was created with a per-part dense split (grande layout)")
    nd = g["n_dense"][0]
    dev = _check_dense(g, B_parts, nd)
    parts = []
    for j, b in enumerate(B_parts):
        if b.dim() != 2 or b.size(0) != g["cols"] or b.size(1) != g["dense_cols"][j]:
            raise RuntimeError(f"dense part {j} has shape {tuple(b.shape)}, expected ({g['cols']}, {g['dense_cols'][j]})")
        parts.append(b.contiguous())
    out = _new_out((g["rows"], g["h"]), g["dtype"], dev)
    _lib.spmm_run_group(handle, [b.data_ptr() for b in parts], out.data_ptr(), _stream_of(out))
    _log_timers(handle, not out.is_cuda)
    return out


def _grande_csr_to_device_group(row_indices, col_indices, values, nrows, ncols, dense_cols, h_size):
    # grande: dense_cols is one int32 tensor of per-window widths per sparse part
    # (spmm_grande/pytorch_api.cpp:226, 240; backend_pim/grande.py:64-72)
    n_dense, flat = [], []
    for t in dense_cols:
        if t.dtype != torch.int32:
            raise RuntimeError(f"expected scalar type Int but found {t.dtype} for dense_cols")
        w = [int(x) for x in t.cpu().tolist()]
        n_dense.append(len(w))
        flat += w
    return _to_device_group(_lib.CSR, row_indices, col_indices, values, nrows, ncols, n_dense, flat, h_size)


def _grande_run_group(handle, B_parts):
    g = _group(handle)
    dev = _check_dense(g, B_parts, sum(g["n_dense"]))
    parts = [b.contiguous() for b in B_parts]
    owner = [i for i, nd in enumerate(g["n_dense"]) for _ in range(nd)]  # sparse part of every window
    for k, b in enumerate(parts):
        if b.dim() != 2 or b.size(0) != g["part_cols"][owner[k]] or b.size(1) < g["dense_cols"][k]:
            raise RuntimeError(f"window {k} has shape {tuple(b.shape)}, expected ({g['part_cols'][owner[k]]}, "
                               f">= {g['dense_cols'][k]})")
    out = _new_out((g["rows"], g["h"]), g["dtype"], dev)
    _lib.grande_run_group(handle, [b.data_ptr() for b in parts], [b.size(1) for b in parts], out.data_ptr(),
                          _stream_of(out))
    _log_timers(handle, not out.is_cuda)
    return out


def _spmv_coo_to_device_group(row_indices, col_indices, values, nrows, ncols, dense_cols, h_size, ranks_per_spmv=1):
    # ranks_per_spmv spreads one SpMV over several UPMEM ranks (spmv_sparseP/ops.hpp:93-113);
    # one GPU already runs the whole vector, so the knob only validates.
    if ranks_per_spmv < 1:
        raise RuntimeError("ranks_per_spmv must be >= 1")
    n = len(ncols)
    return _to_device_group(_lib.COO, row_indices, col_indices, values, nrows, ncols, [len(dense_cols)] * n,
                            list(dense_cols) * n, h_size)


def _spmv_run_group(handle, B_parts):
    g = _group(handle)
    nd = g["n_dense"][0]
    dev = _check_dense(g, B_parts, nd)
    vecs = []
    for j, b in enumerate(B_parts):
        if b.numel() != g["cols"]:
            raise RuntimeError(f"vector {j} has {b.numel()} elements, expected {g['cols']}")
        vecs.append(b.contiguous())
    out = _new_out((g["rows"], nd), g["dtype"], dev)
    _lib.spmv_run_group(handle, [b.data_ptr() for b in vecs], out.data_ptr(), _stream_of(out))
    _log_timers(handle, not out.is_cuda)
    return out


# ---- MatrixMarket debug loaders of the default variant (spmm_default/utils.hpp:139-173) -------------
def _read_mtx_csr(filename):
    """The reference's reader, restated (spmm_default/utils.hpp:16-70 readCOOMatrix, :86-127 coo2csr):
    skip '%' lines; first data line = rows cols nnz; row and column counts padded up to even; 1-based
    indices; the value column is IGNORED (every stored value is 1); entries keep file order inside a row."""
    import numpy as np

    rows, cols = [], []
    shape = None
    with open(filename) as fh:
        for line in fh:
            tok = line.split()
            if not tok or tok[0].startswith("%"):
                continue
            if shape is None:
                shape = (int(tok[0]), int(tok[1]), int(tok[2]))
                continue
            rows.append(int(tok[0]) - 1)
            cols.append(int(tok[1]) - 1)
    nrows, ncols, nnz = shape
    nrows += nrows % 2
    ncols += ncols % 2
    r = np.asarray(rows[:nnz], dtype=np.int64)
    c = np.asarray(cols[:nnz], dtype=np.int64)
    order = np.argsort(r, kind="stable")
    rowptr = np.zeros(nrows + 1, dtype=np.int64)
    np.cumsum(np.bincount(r, minlength=nrows), out=rowptr[1:])
    return types.SimpleNamespace(indptr=rowptr, indices=c[order], data=np.ones(len(order), dtype=np.int32),
                                 shape=(nrows, ncols))


def _read_matrix_rowptr(filename: str):
    return torch.from_numpy(_read_mtx_csr(filename).indptr.astype("int32"))


def _read_matrix_colind(filename: str):
    return torch.from_numpy(_read_mtx_csr(filename).indices.astype("int32"))


def _read_matrix_values(filename: str):
    return torch.from_numpy(_read_mtx_csr(filename).data.astype("int32"))  # int32 view, like the reference


def _read_matrix_nrows(filename: str):
    return int(_read_mtx_csr(filename).shape[0])


def _read_matrix_ncols(filename: str):
    return int(_read_mtx_csr(filename).shape[1])


_SCHEMAS = {
    "spmm": [
        ("dpu_init_ranks(int nr_ranks) -> ()", _dpu_init_ranks_void),
        ("dpu_init_dpus(int nr_dpus) -> ()", _dpu_init_dpus_void),
        ("dpu_release() -> ()", _dpu_release),
        ("spmm_free_group(int sp_group_ptr) -> ()", _spmm_free_group),
        ("spmm_csr_to_device_group(Tensor[] row_indices, Tensor[] col_indices, Tensor[] values, int[] nrows, "
         "int[] ncols, int[] dense_cols, int h_size) -> int", _spmm_csr_to_device_group),
        ("spmm_csr_run_group(int sp_group_ptr, Tensor[] B_parts) -> Tensor", _spmm_run_group),
        ("spmm_coo_to_device_group(Tensor[] row_indices, Tensor[] col_indices, Tensor[] values, int[] nrows, "
         "int[] ncols, int[] dense_cols, int h_size) -> int", _spmm_coo_to_device_group),
        ("spmm_coo_run_group(int sp_group_ptr, Tensor[] B_parts) -> Tensor", _spmm_run_group),
        ("read_matrix_rowptr(str filename) -> Tensor", _read_matrix_rowptr),
        ("read_matrix_colind(str filename) -> Tensor", _read_matrix_colind),
        ("read_matrix_values(str filename) -> Tensor", _read_matrix_values),
        ("read_matrix_nrows(str filename) -> int", _read_matrix_nrows),
        ("read_matrix_ncols(str filename) -> int", _read_matrix_ncols),
    ],
    "grande": [
        ("dpu_init_ranks(int nr_ranks) -> int[]", _dpu_init_ranks_list),
        ("dpu_init_dpus(int nr_dpus) -> int[]", _dpu_init_dpus_list),
        ("dpu_release() -> ()", _dpu_release),
        ("spmm_free_group(int sp_group_ptr) -> ()", _spmm_free_group),
        ("spmm_csr_to_device_group(Tensor[] row_indices, Tensor[] col_indices, Tensor[] values, int[] nrows, "
         "int[] ncols, Tensor[] dense_cols, int h_size) -> int", _grande_csr_to_device_group),
        ("spmm_csr_run_group(int sp_group_ptr, Tensor[] B_parts) -> Tensor", _grande_run_group),
        # (the grande library registers the .mtx debug ops too: spmm_grande/pytorch_api.cpp:338-342, the same utils.hpp)
        ("read_matrix_rowptr(str filename) -> Tensor", _read_matrix_rowptr),
        ("read_matrix_colind(str filename) -> Tensor", _read_matrix_colind),
        ("read_matrix_values(str filename) -> Tensor", _read_matrix_values),
        ("read_matrix_nrows(str filename) -> int", _read_matrix_nrows),
        ("read_matrix_ncols(str filename) -> int", _read_matrix_ncols),
    ],
    "spmv": [
        ("dpu_init_ranks(int nr_ranks) -> ()", _dpu_init_ranks_void),
        ("dpu_init_dpus(int nr_dpus) -> ()", _dpu_init_dpus_void),
        ("dpu_release() -> ()", _dpu_release),
        ("spmm_free_group(int sp_group_ptr) -> ()", _spmm_free_group),
        ("spmv_coo_to_device_group(Tensor[] row_indices, Tensor[] col_indices, Tensor[] values, int[] nrows, "
         "int[] ncols, int[] dense_cols, int h_size, int ranks_per_spmv=1) -> int", _spmv_coo_to_device_group),
        ("spmv_coo_run_group(int sp_group_ptr, Tensor[] B_parts) -> Tensor", _spmv_run_group),
    ],
}


def current_variant():
    return _variant


def load(variant: str = "spmm"):
    """Register ``torch.ops.pim_ops`` with the schemas of one reference variant.

    Fails (ImportError) when the HIP library is not built: there is no CPU fallback.
    Only one variant is live per process at a time, like the reference (one
    ``libbackend_pim.so`` per process); loading another one replaces the first.
    """
    global _library, _variant
    if variant not in VARIANTS:
        raise ValueError(f"variant must be one of {VARIANTS}")
    _lib.lib()  # dlopen now: a missing extension must not go unnoticed
    if _variant == variant:
        return
    if _library is not None:
        if _library == "native":
            raise RuntimeError("a libbackend_pim.so is already loaded in this process (one variant per process, like the reference)")
        _library._destroy()
        _library = None
    lib = torch.library.Library("pim_ops", "DEF")
    for schema, fn in _SCHEMAS[variant]:
        lib.define(schema)
        lib.impl(schema.split("(")[0], fn, "CompositeExplicitAutograd")
    _library, _variant = lib, variant


def variant_of(path: str) -> str:
    """variant a reference-style library path names (``.../spmm_grande/build/libbackend_pim.so``)"""
    p = path.lower()
    if "grande" in p:
        return "grande"
    if "spmv" in p or "sparsep" in p:
        return "spmv"
    return "spmm"


def load_library(path: str):
    """``torch.ops.load_library(args.lib_path)`` counterpart (spmm_test.py:111, inference.py:134).

    An existing shared object is loaded exactly that way (the TORCH_LIBRARY shim over the C ABI that
    ``make -C pygim_amd/csrc shims`` puts at the reference's paths); a path that does not exist selects
    the variant by its directory name and registers the same ops from Python over the same C ABI.
    """
    global _library, _variant
    import os

    path = path.strip('"')
    variant = variant_of(path)
    if not os.path.isfile(path):
        return load(variant)
    _lib.lib()  # libpygim_hip.so (and torch's HIP runtime) first; fails loudly when not built
    if _variant == variant:
        return
    if _library is not None:
        if _library == "native":
            raise RuntimeError("a libbackend_pim.so is already loaded in this process (one variant per process, like the reference)")
        _library._destroy()
    torch.ops.load_library(path)
    _library, _variant = "native", variant
