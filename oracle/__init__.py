"""numpy front-end of the CPU oracle (oracle/spmm_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from pygim_amd/.  Parity status is stated in
the header of spmm_oracle.c: PINNED -- arithmetic, merge helpers and group drivers against
the reference's own host loops, partitioning against its partition.c, all compiled in place
into oracle/_ref (oracle/Makefile, oracle/build_ref_host.sh).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libref_partition.so")

SUFFIX = {
    np.dtype(np.int8): "i8",
    np.dtype(np.int16): "i16",
    np.dtype(np.int32): "i32",
    np.dtype(np.int64): "i64",
    np.dtype(np.float32): "f32",
    np.dtype(np.float64): "f64",
}


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "spmm_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    elif not os.path.exists(_REF_PATH):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(_HERE, "_ref", "libref_utils.so")) and os.path.isdir("/root/reference"):
        subprocess.call(["make", "-C", _HERE, "ref_utils"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if not have_ref_host() and os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-C", _HERE, "ref_host"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _sfx(x):
    return SUFFIX[np.dtype(x.dtype)]


def spmm_csr(rowptr, colind, values, x, ldx=None):
    """y = A @ x in the reference's row->feature->entry order; values None = ones."""
    x = np.ascontiguousarray(x)
    rowptr, colind = _u32(rowptr), _u32(colind)
    nrows, ncols = len(rowptr) - 1, x.shape[1] if ldx is None else None
    if ldx is None:
        ldx = x.shape[1]
    else:
        ncols = x.shape[1]
    if values is not None:
        values = np.ascontiguousarray(values, dtype=x.dtype)
    y = np.zeros((nrows, ncols), dtype=x.dtype)
    getattr(lib(), "oracle_spmm_csr_" + _sfx(x))(
        _p(y), ctypes.c_uint32(nrows), _p(rowptr), _p(colind), _p(values), _p(x),
        ctypes.c_uint32(ncols), ctypes.c_uint32(ldx))
    return y


def spmm_coo(row, col, values, x, nrows):
    x = np.ascontiguousarray(x)
    row, col = _u32(row), _u32(col)
    if values is not None:
        values = np.ascontiguousarray(values, dtype=x.dtype)
    y = np.zeros((nrows, x.shape[1]), dtype=x.dtype)
    getattr(lib(), "oracle_spmm_coo_" + _sfx(x))(
        _p(y), ctypes.c_uint32(len(row)), _p(row), _p(col), _p(values), _p(x),
        ctypes.c_uint32(x.shape[1]))
    return y


def spmm_csr_rowpar(rowptr, colind, values, x, nthreads=1, out=None):
    """Row-parallel CSR product (the cpu_baseline 'port'); same sums, same order."""
    x = np.ascontiguousarray(x)
    rowptr, colind = _u32(rowptr), _u32(colind)
    nrows = len(rowptr) - 1
    if values is not None:
        values = np.ascontiguousarray(values, dtype=x.dtype)
    y = np.zeros((nrows, x.shape[1]), dtype=x.dtype) if out is None else out
    getattr(lib(), "oracle_spmm_csr_rowpar_" + _sfx(x))(
        _p(y), ctypes.c_uint32(nrows), _p(rowptr), _p(colind), _p(values), _p(x),
        ctypes.c_uint32(x.shape[1]), ctypes.c_int(nthreads))
    return y


def group(is_coo, ptr_or_row, colind, values, nrows, ncols, x_parts, h):
    """Group product over sparse column blocks x dense feature blocks.

    ptr_or_row / colind / values: lists (one per sparse part); values may be None.
    x_parts: list of [total_cols, h_j] arrays.  Returns out[nrows[0], h].
    """
    n = len(colind)
    dt = x_parts[0].dtype
    pr = [_u32(a) for a in ptr_or_row]
    ci = [_u32(a) for a in colind]
    vals = None if values is None else [np.ascontiguousarray(v, dtype=dt) for v in values]
    xs = [np.ascontiguousarray(x) for x in x_parts]
    VP = ctypes.c_void_p
    arr = lambda lst: (VP * len(lst))(*[a.ctypes.data for a in lst])
    nnz = np.array([len(c) for c in ci], dtype=np.uint32)
    nr = np.array(nrows, dtype=np.uint32)
    nc = np.array(ncols, dtype=np.uint32)
    dn = np.array([x.shape[1] for x in xs], dtype=np.uint32)
    out = np.zeros((int(nr[0]), h), dtype=dt)
    getattr(lib(), "oracle_group_" + SUFFIX[np.dtype(dt)])(
        _p(out), ctypes.c_int(1 if is_coo else 0), ctypes.c_uint32(n), arr(pr), arr(ci),
        None if vals is None else arr(vals), _p(nr), _p(nc), _p(nnz),
        ctypes.c_uint32(len(xs)), arr(xs), _p(dn), ctypes.c_uint32(h))
    return out


def partition_by_row(nrows, nparts):
    split = np.zeros(nparts + 1, dtype=np.uint32)
    lib().oracle_partition_by_row(ctypes.c_uint32(nrows), _p(split), ctypes.c_int(nparts))
    return split


def partition_by_nnz(rowptr, nparts):
    rowptr = _u32(rowptr)
    split = np.zeros(nparts + 1, dtype=np.uint32)
    lib().oracle_partition_by_nnz(ctypes.c_uint32(len(rowptr) - 1), _p(rowptr), _p(split),
                                  ctypes.c_int(nparts))
    return split


def partition_equal_nnz(nnz, nparts):
    split = np.zeros(nparts + 1, dtype=np.uint32)
    lib().oracle_partition_equal_nnz(ctypes.c_uint32(nnz), _p(split), ctypes.c_int(nparts))
    return split


def max_threads():
    return int(lib().oracle_max_threads())


# --------------------------------------------------------------------------- #
# oracle/_ref : the reference's own support/partition.c, compiled in place.    #
# --------------------------------------------------------------------------- #
class _RefCSR(ctypes.Structure):
    # support/matrix.h:23-33
    _fields_ = [("nrows", ctypes.c_uint32), ("ncols", ctypes.c_uint32), ("nnz", ctypes.c_uint32),
                ("rowptr", ctypes.c_void_p), ("colind", ctypes.c_void_p), ("values", ctypes.c_void_p),
                ("rowptr_size", ctypes.c_uint32), ("colind_size", ctypes.c_uint32),
                ("values_size", ctypes.c_uint32)]


class _RefCOO(ctypes.Structure):
    # support/matrix.h:10-19
    _fields_ = [("nrows", ctypes.c_uint32), ("ncols", ctypes.c_uint32), ("nnz", ctypes.c_uint32),
                ("rows", ctypes.c_void_p), ("rowind", ctypes.c_void_p), ("colind", ctypes.c_void_p),
                ("val", ctypes.c_void_p), ("nnz_size", ctypes.c_uint32)]


def have_ref():
    return os.path.exists(_REF_PATH)


def ref_partition_by_nnz_csr(rowptr, nparts):
    ref = ctypes.CDLL(_REF_PATH)
    rowptr = _u32(rowptr)
    m = _RefCSR(len(rowptr) - 1, 0, int(rowptr[-1]), rowptr.ctypes.data, None, None, len(rowptr), 0, 0)
    # the reference writes split[split_cnt] before checking bounds only up to nparts; give slack
    split = np.zeros(nparts + 2, dtype=np.uint32)
    ref.partition_by_nnz_csr(ctypes.byref(m), _p(split), ctypes.c_int(nparts))
    return split[: nparts + 1]


def ref_partition_by_row_csr(nrows, nparts):
    ref = ctypes.CDLL(_REF_PATH)
    m = _RefCSR(nrows, 0, 0, None, None, None, 0, 0, 0)
    split = np.zeros(nparts + 2, dtype=np.uint32)
    ref.partition_by_row_csr(ctypes.byref(m), _p(split), ctypes.c_int(nparts))
    return split[: nparts + 1]


def ref_partition_by_nnz_rgrn_coo(rows_hist, nparts):
    ref = ctypes.CDLL(_REF_PATH)
    rows_hist = _u32(rows_hist)
    m = _RefCOO(len(rows_hist), 0, int(rows_hist.sum()), rows_hist.ctypes.data, None, None, None, 0)
    split = np.zeros(nparts + 2, dtype=np.uint32)
    ref.partition_by_nnz_rgrn_coo(ctypes.byref(m), _p(split), ctypes.c_int(nparts))
    return split[: nparts + 1]


def ref_partition_tsklt_by_nnz_coo(nnz, nparts):
    ref = ctypes.CDLL(_REF_PATH)
    split = np.zeros(nparts + 2, dtype=np.uint32)
    ref.partition_tsklt_by_nnz_coo(ctypes.c_uint32(nnz), _p(split), ctypes.c_int(nparts))
    return split[: nparts + 1]


# --------------------------------------------------------------------------- #
# oracle/_ref/libref_host_<variant>_<DTYPE>.so : the reference's OWN host loops #
# (spmm_host_coo, spmm_host_csr, grande's valued spmm_host_csr, spmv's           #
# spmm_host, the merge helpers and the group drivers), cut out of the reference #
# files by name at build time and compiled in place (oracle/build_ref_host.sh). #
# This is what pins the arithmetic of the oracle.                               #
# --------------------------------------------------------------------------- #
REF_DTYPE_NAME = {
    np.dtype(np.int8): "INT8", np.dtype(np.int16): "INT16", np.dtype(np.int32): "INT32",
    np.dtype(np.int64): "INT64", np.dtype(np.float32): "FLT32", np.dtype(np.float64): "DBL64",
}
REF_HOST_PINNED_BY = ("reference spmm_host_coo (spmm_default/spmm_mul_coo.c:40-51), spmm_host_csr (spmm_grande/spmm_mul_csr.c:119-136, "
                      "spmm_default/spmm_mul_csr.c:100-113) and spmm_host (spmv_sparseP/spmv_mul_coo.c:92-103) compiled in place "
                      "(oracle/build_ref_host.sh)")
_ref_host_libs = {}


def _ref_host_path(variant, dtype):
    return os.path.join(_HERE, "_ref", f"libref_host_{variant}_{REF_DTYPE_NAME[np.dtype(dtype)]}.so")


def have_ref_host():
    return all(os.path.exists(_ref_host_path(v, d)) for v in ("default", "grande", "spmv") for d in REF_DTYPE_NAME)


def ref_host(variant, dtype):
    key = (variant, np.dtype(dtype))
    if key not in _ref_host_libs:
        lib_ = ctypes.CDLL(_ref_host_path(variant, dtype))
        assert lib_.ref_sizeof_val_dt() == np.dtype(dtype).itemsize
        _ref_host_libs[key] = lib_
    return _ref_host_libs[key]


def _ones_like_nnz(values, n, dtype):
    return np.ones(n, dtype=dtype) if values is None else np.ascontiguousarray(values, dtype=dtype)


def ref_spmm_host_coo(row, col, values, x, nrows, variant="default"):
    """The reference's COO host loop: spmm_host_coo (default) or spmm_host (spmv).  Accumulates into a zeroed y."""
    x = np.ascontiguousarray(x)
    row, col = _u32(row), _u32(col)
    v = _ones_like_nnz(values, len(col), x.dtype)
    m = _RefCOO(nrows, x.shape[0], len(col), None, row.ctypes.data, col.ctypes.data, v.ctypes.data, len(col))
    y = np.zeros((nrows, x.shape[1]), dtype=x.dtype)
    f = getattr(ref_host(variant, x.dtype), "spmm_host_coo" if variant == "default" else "spmm_host")
    f(_p(y), ctypes.byref(m), _p(x), ctypes.c_uint32(x.shape[1]))
    return y


def ref_spmm_host_csr(rowptr, colind, values, x, variant="grande", ldx=None):
    """The reference's CSR host loop.  variant "grande": the valued loop with a padded X stride
    (spmm_grande/spmm_mul_csr.c:119-136); "default": the unit-weight loop that reads but ignores the
    values (spmm_default/spmm_mul_csr.c:100-113)."""
    x = np.ascontiguousarray(x)
    rowptr, colind = _u32(rowptr), _u32(colind)
    v = _ones_like_nnz(values, len(colind), x.dtype)
    nrows = len(rowptr) - 1
    m = _RefCSR(nrows, x.shape[0], len(colind), rowptr.ctypes.data, colind.ctypes.data, v.ctypes.data,
                len(rowptr), len(colind), len(colind))
    lib_ = ref_host(variant, x.dtype)
    if variant == "grande":
        ncols = x.shape[1] if ldx is None else ldx[0]
        stride = x.shape[1] if ldx is None else ldx[1]
        y = np.zeros((nrows, ncols), dtype=x.dtype)
        lib_.spmm_host_csr(_p(y), ctypes.byref(m), _p(x), ctypes.c_uint32(ncols), ctypes.c_uint32(stride))
    else:
        y = np.zeros((nrows, x.shape[1]), dtype=x.dtype)
        lib_.spmm_host_csr(_p(y), ctypes.byref(m), _p(x), ctypes.c_uint32(x.shape[1]))
    return y


def ref_group(is_coo, ptr_or_row, colind, values, nrows, ncols, x_parts, h, variant="default"):
    """The reference's group driver (spmm_host_csr_group / spmm_host_coo_group, spmm_default/ops.hpp:42-62,97-118;
    spmm_host_group, spmv_sparseP/spmv_mul_coo.c:128-148) on the reference's own structs; arguments as group()."""
    n = len(colind)
    dt = x_parts[0].dtype
    pr = [_u32(a) for a in ptr_or_row]
    ci = [_u32(a) for a in colind]
    vals = None if values is None else [np.ascontiguousarray(v, dtype=dt) for v in values]
    xs = [np.ascontiguousarray(x) for x in x_parts]
    VP = ctypes.c_void_p
    arr = lambda lst: (VP * len(lst))(*[a.ctypes.data for a in lst])
    nnz = np.array([len(c) for c in ci], dtype=np.uint32)
    nr = np.array(nrows, dtype=np.uint32)
    nc = np.array(ncols, dtype=np.uint32)
    dn = np.array([x.shape[1] for x in xs], dtype=np.uint32)
    out = np.zeros((int(nr[0]), h), dtype=dt)
    rc = ref_host(variant, dt).ref_group(
        _p(out), ctypes.c_int(1 if is_coo else 0), ctypes.c_uint32(n), arr(pr), arr(ci),
        None if vals is None else arr(vals), _p(nr), _p(nc), _p(nnz),
        ctypes.c_uint32(len(xs)), arr(xs), _p(dn), ctypes.c_uint32(h))
    assert rc == 0
    return out


def ref_merge(name, dest, src, off_x, off_y, len_x, len_y, variant="default"):
    """add_2D / memadd_2D / memcpy_2D of the reference (spmm_default/spmm_mul_csr.c:41-86,
    spmv_sparseP/spmv_mul_coo.c:54-115) on row-major 2-D arrays; dest is modified in place."""
    assert dest.flags.c_contiguous and src.flags.c_contiguous and dest.dtype == src.dtype
    getattr(ref_host(variant, dest.dtype), name)(
        _p(dest), _p(src), ctypes.c_uint32(dest.shape[1]), ctypes.c_uint32(src.shape[1]),
        ctypes.c_uint32(off_x), ctypes.c_uint32(off_y), ctypes.c_uint32(len_x), ctypes.c_uint32(len_y))
    return dest


def ref_matrix_add(a, b):
    """matrix_add, spmm_default/spmm_mul_csr.c:55-60."""
    getattr(ref_host("default", a.dtype), "matrix_add")(_p(a), _p(b), ctypes.c_uint32(a.shape[0]), ctypes.c_uint32(a.shape[1]))
    return a


def add_2d(dest, src, off_x, off_y, len_x, len_y):
    """The oracle's own restatement of add_2D (oracle_add_2d_<T>)."""
    getattr(lib(), "oracle_add_2d_" + SUFFIX[np.dtype(dest.dtype)])(
        _p(dest), _p(src), ctypes.c_uint32(dest.shape[1]), ctypes.c_uint32(src.shape[1]),
        ctypes.c_uint32(off_x), ctypes.c_uint32(off_y), ctypes.c_uint32(len_x), ctypes.c_uint32(len_y))
    return dest


_REF_UTILS_PATH = os.path.join(_HERE, "_ref", "libref_utils.so")


def have_ref_utils():
    return os.path.exists(_REF_UTILS_PATH)


def ref_read_matrix_csr(path):
    """The reference's own reader (spmm_default/utils.hpp:15-70 readCOOMatrix, :87-127 coo2csr,
    compiled in place by `make ref_utils` with INT32 values): returns (nrows, ncols, rowptr, colind,
    values) exactly as its read_matrix_* debug ops would see them."""
    import torch  # noqa: F401  (libref_utils.so links libtorch: load torch's copies first)
    ref = ctypes.CDLL(_REF_UTILS_PATH)
    rd = getattr(ref, "_Z13readCOOMatrixPKc")
    rd.restype = ctypes.POINTER(_RefCOO)
    cv = getattr(ref, "_Z7coo2csrP9COOMatrix")
    cv.restype = ctypes.POINTER(_RefCSR)
    cv.argtypes = [ctypes.POINTER(_RefCOO)]
    coo = rd(os.fsencode(path))
    csr = cv(coo).contents
    n, nnz = int(csr.nrows), int(csr.nnz)
    take = lambda ptr, cnt, dt: np.ctypeslib.as_array(
        ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint32)), shape=(cnt,)).view(dt).copy()
    return (n, int(csr.ncols), take(csr.rowptr, n + 1, np.uint32), take(csr.colind, max(nnz, 1), np.uint32)[:nnz],
            take(csr.values, max(nnz, 1), np.int32)[:nnz])


# --------------------------------------------------------------------------- #
# quantiser of the conv layers (reference models/quantize.py:20-42), numpy     #
# restatement: float32 arithmetic, round half to even, C cast to the type.     #
# "parity unpinned": quantize.py imports torch_sparse at module level and so   #
# cannot be imported here; the statement below follows it line by line.        #
# --------------------------------------------------------------------------- #
def symmetric_quantize(v, np_dtype):
    v = np.asarray(v, dtype=np.float32)
    abs_max = np.float32(np.max(np.abs(v)))
    bits = {np.dtype(np.int8): 5, np.dtype(np.int16): 10, np.dtype(np.int32): 20}.get(np.dtype(np_dtype), 20)
    scale = np.float32(np.float32(abs_max * np.float32(2)) / np.float32(2 ** bits))
    q = np.rint(v / scale)  # float32 division, half to even
    target = np_dtype if np.dtype(np_dtype) in (np.dtype(np.int8), np.dtype(np.int16), np.dtype(np.int32)) else np.float32
    return scale, q.astype(target)


def symmetric_dequantize(out_q, scale_edge, scale_x):
    return out_q.astype(np.float32) * np.float32(np.float32(scale_edge) * scale_x)
