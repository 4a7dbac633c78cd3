"""ctypes binding of the C-ABI HIP library (include/pygim_hip.h).

The library is built in-tree (``pygim_amd/libpygim_hip.so``, see csrc/Makefile or
``__graft_entry__.build``).  There is no fallback: if it is missing, importing the
ops fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PYGIM_HIP_LIB", os.path.join(_HERE, "libpygim_hip.so"))

# symbols include/pygim_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "pygim_init_ranks", "pygim_init_units", "pygim_release", "pygim_is_initialized", "pygim_last_error",
    "pygim_device_info", "pygim_group_create", "pygim_group_free", "pygim_spmm_run_group",
    "pygim_grande_run_group", "pygim_spmv_run_group", "pygim_block_run", "pygim_group_timers",
    "pygim_group_info", "pygim_set_tunable", "pygim_group_kernel_ms", "pygim_quant_spmm_run",
    "pygim_quant_absmax", "pygim_quantize", "pygim_dequantize", "pygim_spmm_run_group_x", "pygim_block_run_x",
    "pygim_group_kernel_events", "pygim_group_plan", "pygim_spmm_run_dequant",
    "pygim_quant_spmm_run_post", "pygim_generation", "pygim_group_lds_plan", "pygim_group_lds_code", "pygim_group_lds_geometry", "pygim_group_lds_note",
    "pygim_group_lds_tiles", "pygim_group_lds_runs", "pygim_group_serial", "pygim_group_host_windows",
]

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_UNSORTED = 0, 1, 2, 3, 4
INT8, INT16, INT32, INT64, FLT32, DBL64 = range(6)
CSR, COO = 0, 1


class PygimError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pygim_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the .so is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP extension first "
                f"(`make -C pygim_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`). "
                f"This backend has no CPU fallback.")
        # torch ships its own libamdhip64; load it FIRST so that this library binds to the same
        # HIP runtime (device pointers, streams and events are shared with torch tensors)
        import torch  # noqa: F401

        L = ctypes.CDLL(LIB_PATH)
        c_i64, c_int, vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
        p_i64 = ctypes.POINTER(ctypes.c_int64)
        L.pygim_last_error.restype = ctypes.c_char_p
        L.pygim_init_ranks.argtypes = [c_i64, p_i64]
        L.pygim_init_units.argtypes = [c_i64, p_i64, p_i64]
        L.pygim_device_info.argtypes = [ctypes.c_char_p, c_int, ctypes.POINTER(c_int), p_i64]
        L.pygim_group_create.argtypes = [c_int, c_int, c_int, vp, vp, vp, p_i64, p_i64, p_i64, p_i64, p_i64, c_i64,
                                         p_i64]
        L.pygim_group_free.argtypes = [c_i64]
        L.pygim_group_serial.argtypes = [c_i64, p_i64]
        L.pygim_spmm_run_group.argtypes = [c_i64, vp, vp, vp]
        L.pygim_spmm_run_group_x.argtypes = [c_i64, vp, vp, c_int, vp]
        L.pygim_grande_run_group.argtypes = [c_i64, vp, p_i64, vp, vp]
        L.pygim_spmv_run_group.argtypes = [c_i64, vp, vp, vp]
        L.pygim_block_run.argtypes = [c_i64, c_int, vp, c_i64, vp, c_i64, c_i64, c_int, vp]
        L.pygim_block_run_x.argtypes = [c_i64, c_int, vp, c_i64, vp, c_i64, c_i64, c_int, c_int, vp]
        L.pygim_group_timers.argtypes = [c_i64, ctypes.POINTER(ctypes.c_double)]
        L.pygim_group_info.argtypes = [c_i64, p_i64]
        L.pygim_group_kernel_ms.argtypes = [c_i64, ctypes.POINTER(ctypes.c_double), p_i64, c_int]
        L.pygim_group_kernel_events.argtypes = [c_i64, c_int]
        L.pygim_group_plan.argtypes = [c_i64, p_i64]
        L.pygim_group_lds_plan.argtypes = [c_i64, p_i64]
        L.pygim_group_lds_code.argtypes = [c_i64, p_i64]
        L.pygim_group_lds_geometry.argtypes = [c_i64, p_i64]
        L.pygim_group_lds_note.argtypes = [c_i64, ctypes.c_char_p, c_i64]
        L.pygim_group_lds_tiles.argtypes = [c_i64, p_i64]
        L.pygim_group_lds_runs.argtypes = [c_i64, p_i64]
        L.pygim_group_host_windows.argtypes = [c_i64, p_i64, p_i64]
        L.pygim_generation.restype = c_i64
        L.pygim_quant_spmm_run.argtypes = [c_i64, vp, c_i64, vp, vp, vp]
        L.pygim_quant_spmm_run_post.argtypes = [c_i64, vp, c_i64, vp, vp, vp, vp, c_int, vp]
        L.pygim_quant_absmax.argtypes = [vp, c_i64, c_i64, c_i64, vp, vp]
        L.pygim_quantize.argtypes = [c_int, vp, c_i64, c_i64, c_i64, vp, vp, vp, vp]
        L.pygim_dequantize.argtypes = [c_int, vp, c_i64, vp, vp, vp]
        L.pygim_spmm_run_dequant.argtypes = [c_i64, vp, c_i64, vp, vp, vp]
        L.pygim_set_tunable.argtypes = [ctypes.c_char_p, c_i64]
        L.pygim_set_tunable.restype = c_i64
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise PygimError(rc, lib().pygim_last_error().decode())


def i64_array(values):
    return (ctypes.c_int64 * len(values))(*[int(v) for v in values])


def ptr_array(ptrs):
    return (ctypes.c_void_p * len(ptrs))(*[ctypes.c_void_p(int(p) if p else None) for p in ptrs])


def _env_tunables():
    """PYGIM_TUNE="name=value,name=value": kernel tunables for a whole run without touching the driver scripts (A/B runs)"""
    for kv in filter(None, os.environ.get("PYGIM_TUNE", "").split(",")):
        name, _, value = kv.partition("=")
        if set_tunable(name.strip(), int(value)) == -1:
            raise ValueError(f"PYGIM_TUNE: unknown tunable {name!r}")


def init_ranks(nr_ranks, want_units=False):
    out = (ctypes.c_int64 * max(int(nr_ranks), 1))()
    check(lib().pygim_init_ranks(int(nr_ranks), out))
    _env_tunables()
    return list(out)[: int(nr_ranks)] if want_units else None


def init_units(nr_units):
    n = max((int(nr_units) + 7) // 8, 1)
    out = (ctypes.c_int64 * n)()
    ranks = ctypes.c_int64(0)
    check(lib().pygim_init_units(int(nr_units), out, ctypes.byref(ranks)))
    return list(out)[: ranks.value]


def release():
    check(lib().pygim_release())


def is_initialized():
    return bool(lib().pygim_is_initialized())


def device_info():
    name = ctypes.create_string_buffer(256)
    cu = ctypes.c_int(0)
    mem = ctypes.c_int64(0)
    check(lib().pygim_device_info(name, 256, ctypes.byref(cu), ctypes.byref(mem)))
    return name.value.decode(), cu.value, mem.value


def group_create(fmt, dtype, idx0_ptrs, col_ptrs, val_ptrs, nrows, ncols, nnz, n_dense, dense_cols, h):
    n = len(col_ptrs)
    handle = ctypes.c_int64(0)
    vals = None if val_ptrs is None else ptr_array(val_ptrs)
    check(lib().pygim_group_create(fmt, dtype, n, ptr_array(idx0_ptrs), ptr_array(col_ptrs), vals, i64_array(nrows),
                                   i64_array(ncols), i64_array(nnz), i64_array(n_dense), i64_array(dense_cols),
                                   int(h), ctypes.byref(handle)))
    return handle.value


def group_free(handle):
    check(lib().pygim_group_free(int(handle)))


def spmm_run_group(handle, b_ptrs, out_ptr, stream=0, x_unchanged=False):
    """``x_unchanged``: the dense parts hold exactly what the most recent product on the same pointers held
    (include/pygim_hip.h, "Threading"): its slice-major copy is reused instead of being made again."""
    check(lib().pygim_spmm_run_group_x(int(handle), ptr_array(b_ptrs), ctypes.c_void_p(out_ptr),
                                       1 if x_unchanged else 0, ctypes.c_void_p(stream or None)))


def grande_run_group(handle, b_ptrs, lds, out_ptr, stream=0):
    check(lib().pygim_grande_run_group(int(handle), ptr_array(b_ptrs), i64_array(lds), ctypes.c_void_p(out_ptr),
                                       ctypes.c_void_p(stream or None)))


def spmv_run_group(handle, b_ptrs, out_ptr, stream=0):
    check(lib().pygim_spmv_run_group(int(handle), ptr_array(b_ptrs), ctypes.c_void_p(out_ptr),
                                     ctypes.c_void_p(stream or None)))


def block_run(handle, part, x_ptr, ldx, c_ptr, ldc, width, accumulate=False, stream=0, x_unchanged=False):
    check(lib().pygim_block_run_x(int(handle), int(part), ctypes.c_void_p(x_ptr), int(ldx), ctypes.c_void_p(c_ptr),
                                  int(ldc), int(width), 1 if accumulate else 0, 1 if x_unchanged else 0,
                                  ctypes.c_void_p(stream or None)))


def group_timers(handle):
    out = (ctypes.c_double * 5)()
    check(lib().pygim_group_timers(int(handle), out))
    return list(out)


def group_kernel_events(handle, on=True):
    check(lib().pygim_group_kernel_events(int(handle), 1 if on else 0))


def group_plan(handle):
    out = (ctypes.c_int64 * 8)()
    check(lib().pygim_group_plan(int(handle), out))
    keys = ["n_panels", "panel_cols", "n_items", "col16", "n_coop_items", "n_segment_tasks", "merged", "has_extra"]
    return dict(zip(keys, [int(v) for v in out]))


def group_lds_plan(handle):
    out = (ctypes.c_int64 * 4)()
    check(lib().pygim_group_lds_plan(int(handle), out))
    return dict(zip(["tiles", "chunk_fills", "tokens", "nnz"], [int(v) for v in out]))


def group_lds_code(handle):
    out = (ctypes.c_int64 * 4)()
    check(lib().pygim_group_lds_code(int(handle), out))
    return dict(zip(["code_bytes", "paired_entries", "active", "device_generated"], [int(v) for v in out[:4]]))


def group_lds_tiles(handle):
    """which rows share a tile: similarity order (label propagation) or consecutive rows"""
    out = (ctypes.c_int64 * 4)()
    check(lib().pygim_group_lds_tiles(int(handle), out))
    return dict(zip(["similarity", "labels", "largest_label_rows", "sweep_locality"], [int(v) for v in out[:4]]))


def group_lds_runs(handle):
    """products (or blocks) the LDS-staged kernels served for this group so far"""
    out = ctypes.c_int64(0)
    check(lib().pygim_group_lds_runs(int(handle), ctypes.byref(out)))
    return int(out.value)


def group_host_windows(handle):
    """feature windows the last run with host operands moved as a pipeline (1 = upload, product, download one after the other)"""
    return group_host_call(handle)["windows"]


def group_host_call(handle):
    """... and whether their products stored straight into the caller's page-locked result (``direct``)"""
    w, d = ctypes.c_int64(0), ctypes.c_int64(0)
    check(lib().pygim_group_host_windows(int(handle), ctypes.byref(w), ctypes.byref(d)))
    return {"windows": int(w.value), "direct": int(d.value)}


def group_lds_geometry(handle):
    out = (ctypes.c_int64 * 8)()
    check(lib().pygim_group_lds_geometry(int(handle), out))
    return dict(zip(["waves", "acc_per_wave", "chunk_cols", "buffers", "group", "x_sets", "shared_entries", "col_splits"], [int(v) for v in out]))


def group_lds_note(handle):
    """which form of the product the group got (code stream / token kernels / sweep) and why"""
    buf = ctypes.create_string_buffer(1024)
    check(lib().pygim_group_lds_note(int(handle), buf, 1024))
    return buf.value.decode()


def generation():
    """count of pygim_release calls: handles of an earlier generation are dead"""
    return int(lib().pygim_generation())


def group_serial(handle):
    """creation serial of a live group (never reused, unlike the address that is the handle); PygimError for a dead handle"""
    out = ctypes.c_int64(0)
    check(lib().pygim_group_serial(int(handle), ctypes.byref(out)))
    return int(out.value)


def group_info(handle):
    out = (ctypes.c_int64 * 8)()
    check(lib().pygim_group_info(int(handle), out))
    keys = ["total_rows", "total_cols", "h", "n_parts", "n_long_rows", "all_ones", "n_panels", "n_items"]
    return dict(zip(keys, list(out)))


def set_tunable(name, value):
    return int(lib().pygim_set_tunable(name.encode(), int(value)))


def group_kernel_ms(handle, reset=True):
    """(sum of milliseconds, launches) of the dominant kernel since the last reset
    (needs set_tunable('kernel_events', 1))."""
    ms = ctypes.c_double(0)
    n = ctypes.c_int64(0)
    check(lib().pygim_group_kernel_ms(int(handle), ctypes.byref(ms), ctypes.byref(n), 1 if reset else 0))
    return ms.value, n.value


def quant_absmax(x_ptr, ldx, rows, width, bits_ptr, stream=0):
    check(lib().pygim_quant_absmax(ctypes.c_void_p(x_ptr or None), int(ldx), int(rows), int(width), ctypes.c_void_p(bits_ptr),
                                   ctypes.c_void_p(stream or None)))


def quantize(dtype, x_ptr, ldx, rows, width, bits_ptr, xq_ptr, scale_ptr=0, stream=0):
    check(lib().pygim_quantize(int(dtype), ctypes.c_void_p(x_ptr or None), int(ldx), int(rows), int(width),
                               ctypes.c_void_p(bits_ptr), ctypes.c_void_p(xq_ptr or None), ctypes.c_void_p(scale_ptr or None),
                               ctypes.c_void_p(stream or None)))


def dequantize(dtype, q_ptr, n, bits_ptr, out_ptr, stream=0):
    check(lib().pygim_dequantize(int(dtype), ctypes.c_void_p(q_ptr or None), int(n), ctypes.c_void_p(bits_ptr),
                                 ctypes.c_void_p(out_ptr or None), ctypes.c_void_p(stream or None)))


def spmm_run_dequant(handle, xq_ptr, ldx, out_ptr, bits_ptr, stream=0):
    check(lib().pygim_spmm_run_dequant(int(handle), ctypes.c_void_p(xq_ptr), int(ldx), ctypes.c_void_p(out_ptr),
                                       ctypes.c_void_p(bits_ptr), ctypes.c_void_p(stream or None)))


def quant_spmm_run(handle, x_ptr, ldx, out_ptr, scale_ptr=0, stream=0, col_mul_ptr=0, col_add_ptr=0, relu=False):
    """quantise -> aggregate -> dequantise; with ``col_mul_ptr`` / ``col_add_ptr`` (device float[h]) the per-column
    epilogue out = col_mul * out + col_add (then ReLU) runs in the sweep's last store"""
    check(lib().pygim_quant_spmm_run_post(int(handle), ctypes.c_void_p(x_ptr), int(ldx), ctypes.c_void_p(out_ptr),
                                          ctypes.c_void_p(scale_ptr or None), ctypes.c_void_p(col_mul_ptr or None),
                                          ctypes.c_void_p(col_add_ptr or None), 1 if relu else 0,
                                          ctypes.c_void_p(stream or None)))
