"""A test double for pygim_amd._lib: the same Python-level functions, but the work is done by
the CPU oracle on the HOST memory the pointers refer to.  Lets the CPU suite drive the whole
Python surface (pim_ops registration, backend_pim wrappers, pointer marshalling) without a GPU.
Test infrastructure only."""
import ctypes

import numpy as np

import oracle

INT8, INT16, INT32, INT64, FLT32, DBL64 = range(6)
CSR, COO = 0, 1
NP_OF = {INT8: np.int8, INT16: np.int16, INT32: np.int32, INT64: np.int64, FLT32: np.float32, DBL64: np.float64}


class PygimError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def _view(ptr, n, npdt):
    if n == 0:
        return np.zeros(0, dtype=npdt)
    buf = (ctypes.c_char * (n * np.dtype(npdt).itemsize)).from_address(int(ptr))
    return np.frombuffer(buf, dtype=npdt, count=n)


class FakeLib:
    INT8, INT16, INT32, INT64, FLT32, DBL64 = range(6)
    CSR, COO = 0, 1
    PygimError = PygimError

    def __init__(self):
        self.groups, self.inited, self.next = {}, False, 1000

    def lib(self):
        return self

    def init_ranks(self, n, want_units=False):
        self.inited = True
        return [8] * int(n) if want_units else None

    def init_units(self, n):
        self.inited = True
        return [8] * ((int(n) + 7) // 8)

    def release(self):
        self.groups.clear()
        self.inited = False

    def is_initialized(self):
        return self.inited

    def group_create(self, fmt, dtype, idx0, cols, vals, nrows, ncols, nnz, n_dense, dense_cols, h):
        if not self.inited:
            raise PygimError(2, "backend not initialised")
        npdt = NP_OF[dtype]
        parts = []
        for i in range(len(cols)):
            n0 = nrows[i] + 1 if fmt == CSR else nnz[i]
            parts.append(dict(idx0=_view(idx0[i], n0, np.int32).copy(), col=_view(cols[i], nnz[i], np.int32).copy(),
                              val=None if vals is None else _view(vals[i], nnz[i], npdt).copy(),
                              nrows=int(nrows[i]), ncols=int(ncols[i])))
        self.next += 1
        self.groups[self.next] = dict(fmt=fmt, dt=npdt, parts=parts, n_dense=list(n_dense), dense_cols=list(dense_cols), h=int(h))
        return self.next

    def group_free(self, handle):
        self.groups.pop(int(handle))

    def _product(self, g, part, x):
        if g["fmt"] == CSR:
            return oracle.spmm_csr(part["idx0"], part["col"], part["val"], x)
        return oracle.spmm_coo(part["idx0"], part["col"], part["val"], x, part["nrows"])

    def block_run(self, handle, part, x_ptr, ldx, c_ptr, ldc, width, accumulate=False, stream=0, x_unchanged=False):
        """pygim_block_run: C[:, :width] (+)= A_part . X[:, :width] on strided windows (row strides ldx / ldc in elements)"""
        g = self.groups[int(handle)]
        p = g["parts"][int(part)]
        es = np.dtype(g["dt"]).itemsize

        def window(ptr, rows, ld):
            if rows == 0:
                return np.zeros((0, width), dtype=g["dt"])
            flat = _view(ptr, (rows - 1) * ld + width, g["dt"])
            return np.lib.stride_tricks.as_strided(flat, shape=(rows, width), strides=(ld * es, es), writeable=True)

        xw = np.ascontiguousarray(window(x_ptr, p["ncols"], int(ldx)))
        cw = window(c_ptr, p["nrows"], int(ldc))
        prod = self._product(g, p, xw)
        if accumulate:
            cw += prod
        else:
            cw[:] = prod

    def group_kernel_events(self, handle, on=True):
        pass

    def spmm_run_group(self, handle, b_ptrs, out_ptr, stream=0, x_unchanged=False):
        g = self.groups[int(handle)]
        total_cols = sum(p["ncols"] for p in g["parts"])
        widths = g["dense_cols"][: g["n_dense"][0]]
        xs = [_view(b_ptrs[j], total_cols * w, g["dt"]).reshape(total_cols, w) for j, w in enumerate(widths)]
        out = _view(out_ptr, g["parts"][0]["nrows"] * g["h"], g["dt"]).reshape(-1, g["h"])
        out[:] = 0
        brow = 0
        for p in g["parts"]:
            acol = 0
            for j, w in enumerate(widths):
                out[:, acol:acol + w] += self._product(g, p, xs[j][brow:brow + p["ncols"]])
                acol += w
            brow += p["ncols"]

    def grande_run_group(self, handle, b_ptrs, lds, out_ptr, stream=0):
        g = self.groups[int(handle)]
        out = _view(out_ptr, g["parts"][0]["nrows"] * g["h"], g["dt"]).reshape(-1, g["h"])
        out[:] = 0
        k = 0
        for i, p in enumerate(g["parts"]):
            acol = 0
            for j in range(g["n_dense"][i]):
                w, ld = g["dense_cols"][k], lds[k]
                xw = _view(b_ptrs[k], p["ncols"] * ld, g["dt"]).reshape(p["ncols"], ld)[:, :w]
                out[:, acol:acol + w] += self._product(g, p, np.ascontiguousarray(xw))
                acol += w
                k += 1

    def spmv_run_group(self, handle, b_ptrs, out_ptr, stream=0):
        g = self.groups[int(handle)]
        nvec = g["n_dense"][0]
        total_cols = sum(p["ncols"] for p in g["parts"])
        x = np.stack([_view(b_ptrs[j], total_cols, g["dt"]) for j in range(nvec)], axis=1)
        out = _view(out_ptr, g["parts"][0]["nrows"] * nvec, g["dt"]).reshape(-1, nvec)
        out[:] = 0
        brow = 0
        for p in g["parts"]:
            out += self._product(g, p, np.ascontiguousarray(x[brow:brow + p["ncols"]]))
            brow += p["ncols"]
