#!/usr/bin/env python3
"""Timing table over BASELINE-style configurations (one GPU, one process): shape x format x dtype x h.
Checks every result with the column-count checksum (exact for the integer-valued driver features)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pygim_amd import _lib, synth  # noqa: E402

TD = {"i8": (torch.int8, _lib.INT8), "i16": (torch.int16, _lib.INT16), "i32": (torch.int32, _lib.INT32),
      "i64": (torch.int64, _lib.INT64), "f32": (torch.float32, _lib.FLT32), "f64": (torch.float64, _lib.DBL64)}


def bench(hd, xs, out, run, steps=5, warm=2):
    for _ in range(warm):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(steps):
        a.record()
        run()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="reddit:CSR:f32:256,reddit:COO:i32:256,reddit:CSR:i8:256,reddit:CSR:f64:256,"
                                       "reddit:CSR:i32:100,ogbn-products:COO:i32:256,ogbn-products:CSR:f32:256")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    _lib.init_ranks(1)
    cache = {}
    for case in args.cases.split(","):
        shape, fmt, dt, h = case.split(":")
        h = int(h)
        tdt, code = TD[dt]
        n, nnz, dmax = synth.SHAPES[shape]
        if shape not in cache:
            cache.clear()
            cache[shape] = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
        rowptr, col = cache[shape]
        x = synth.features(n, h, tdt, seed=0, device=dev)
        out = torch.empty((n, h), dtype=tdt, device=dev)
        if fmt == "CSR":
            keep = (rowptr, col)
            hd = _lib.group_create(_lib.CSR, code, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
            wsum = torch.bincount(col.long(), minlength=n).double()
            m = nnz
        else:
            row, ccol, val = synth.csr_to_coo_coalesced(rowptr, col, tdt)
            keep = (row, ccol, val)
            m = row.numel()
            hd = _lib.group_create(_lib.COO, code, [row.data_ptr()], [ccol.data_ptr()], [val.data_ptr()], [n], [n], [m], [1], [h], h)
            wsum = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, ccol.long(), val.double())
        st = torch.cuda.current_stream().cuda_stream
        run = lambda: _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), st)
        t = bench(hd, x, out, run)
        info = _lib.group_info(hd)
        ok = "n/a"
        if dt not in ("i8", "i16"):  # narrow ints wrap: checksum below is for non-wrapping sums
            ok = bool(torch.equal(out.double().sum(0), wsum @ x.double()))
        es = x.element_size()
        print(f"{case:34s} {t:9.3f} ms  {2.0 * m * h / t / 1e9:7.2f} Tops/s  gather {m * h * es / t / 1e9:6.2f} TB/s  "
              f"alg {synth.algorithmic_bytes(n, n, m, h, es, fmt) / t / 1e6:7.1f} GB/s  panels {info['n_panels']}  checksum_ok {ok}",
              flush=True)
        _lib.group_free(hd)
        del keep, x, out
    _lib.release()


if __name__ == "__main__":
    main()
