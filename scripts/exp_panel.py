#!/usr/bin/env python3
"""A/B sweep on one GPU, one process: row-per-wave kernel vs the L2-blocked panel sweep at
several L2 budgets, on the Reddit-shaped graph (and optionally others).  Prints one line per
variant; used to pick defaults (results go to DESIGN.md / profiles/)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pygim_amd import _lib, synth  # noqa: E402


def timeit(handle, x, out, steps=5, warm=2):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(warm):
        _lib.spmm_run_group(handle, [x.data_ptr()], out.data_ptr(), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(steps):
        a.record()
        _lib.spmm_run_group(handle, [x.data_ptr()], out.data_ptr(), st)
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts), sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="reddit")
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--clustered", action="store_true")
    ap.add_argument("--budgets", default="1,1.5,2,3,4,6")
    ap.add_argument("--long-ab", action="store_true")
    ap.add_argument("--ncols", type=int, default=0, help="restrict column ids to [0, ncols) (cache-residency probe)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    tdt, code = {"f32": (torch.float32, _lib.FLT32), "i32": (torch.int32, _lib.INT32),
                 "i8": (torch.int8, _lib.INT8), "f64": (torch.float64, _lib.DBL64)}[args.dtype]
    n, nnz, d_max = synth.SHAPES[args.shape]
    h = args.hidden
    ncols = args.ncols or n
    rowptr, col = synth.make_csr(n, nnz, d_max, seed=0, device=dev, clustered=args.clustered, ncols=ncols)
    x = synth.features(ncols, h, tdt, seed=0, device=dev)
    out = torch.empty((n, h), dtype=tdt, device=dev)
    _lib.init_ranks(1)
    flops = synth.flops(nnz, h)
    gb = synth.gather_bytes(n, nnz, h, x.element_size())

    def variant(name, mode, budget_mb=2.0, **tun):
        _lib.set_tunable("panel_mode", mode)
        _lib.set_tunable("panel_bytes", int(budget_mb * (1 << 20)))
        for k, v in tun.items():
            _lib.set_tunable(k, v)
        t0 = time.perf_counter()
        hd = _lib.group_create(_lib.CSR, code, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [ncols], [nnz], [1], [h], h)
        tc = time.perf_counter() - t0
        best, med = timeit(hd, x, out)
        ref_sum = out.to(torch.float64).sum().item()
        _lib.group_free(hd)
        print(f"{name:34s} best {best:8.3f} ms  median {med:8.3f} ms  {flops / best / 1e9:8.2f} TFLOP/s  "
              f"gather-model {gb / best / 1e9:7.2f} TB/s  create {tc:.2f}s  checksum {ref_sum:.6e}", flush=True)

    variant("row-per-wave (panel off)", 2)
    if args.long_ab:
        variant("panel 4 MiB, no long-row split", 1, 4, long_row_threshold=1 << 30)
        variant("panel 4 MiB, long rows > 2048 split", 1, 4, long_row_threshold=2048)
        _lib.set_tunable("long_row_threshold", 4096)
    for b in [float(v) for v in args.budgets.split(",")]:
        variant(f"panel {b} MiB, row-major X", 1, b, panel_pack=0)
        variant(f"panel {b} MiB, slice-major X", 1, b, panel_pack=1)
    _lib.release()


if __name__ == "__main__":
    main()
