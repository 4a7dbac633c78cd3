#!/usr/bin/env python3
"""INT8 dense-cell MFMA prototype (round 6): decide with numbers.

For a Reddit-shaped graph WITH locality (columns clustered near the row id; a stochastic block model with community-contiguous ids -- what the
library's label propagation recovers for shuffled ids), INT8 features, h = 256 -- the conv layers' own inference type (models/quantize.py:22-23,
pyg_gcn_conv.py:130-137):
    before : the library's product on the whole adjacency (k_lds_code8_i8: the INT16 code stream on widened features)
    after  : 32 x 32 cells with >= T stored entries expanded to int8 panels and contracted with v_mfma_i32_32x32x32_i8 (scripts/micro/mfma_cells.hip),
             every other entry through the library as before, the two halves added modulo 2^8
Both are checked against each other element by element (and the library's product against the oracle on sampled rows).
usage: exp_mfma_cells.py [graphs: clustered,sbm,uniform] [thresholds: 26,52,103] [tn: 8|4|2]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import oracle
from pygim_amd import _lib, synth

_SO = os.path.join(ROOT, "scripts", "micro", "libmfma_cells.so")
if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SO.replace("libmfma_cells.so", "mfma_cells.hip")):   # (a prototype: not part of the library's build)
    import subprocess

    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", _SO.replace("libmfma_cells.so", "mfma_cells.hip"), "-o", _SO])
L = ctypes.CDLL(_SO)
vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
L.mfma_cells_run.argtypes = [vp, vp, vp, vp, vp, ci, ci, i64, ci, vp]
L.mfma_cells_lds_run.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, i64, vp]
L.mfma_cells_deep_run.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, i64, ci, vp]
L.mfma_pack_xt.argtypes = [vp, i64, ci, ci, vp, ci, vp]
L.mfma_combine.argtypes = [vp, vp, vp, i64, vp]

dev = torch.device("cuda", 0)
_lib.init_ranks(1)
graphs = (sys.argv[1] if len(sys.argv) > 1 else "clustered,sbm").split(",")
thresholds = [int(t) for t in (sys.argv[2] if len(sys.argv) > 2 else "26,52,103").split(",")]
TN = int(sys.argv[3]) if len(sys.argv) > 3 else 4
h = 256
n, nnz, dmax = synth.SHAPES["reddit"]


def med(fn, reps=7, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for i in range(reps):
        ev[i].record(); fn()
    ev[reps].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2]


def lib_product(rowptr, col, x, nrows):
    m = col.numel()
    hd = _lib.group_create(_lib.CSR, _lib.INT8, [rowptr.data_ptr()], [col.data_ptr()], None, [nrows], [n], [m], [1], [h], h)
    out = torch.empty((nrows, h), dtype=torch.int8, device=dev)
    t = med(lambda: _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0))
    note, runs = _lib.group_lds_note(hd), _lib.group_lds_runs(hd)
    _lib.group_free(hd)
    return out, t, note[:70], runs


for gname in graphs:
    if gname == "clustered":
        rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev, clustered=True)
    elif gname == "sbm":
        rowptr, col = synth.make_shape("reddit", seed=0, device=dev, kind="sbm", shuffle=False)
    else:
        rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
    x = synth.features(n, h, torch.int8, seed=1, device=dev)
    full, t_full, note_full, runs_full = lib_product(rowptr, col, x, n)
    # the library's product against the oracle on sampled rows
    rp = rowptr.to(torch.int64)
    xh = x.cpu().numpy()
    for r0, r1 in ((0, 40), (n - 40, n)):
        lo, hi = int(rp[r0]), int(rp[r1])
        ref = oracle.spmm_csr((rp[r0:r1 + 1] - lo).cpu().numpy().astype(np.int32), col[lo:hi].cpu().numpy(), None, xh)
        assert np.array_equal(full[r0:r1].cpu().numpy(), ref)
    print(f"== {gname}: Reddit-shaped, INT8, h = {h}; library on the whole adjacency: {t_full:6.3f} ms  [{note_full}] lds_runs {runs_full}", flush=True)
    deg = (rowptr[1:] - rowptr[:-1]).long()
    row = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    nbc = (n + 31) // 32
    nrb = (n + 31) // 32
    key = (row >> 5) * nbc + (col.long() >> 5)
    ukey, inv, cnt = torch.unique(key, return_inverse=True, return_counts=True)
    del key
    hist = [(t, float(cnt[cnt >= t].sum()) / nnz, int((cnt >= t).sum())) for t in (13, 26, 52, 103, 205)]
    print("   32 x 32 cells: " + "; ".join(f">= {t} entries ({t / 10.24:.1f} %): {f * 100:5.1f} % of the entries in {b} cells" for t, f, b in hist), flush=True)
    # packed X in MFMA B-operand order (per product, like the library's slice pack)
    h_tiles = h // 32
    xt = torch.empty((nbc * h_tiles * 64 * 16,), dtype=torch.int8, device=dev)
    t_pack = med(lambda: L.mfma_pack_xt(x.data_ptr(), h, n, h_tiles, xt.data_ptr(), nbc, None))
    for T in thresholds:
        dense_blk = cnt >= T
        nblk = int(dense_blk.sum())
        if nblk == 0:
            print(f"   T = {T}: no dense cells")
            continue
        dense_entry = dense_blk[inv]
        nd = int(dense_entry.sum())
        keep = ~dense_entry
        rp_s = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        rp_s[1:] = torch.cumsum(torch.bincount(row[keep], minlength=n), 0)
        rp_s32, col_s = rp_s.to(torch.int32), col[keep].contiguous()
        p_of_blk = torch.cumsum(dense_blk.long(), 0) - 1
        p = p_of_blk[inv[dense_entry]]
        r = row[dense_entry] & 31
        k = col[dense_entry].long() & 31
        idx = p * 1024 + (k >> 4) * 512 + r * 16 + (k & 15)
        a32 = torch.zeros(nblk * 1024, dtype=torch.int32, device=dev)
        a32.index_add_(0, idx, torch.ones_like(idx, dtype=torch.int32))
        assert int(a32.max()) <= 127
        a8 = a32.to(torch.int8)
        del a32, idx, p, r, k
        dk = ukey[dense_blk]
        rb, cb = dk // nbc, (dk % nbc).to(torch.int32).contiguous()
        rb_ptr = torch.zeros(nrb + 1, dtype=torch.int64, device=dev)
        rb_ptr[1:] = torch.cumsum(torch.bincount(rb, minlength=nrb), 0)
        rb_ptr = rb_ptr.to(torch.int32)
        cd = torch.zeros((nrb * 32, h), dtype=torch.int32, device=dev)
        run = lambda: L.mfma_cells_run(rb_ptr.data_ptr(), cb.data_ptr(), a8.data_ptr(), xt.data_ptr(), cd.data_ptr(), nrb, h_tiles, h, TN, None)
        assert run() == 0
        t_mfma = med(run)
        cd_first = cd.clone()
        # second form: 8 row blocks per workgroup, the column block's tile of X shared through LDS
        srb = rb >> 3
        pkey = srb * nbc + cb.long()
        upair, pinv = torch.unique(pkey, return_inverse=True)          # sorted: by super row block, then column block
        npairs = upair.numel()
        pair_cell = torch.full((npairs * 8,), -1, dtype=torch.int32, device=dev)
        pair_cell[pinv * 8 + (rb & 7)] = torch.arange(nblk, dtype=torch.int32, device=dev)
        pair_cb = (upair % nbc).to(torch.int32).contiguous()
        nsrb = (nrb + 7) // 8
        srb_ptr = torch.zeros(nsrb + 1, dtype=torch.int64, device=dev)
        srb_ptr[1:] = torch.cumsum(torch.bincount(upair // nbc, minlength=nsrb), 0)
        srb_ptr = srb_ptr.to(torch.int32)
        cd.zero_()
        run2 = lambda: L.mfma_cells_lds_run(srb_ptr.data_ptr(), pair_cb.data_ptr(), pair_cell.data_ptr(), a8.data_ptr(), xt.data_ptr(), cd.data_ptr(), nsrb, nrb, h_tiles, h, None)
        assert run2() == 0
        torch.cuda.synchronize()
        same2 = torch.equal(cd, cd_first)
        t_mfma2 = med(run2)
        del cd_first
        print(f"   T = {T:3d}: LDS-shared form: {npairs} (256-row block, column block) pairs, {nblk / npairs:4.2f} cells per pair: mfma {t_mfma2:6.3f} ms "
              f"({t_mfma2 * 1e-3 * 2.4e9 * 256 / nblk:6.1f} CU-cycles per cell, {2 * nblk * 32 * 32 * h / (t_mfma2 * 1e-3) / 1e12:6.1f} dense TOP/s) {'== first form' if same2 else 'MISMATCH'}", flush=True)
        for depth in (4, 6, 8):
            cd2 = torch.zeros_like(cd)
            run3 = lambda: L.mfma_cells_deep_run(srb_ptr.data_ptr(), pair_cb.data_ptr(), pair_cell.data_ptr(), a8.data_ptr(), xt.data_ptr(), cd2.data_ptr(), nsrb, nrb, h_tiles, h, depth, None)
            assert run3() == 0
            torch.cuda.synchronize()
            same3 = torch.equal(cd2, cd)
            t3 = med(run3)
            print(f"   T = {T:3d}: ... with the next {depth:2d} pairs' operands in flight: mfma {t3:6.3f} ms ({t3 * 1e-3 * 2.4e9 * 256 / nblk:6.1f} CU-cycles per cell, "
                  f"{2 * nblk * 32 * 32 * h / (t3 * 1e-3) / 1e12:6.1f} dense TOP/s) {'== first form' if same3 else 'MISMATCH'}", flush=True)
            t_mfma2 = min(t_mfma2, t3) if same3 else t_mfma2
            del cd2
        if col_s.numel() > 0:
            sp, t_sp, note_sp, runs_sp = lib_product(rp_s32, col_s, x, n)
        else:
            sp, t_sp, note_sp, runs_sp = torch.zeros((n, h), dtype=torch.int8, device=dev), 0.0, "-", 0
        out8 = torch.empty((n, h), dtype=torch.int8, device=dev)
        t_comb = med(lambda: L.mfma_combine(sp.data_ptr(), cd.data_ptr(), out8.data_ptr(), n * h, None))
        ok = torch.equal(out8, full)
        t_mfma_first = t_mfma
        t_mfma = min(t_mfma, t_mfma2)
        after = t_sp + t_pack + t_mfma + t_comb
        per_cell_cycles = t_mfma * 1e-3 * 2.4e9 * 256 / nblk
        print(f"   T = {T:3d} ({T / 10.24:4.1f} %): {nblk:8d} dense cells hold {nd / nnz * 100:5.1f} % of the entries ({nd / nblk:6.1f} per cell) | mfma {t_mfma:6.3f} ms (first form {t_mfma_first:6.3f}: "
              f"{per_cell_cycles:6.1f} CU-cycles per cell) + pack {t_pack:5.3f} + rest through the library {t_sp:6.3f} "
              f"[{note_sp[:40]}; lds_runs {runs_sp}] + combine {t_comb:5.3f} = {after:6.3f} ms  vs {t_full:6.3f} ms before: x{t_full / after:4.2f}   "
              f"{'equal to the library product, element by element' if ok else 'MISMATCH'}", flush=True)
        del a8, cd, sp, out8, col_s, rp_s, rp_s32
    del row, inv, ukey, cnt, xt, x, full
    torch.cuda.empty_cache()
_lib.release()
