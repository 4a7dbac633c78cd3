"""BASELINE.json's other configurations on the driver-timed line: measured by ``bench.py`` OUTSIDE its timed region,
each with its own ``roofline`` object and check, mirroring the reference driver's measurement (spmm_test.py:29-35: one ``mul`` per timed
step on resident operands, ``[DATA]pim_time_spmm(ms)``):

    configs[2]  ogbn-products-shaped COO, INT32, h = 256 -- uniform columns, coalesced multigraph
    configs[3]  Reddit-shaped GCN, 3 layers, h = 256, FLT32 -- one GPU's forward pass (the 8-GPU row split is the driver's SCALE run)
    configs[4]  ogbn-papers100M-shaped CSR, FLT32, h = 128 over 8 GPUs -- the work of ONE GPU: ds_parts = 8 (all rows x 16 features) and the
                2 x 4 grid share (half the rows x 32 features) the chooser prefers
    end to end  the reference driver's DEFAULT call (--device cpu): CPU tensors in, CPU tensor out, per ``mul``

Algorithmic bytes are SURVEY.md 8(d)'s compulsory traffic (pygim_amd.synth.algorithmic_bytes) for what the launch really holds (the
coalesced entry count, this GPU's rows and features); ``line_bytes`` is what the gathers must move at 128 bytes a line -- the gap between
the two is why a far-gather shape sits at a few percent of the HBM roofline while its lines run at most of the HBM peak.
"""
from __future__ import annotations

import json
import os
import time

import numpy as np
import torch

from . import _lib, synth

HBM_PEAK_GBS = 8000.0


def _median_ms(fn, steps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    for i in range(steps):
        ev[i].record()
        fn()
    ev[steps].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    return ts[len(ts) // 2]


def _kernel_name(hd, type_name):
    lp, lc = _lib.group_lds_plan(hd), _lib.group_lds_code(hd)
    if lp["tiles"] > 0 and lc["active"]:
        g = _lib.group_lds_geometry(hd)
        return f"k_lds_code8_{type_name}" if g["waves"] == 8 else f"k_lds_code_{type_name}"
    if lp["tiles"] > 0:
        return f"k_lds_spmm_{type_name} (token form)"
    plan = _lib.group_plan(hd)
    return f"k_csr_panel<{type_name}> x {max(int(plan['n_panels']), 1)} panel(s) (L2 / far-gather sweep)"


CFG_TRAFFIC_JSON = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "cfg_traffic_latest.json")


def _replayed_traffic(key, k_ms):
    """HBM bytes per product of a configuration's dominant kernel from the committed counter run (scripts/cfg_pmc.sh: 2 x FETCH_SIZE + WRITE_SIZE as
    MI355X_MICROARCH.md prescribes) -- replayed, and labelled so, only when this run's kernel time is within 10 % of the counter run's"""
    try:
        rec = json.load(open(CFG_TRAFFIC_JSON)).get(key)
    except Exception:  # noqa: BLE001
        return None, None
    if not rec:
        return None, None
    src = {"replayed_from": "profiles/cfg_traffic_latest.json", "collected": rec.get("collected"), "command": rec.get("command"), "kernel_ms_then": rec.get("kernel_ms_per_product"),
           "l2_hit_rate": rec.get("l2_hit_rate")}
    then = rec.get("kernel_ms_per_product")
    if not then or not k_ms or abs(k_ms - then) / then > 0.10:
        src["not_replayed"] = f"the counter run's kernel(s) took {then} ms per product, this run's {k_ms:.3f} ms: more than 10 % apart"
        return None, src
    return rec.get("hbm_bytes_per_product"), src


def _roofline(alg_bytes, k_ms, kernel, line_bytes, launches_note=None, traffic_key=None):
    ach = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic, traffic_src = _replayed_traffic(traffic_key, k_ms) if traffic_key else (None, None)
    out = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
           "kernel": kernel, "kernel_ms": round(k_ms, 4), "algorithmic_bytes": int(alg_bytes)}
    if traffic_src:
        out["traffic_source"] = traffic_src
    if traffic:   # what the memory system really moved, against the HBM peak: the number that says whether the kernel or the formulation is the limit
        out["traffic_GBs"] = round(traffic / (k_ms * 1e-3) / 1e9, 1)
        out["traffic_frac_of_hbm_peak"] = round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if line_bytes is not None:   # far-gather shapes: what the gathers must move at a cache line per stored entry, against the HBM peak
        out.update({"line_bytes": int(line_bytes), "line_GBs": round(line_bytes / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else None,
                    "line_frac_of_hbm_peak": round(line_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_ms > 0 else None})
    if launches_note:
        out["note"] = launches_note
    return out


def _kernel_ms(hd, fn, reps=3):
    """the library's own HIP events around the dominant kernel(s) of a product, on its launch stream"""
    _lib.group_kernel_events(hd, True)
    _lib.group_kernel_ms(hd, reset=True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms, cnt = _lib.group_kernel_ms(hd, reset=True)
    _lib.group_kernel_events(hd, False)
    return ms / max(cnt, 1)


def _sampled_rows_equal(rowptr, col, vals, x, out, picks, oracle):
    rp = rowptr.to(torch.int64)
    for r0, r1 in picks:
        lo, hi = int(rp[r0]), int(rp[r1])
        sub_rp = (rp[r0:r1 + 1] - lo).cpu().numpy().astype(np.int32)
        sub_col = col[lo:hi].cpu().numpy()
        used = np.unique(sub_col)   # only the rows of X these entries touch travel to the host
        xs = x[torch.from_numpy(used).to(x.device).long()].cpu().numpy()
        remap = np.searchsorted(used, sub_col).astype(np.int32)
        sub_val = None if vals is None else vals[lo:hi].cpu().numpy()
        ref = oracle.spmm_csr(sub_rp, remap, sub_val, xs)
        if not np.array_equal(out[r0:r1].cpu().numpy(), ref):
            return False
    return True


def config3_products_coo(dev, stream, h=256, steps=6):
    """configs[2] as SURVEY.md 8(d) defines it: products-shaped, UNIFORM columns, row-sorted coalesced COO (duplicates of the multigraph
    become values > 1, backend_pim/spmm.py:40-42), INT32, bit-exact"""
    import oracle

    n, nnz, d_max = synth.SHAPES["ogbn-products"]
    rowptr, col = synth.make_csr(n, nnz, d_max, seed=0, device=dev)
    row, ccol, val = synth.csr_to_coo_coalesced(rowptr, col, torch.int32)
    del col
    m = row.numel()
    x = synth.features(n, h, torch.int32, seed=0, device=dev)
    out = torch.empty((n, h), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hd = _lib.group_create(_lib.COO, _lib.INT32, [row.data_ptr()], [ccol.data_ptr()], [val.data_ptr()], [n], [n], [m], [1], [h], h)
    torch.cuda.synchronize()
    create_ms = (time.perf_counter() - t0) * 1e3
    try:
        fn = lambda: _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), stream)
        ms = _median_ms(fn, steps)
        k_ms = _kernel_ms(hd, fn)
        w = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, ccol.long(), val.double())
        ok = torch.equal(out.double().sum(0), w @ x.double())
        rp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        rp[1:] = torch.cumsum(torch.bincount(row.long(), minlength=n), 0)
        ok_rows = _sampled_rows_equal(rp, ccol, val, x, out, [(0, 300), (n // 2, n // 2 + 200), (n - 300, n)], oracle)
        alg = synth.algorithmic_bytes(n, n, m, h, 4, "COO", with_values=True)
        return {"workload": "ogbn-products-shaped COO SpMM (configs[2]): uniform columns, coalesced, INT32", "N": n, "nnz_coalesced": m, "nnz_multigraph": nnz, "h": h, "dtype": "i32",
                "ms_per_step": round(ms, 4), "value_GOPs": round(2 * m * h / (ms * 1e-3) / 1e9, 1), "group_create_ms": round(create_ms, 1),
                "roofline": _roofline(alg, k_ms, _kernel_name(hd, "i32"), m * max(h * 4, 128) + 2 * n * h * 4,
                                      "uniform columns over a 2.5 GB operand: every stored entry pulls its own 1 KiB row of X (8 lines) from HBM / Infinity Cache; "
                                      "line_bytes = entries x row bytes + C read and written", traffic_key="c3"),
                "check": ("bit-exact: weighted column-count checksum over all rows + 800 sampled rows against the oracle's COO loop" if ok and ok_rows else "MISMATCH"),
                "lds_note": _lib.group_lds_note(hd)[:160]}
    finally:
        _lib.group_free(hd)


def config5_papers_slices(dev, stream, steps=3, only=None):
    """configs[4]'s per-GPU work at full size (111 M rows, 1.6 G entries): the ds_parts = 8 slice (all rows x 16 of 128 features) and the 2 x 4
    grid share (the first nnz-balanced half of the rows x 32 features)"""
    import oracle
    from .bench_plans import nnz_balanced_row_split

    n, nnz, d_max = synth.SHAPES["ogbn-papers100M"]
    rowptr, col = synth.make_csr(n, nnz, d_max, seed=0, device=dev)
    half = nnz_balanced_row_split(rowptr.cpu(), 2)[1]
    res = {}
    for key, name, nrows, h in (("feature_split_1x8", "ds_parts = 8: all rows x 16 of 128 features (64-byte rows of X)", n, 16),
                                ("grid_2x4", "2 x 4 grid: half the rows (nnz-balanced) x 32 features (128-byte rows of X)", half, 32)):
        if only and key != only:
            continue
        m = int(rowptr[nrows])
        x = synth.features(n, h, torch.float32, seed=0, device=dev)
        out = torch.empty((nrows, h), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [nrows], [n], [m], [1], [h], h)
        torch.cuda.synchronize()
        create_ms = (time.perf_counter() - t0) * 1e3
        try:
            fn = lambda: _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), stream)
            ms = _median_ms(fn, steps, warm=1)
            k_ms = _kernel_ms(hd, fn, reps=2)
            cc = torch.bincount(col[:m].long(), minlength=n).double()
            ok = torch.equal(out.double().sum(0), cc @ x.double())
            del cc
            ok_rows = _sampled_rows_equal(rowptr, col, None, x, out, [(0, 200), (nrows - 200, nrows)], oracle)
            alg = synth.algorithmic_bytes(nrows, n, m, h, 4, "CSR", with_values=True)
            res[key] = {"workload": "ogbn-papers100M-shaped CSR SpMM, FLT32, h = 128 over 8 GPUs (configs[4]) -- ONE GPU's share: " + name,
                        "N": n, "rows": nrows, "nnz": m, "h": h, "dtype": "f32", "ms_per_step": round(ms, 3),
                        "value_GFLOPs": round(2 * m * h / (ms * 1e-3) / 1e9, 1), "group_create_ms": round(create_ms, 1),
                        "roofline": _roofline(alg, k_ms, _kernel_name(hd, "f32"), m * 128 + 2 * nrows * h * 4,
                                              "14.5 entries per row over 111 M uniform columns: one 128-byte line from HBM per stored entry whatever the row holds "
                                              "(a 64-byte row of X uses half of it); line_bytes = entries x 128 + C read and written",
                                              traffic_key="c5a" if key == "feature_split_1x8" else "c5b"),
                        "check": ("column-count checksum exact over all rows + 400 sampled rows bit-exact against the oracle" if ok and ok_rows else "MISMATCH")}
        finally:
            _lib.group_free(hd)
        del x, out
    return res


def config4_gcn_one_gpu(dev, h=256, fin=602, ncls=41, steps=4):
    """configs[3] on ONE GPU: Reddit-shaped graph, GCN of 3 layers, h = 256, FLT32 adjacency (the conv layer's quantise -> aggregate -> dequantise of
    models/pyg_gcn_conv.py:130-137 as one device call per layer); the 8-GPU row split of the same forward pass is inference.py --gpus 8"""
    import oracle
    from . import gnn
    from .dist import RowSplitAdj

    n, nnz, dmax = synth.SHAPES["reddit"]
    rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)

    class Capture:   # the first forward pass keeps every aggregation's input and output for the check
        def __init__(self, adj):
            self.adj, self.calls, self.dtype, self.on = adj, [], adj.dtype, True

        def mul_quantized(self, x, post=None):
            out, scale = self.adj.mul_quantized(x, post)
            if self.on:
                self.calls.append((x.clone(), out.clone()))
            return out, scale

    adj = Capture(RowSplitAdj(rowptr, col, n, torch.float32, h))
    hd = adj.adj.handle
    try:
        torch.manual_seed(1)
        model = gnn.GCN(fin, h, ncls, num_layers=3).to(dev).eval()
        torch.manual_seed(0)
        x = torch.randn(n, fin, device=dev)
        with torch.no_grad():
            logits = model(x, adj, None)
            adj.on = False
            fn = lambda: model(x, adj, None)
            ms = _median_ms(fn, steps, warm=1)
            k_ms = _kernel_ms(hd, fn, reps=2)
        ok = bool(torch.isfinite(logits).all()) and len(adj.calls) == 3
        rp = rowptr.to(torch.int64)
        longest = int(torch.argmax(rowptr[1:] - rowptr[:-1]))
        worst = 0.0
        for xin, out in adj.calls:
            s_ref, xq = oracle.symmetric_quantize(xin.cpu().numpy(), np.float32)
            for r0, r1 in ((0, 48), (n - 48, n), (longest, longest + 1)):
                lo, hi = int(rp[r0]), int(rp[r1])
                sub_rp = (rp[r0:r1 + 1] - lo).cpu().numpy().astype(np.int32)
                sub_col = col[lo:hi].cpu().numpy()
                want = oracle.symmetric_dequantize(oracle.spmm_csr(sub_rp, sub_col, None, xq), 1.0, s_ref).astype(np.float64)
                bound = oracle.spmm_csr(sub_rp, sub_col, None, np.abs(xq)).astype(np.float64) * float(s_ref) + 1e-30
                worst = max(worst, float(np.max(np.abs(out[r0:r1].cpu().numpy().astype(np.float64) - want) / bound)))
        alg = synth.algorithmic_bytes(n, n, nnz, h, 4, "CSR", with_values=True)
        return {"workload": "Reddit-shaped GCN inference, 3 layers, h = 256, FLT32 adjacency (configs[3]) -- one GPU's whole forward pass", "N": n, "nnz": nnz, "h": h,
                "in_features": fin, "classes": ncls, "dtype": "f32", "ms_per_forward": round(ms, 3), "aggregations_per_forward": 3,
                "roofline": _roofline(alg, k_ms, _kernel_name(hd, "f32") + "_deq (per aggregation; k_absmax_bits + k_slice_pack_quant in front)", None,
                                      "per aggregation: the same product as the headline (X staged through LDS, the level that bounds it: the headline's roofline.on_chip) "
                                      "with the dequantisation in the kernel's store"),
                "check": (f"every layer's aggregation on 97 sampled rows (first, last, longest) against the oracle's quantiser + CSR loop: max error {worst:.2e} of |A|.|x_q|.scale "
                          f"(bar 1e-5)" if ok and worst <= 1e-5 else f"MISMATCH (max error {worst:.2e} of the bound)")}
    finally:
        _lib.group_free(hd)


def end_to_end_cpu_tensors(rowptr, col, n, h, steps=4):
    """the reference driver's DEFAULT call pattern (spmm_test.py:29-35 with --device cpu; spmm_default/pytorch_api.cpp:269-271 returns a CPU
    tensor): X on the host, C on the host, per ``mul`` through the whole Python surface -- PCIe-inclusive, never the reported ``value``"""
    from .backend_pim import spmm as spmm_mod
    from . import pim_ops
    from .sparse_tensor import SparseTensorShim

    pim_ops.load("spmm")
    adj = SparseTensorShim(rowptr=rowptr, col=col, sparse_sizes=(n, n))
    A = spmm_mod.SparseTensorCOO(adj, dtype=torch.float32, format="CSR")
    A.to_pim_group(h, 1)
    x = synth.features(n, h, torch.float32, seed=0)   # pageable host memory, as the driver's torch.randint gives
    try:
        out = A.mul(x)   # (untimed: the first result tensor is a fresh page-locked allocation)
        ts = []
        for _ in range(steps + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = A.mul(x)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts_order = list(ts)
        ts = sorted(ts[2:])   # (the first calls of a group may still be settling between the copies and the direct stores: rt_run.inc)
        dev_x = x.cuda()
        want = A.mul(dev_x).cpu()
        call = _lib.group_host_call(A.sp_info_ptr)
        windows = call["windows"]
        timers = _lib.group_timers(A.sp_info_ptr)
        # the same call with upload, product and download one after the other (what every round before this one measured)
        prev = _lib.set_tunable("host_windows", 1)
        try:
            A.mul(x)
            serial = []
            for _ in range(steps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out1 = A.mul(x)
                torch.cuda.synchronize()
                serial.append((time.perf_counter() - t0) * 1e3)
            serial.sort()
        finally:
            _lib.set_tunable("host_windows", prev)
        dev_x = x.cuda()
        want = A.mul(dev_x).cpu()
        ok = out.device.type == "cpu" and torch.equal(out, want) and torch.equal(out1, want)
        return {"ms_per_mul": round(ts[len(ts) // 2], 3), "ms_per_mul_min": round(ts[0], 3), "steps": steps,
                "feature_windows": windows, "direct_stores": call["direct"], "ms_each_call": [round(t, 2) for t in ts_order], "ms_until_x_is_up": round(timers[0], 3), "ms_last_product_after_that": round(timers[1], 3),
                "ms_last_download_after_that": round(timers[2], 3), "ms_per_mul_serial": round(serial[len(serial) // 2], 3),
                "bytes_host_to_device": n * h * 4, "bytes_device_to_host": n * h * 4,
                "check": "equal to the device-resident product, element by element (pipelined and serial)" if ok else "MISMATCH",
                "note": "Reddit-shaped CSR FLT32 h = 256, CPU tensors in and out through backend_pim.spmm.SparseTensorCOO.mul (pageable X, pinned result): "
                        "two windows of 128 features -- window 1 goes up while window 0 is multiplied and comes down (rt_run.inc run_group_windows; "
                        "bit-identical: a feature window keeps each row's stored order); ms_per_mul_serial = upload, product, download one after the other "
                        "(tunable host_windows = 1); outside the timed region"}
    finally:
        A.free_group()
