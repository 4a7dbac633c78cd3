"""Randomised parity sweep on the GPU: shapes, degree skew, widths, element types, formats, sp/ds splits and
kernel-plan knobs drawn from a seeded generator; every result is compared with the oracle (bit-exact:
integer-valued features keep float sums exact in any order).  Also: hipGraph capture of a product."""
import numpy as np
import pytest
import torch

import oracle
from conftest import ALL_DTYPES, NP_DTYPES, coalesce, driver_features, random_csr
from pygim_amd import _lib
from test_parity_gpu import run_group_host

pytestmark = pytest.mark.gpu
# PYGIM_STRESS_SEEDS=N widens the randomised sweeps (default 24 / 48 cases; a one-off run with 1000 is cheap on the GPU)
import os as _os

_EXTRA = int(_os.environ.get("PYGIM_STRESS_SEEDS", "0"))


@pytest.fixture(scope="module", autouse=True)
def backend():
    assert torch.cuda.is_available()
    _lib.init_ranks(1)
    yield
    _lib.release()


def skewed_csr(rng, nrows, ncols, mean_deg, sigma, clustered):
    deg = np.minimum(np.floor(np.exp(rng.normal(np.log(max(mean_deg, 0.3)), sigma, nrows))).astype(np.int64), 6000)
    deg[rng.random(nrows) < 0.1] = 0
    rowptr = np.zeros(nrows + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    if clustered:
        centre = np.repeat((np.arange(nrows) * ncols / max(nrows, 1)).astype(np.int64), deg)
        col = np.clip(centre + rng.integers(-max(ncols // 20, 2), max(ncols // 20, 2), size=int(rowptr[-1])), 0, ncols - 1)
    else:
        col = rng.integers(0, ncols, size=int(rowptr[-1]), dtype=np.int64)
    for r in range(nrows):
        col[rowptr[r]:rowptr[r + 1]].sort()
    return rowptr.astype(np.int32), col.astype(np.int32)


@pytest.mark.parametrize("seed", range(max(24, _EXTRA)))
def test_random_configurations(seed):
    rng = np.random.default_rng(1000 + seed)
    dt = ALL_DTYPES[seed % 6]
    npdt = NP_DTYPES[dt]
    fmt = "COO" if rng.random() < 0.4 else "CSR"
    nrows = int(rng.integers(1, 900))
    ncols = int(rng.integers(1, 2500))
    h = int(rng.choice([1, 2, 7, 16, 31, 32, 48, 64, 100, 128, 200, 256, 320]))
    rowptr, col = skewed_csr(rng, nrows, ncols, float(rng.choice([0.5, 3, 12, 60])), float(rng.choice([0.3, 1.0, 1.6])),
                             bool(rng.random() < 0.4))
    weighted = bool(rng.random() < 0.4)
    knobs = {"panel_mode": int(rng.choice([0, 1, 2])), "panel_bytes": int(rng.choice([128 * 16, 128 * 200, 4 << 20])),
             "panel_coop": int(rng.choice([64, 512, 1 << 20])), "long_row_threshold": int(rng.choice([128, 4096])),
             "long_segment": int(rng.choice([64, 512])), "coo_chunk": int(rng.choice([64, 512])),
             "panel_pack": int(rng.choice([0, 1])), "slice_group_bytes": int(rng.choice([0, 1, ncols * 128 * 3, 640 << 20])),
             "panel_col16": int(rng.choice([0, 1])),
             # round 3: the LDS-staged product (INT32 / FLT32, rows of >= 33 features) in all its geometries, beside the others
             "lds_mode": int(rng.choice([0, 1, 1, 2])), "lds_waves": int(rng.choice([8, 16])),
             "lds_long_slots": int(rng.choice([0, 1, 128])), "lds_round_tiles": int(rng.choice([0, 1])),
             # round 4: the code-stream geometries (8 waves x 228 / 16 x 96 accumulators, rings of 2..6 buffers, read pipeline depth) for every
             # element type that has the form (INT8 widened, INT64 / DBL64 as register pairs)
             "lds_code": int(rng.choice([0, 1, 1, 1])), "lds_code_waves": int(rng.choice([0, 0, 8, 16])), "lds_code_nbuf": int(rng.choice([0, 0, 2, 3, 4, 5, 6])),
             "lds_code_gsize": int(rng.choice([0, 0, 2, 4, 6])), "lds_code_nsets": int(rng.choice([0, 0, 2, 3])), "lds_code_boundary": int(rng.choice([0, 2]))}
    old = {k: _lib.set_tunable(k, v) for k, v in knobs.items()}
    try:
        x = driver_features(rng, ncols, h, npdt)
        if fmt == "CSR":
            vals = rng.integers(-3, 4, size=len(col)).astype(npdt) if weighted else None
            ref = oracle.spmm_csr(rowptr, col, vals, x)
            out, _ = run_group_host("CSR", [rowptr], [col], None if vals is None else [vals], [nrows], [ncols], [x], h)
        else:
            r, c, v = coalesce(rowptr, col, npdt)
            if weighted:
                v = (v * rng.integers(1, 3, size=len(v))).astype(npdt)
            ref = oracle.spmm_coo(r, c, v, x, nrows)
            out, _ = run_group_host("COO", [r], [c], [v], [nrows], [ncols], [x], h)
        assert np.array_equal(out, ref), (seed, dt, fmt, nrows, ncols, h, knobs)
    finally:
        for k, v in old.items():
            _lib.set_tunable(k, v)


def _col_split(rowptr, col, nrows, ncols, parts):
    """reference col_split (spmm.py:127-136): parts of width ceil(ncols / parts), local column ids"""
    step = -(-ncols // parts)
    rows_of = np.repeat(np.arange(nrows), np.diff(rowptr))
    out = []
    for p in range(parts):
        lo, hi = p * step, min(ncols, (p + 1) * step)
        keep = (col >= lo) & (col < hi)
        rp = np.concatenate([[0], np.cumsum(np.bincount(rows_of[keep], minlength=nrows))]).astype(np.int32)
        out.append((rp, (col[keep] - lo).astype(np.int32), rows_of[keep].astype(np.int32), max(hi - lo, 0), keep))
    return out


@pytest.mark.parametrize("seed", range(max(48, _EXTRA)))
def test_random_groups(seed):
    """whole groups: sp_parts column blocks (summed) x ds_parts feature blocks (concatenated), default and grande call
    shapes, host operands, against the oracle's group driver (ops.hpp:42-62,97-118 restated)"""
    rng = np.random.default_rng(5000 + seed)
    dt = ALL_DTYPES[seed % 6]
    npdt = NP_DTYPES[dt]
    fmt = "COO" if seed % 3 == 1 else "CSR"
    nrows, ncols = int(rng.integers(1, 600)), int(rng.integers(4, 1500))
    sp_parts, ds_parts = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    h = int(rng.choice([4, 9, 32, 64, 100, 256]))
    ds_parts = min(ds_parts, h)
    rowptr, col = skewed_csr(rng, nrows, ncols, float(rng.choice([2, 10, 40])), float(rng.choice([0.5, 1.2])), False)
    weighted = bool(rng.random() < 0.5)
    vals_full = rng.integers(-3, 4, size=len(col)).astype(npdt) if weighted else np.ones(len(col), dtype=npdt)
    knobs = {"fuse_windows": int(rng.choice([0, 1])), "panel_mode": int(rng.choice([0, 1, 2])),
             "panel_bytes": int(rng.choice([128 * 64, 4 << 20])), "slice_group_bytes": int(rng.choice([1, 640 << 20])),
             "panel_col16": int(rng.choice([0, 1])), "merge_parts": int(rng.choice([0, 1])),
             "split_unit_pattern": int(rng.choice([0, 1])), "lds_mode": int(rng.choice([0, 1, 1, 2])),
             "lds_long_slots": int(rng.choice([0, 1, 128])), "narrow_vals": int(rng.choice([0, 1])),
             "lds_code": int(rng.choice([0, 1, 1])), "lds_code_waves": int(rng.choice([0, 0, 8, 16])), "lds_code_nbuf": int(rng.choice([0, 0, 2, 3, 4, 5])),
             "lds_code_gsize": int(rng.choice([0, 0, 4])), "lds_code_nsets": int(rng.choice([0, 0, 3]))}
    old = {k: _lib.set_tunable(k, v) for k, v in knobs.items()}
    try:
        x = driver_features(rng, ncols, h, npdt)
        ref = oracle.spmm_csr(rowptr, col, vals_full if weighted else None, x)
        parts = _col_split(rowptr, col, nrows, ncols, sp_parts)
        widths = [len(c) for c in np.array_split(np.arange(h), ds_parts)]  # torch.chunk-like: ceil first
        widths = [w for w in widths if w > 0]
        offs = np.cumsum([0] + widths[:-1])
        chunks = [np.ascontiguousarray(x[:, a:a + w]) for a, w in zip(offs, widths)]
        idx0 = [p[0] if fmt == "CSR" else p[2] for p in parts]
        cols = [p[1] for p in parts]
        vals = [vals_full[p[4]] for p in parts] if (weighted or fmt == "COO") else None
        nr, nc = [nrows] * sp_parts, [p[3] for p in parts]
        out, _ = run_group_host(fmt, idx0, cols, vals, nr, nc, chunks, h)
        assert np.array_equal(out, ref), (seed, dt, fmt, sp_parts, ds_parts, h, knobs, "default")
        if fmt == "CSR":
            # grande call shape: per-part windows [part cols, padded width]
            pad = lambda w: -(-w * npdt().itemsize // 8) * 8 // npdt().itemsize
            wins, lds, lo = [], [], 0
            for p in parts:
                for a, w in zip(offs, widths):
                    buf = np.full((p[3], pad(w)), 55, dtype=npdt)
                    buf[:, :w] = x[lo:lo + p[3], a:a + w]
                    wins.append(buf)
                    lds.append(pad(w))
                lo += p[3]
            out_g, _ = run_group_host("CSR", idx0, cols, vals, nr, nc, wins, h, kind="grande",
                                      n_dense=[len(widths)] * sp_parts, dense_cols=widths * sp_parts, lds=lds)
            assert np.array_equal(out_g, ref), (seed, dt, sp_parts, ds_parts, h, knobs, "grande")
    finally:
        for k, v in old.items():
            _lib.set_tunable(k, v)


def test_product_is_graph_capturable(rng=np.random.default_rng(3)):
    """the run entry points only enqueue work on the caller's stream (plus a forked side stream joined by
    events), so a warmed-up product can be captured into a hipGraph and replayed"""
    rowptr, col = skewed_csr(rng, 3000, 3000, 30, 1.2, False)
    rowptr[-1] = len(col)
    x = driver_features(rng, 3000, 128, np.float32)
    d = lambda a: torch.from_numpy(a).cuda()
    drp, dcol, dx = d(rowptr), d(col), d(x)
    hd = _lib.group_create(_lib.CSR, _lib.FLT32, [drp.data_ptr()], [dcol.data_ptr()], None, [3000], [3000], [len(col)], [1], [128], 128)
    out = torch.zeros((3000, 128), dtype=torch.float32, device="cuda")
    ref = oracle.spmm_csr(rowptr, col, None, x)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):  # warm-up: scratch buffers get their size outside the capture
            _lib.spmm_run_group(hd, [dx.data_ptr()], out.data_ptr(), s.cuda_stream)
    s.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        _lib.spmm_run_group(hd, [dx.data_ptr()], out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    out.zero_()
    dx.copy_(torch.from_numpy(x))  # same buffers, replay
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref)
    _lib.group_free(hd)


def test_concurrent_products_on_two_streams_with_different_features():
    """two groups, two streams, two different X in flight at once: each product gathers from its own slice-major
    copy (one buffer per launch stream), so neither disturbs the other"""
    rng = np.random.default_rng(17)
    n, h = 20000, 256
    rowptr, col = skewed_csr(rng, n, n, 40, 1.0, False)
    d = lambda a: torch.from_numpy(a).cuda()
    rp, cl = d(rowptr), d(col)
    xs = [driver_features(rng, n, h, np.int32) for _ in range(2)]
    refs = [oracle.spmm_csr(rowptr, col, None, x) for x in xs]
    xd = [d(x) for x in xs]
    outs = [torch.empty((n, h), dtype=torch.int32, device="cuda") for _ in range(2)]
    hds = [_lib.group_create(_lib.CSR, _lib.INT32, [rp.data_ptr()], [cl.data_ptr()], None, [n], [n], [len(col)], [1], [h], h)
           for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    try:
        torch.cuda.synchronize()
        for rep in range(6):
            for k in range(2):
                _lib.spmm_run_group(hds[k], [xd[k].data_ptr()], outs[k].data_ptr(), streams[k].cuda_stream)
        torch.cuda.synchronize()
        for k in range(2):
            assert np.array_equal(outs[k].cpu().numpy(), refs[k]), k
    finally:
        for hd in hds:
            _lib.group_free(hd)


def test_group_create_free_does_not_leak_device_memory():
    """groups with every optional structure (merged matrix, correction part, 16-bit ids, long-row plans, staging and
    window buffers) created, run and freed repeatedly: the free device memory comes back"""
    rng = np.random.default_rng(23)
    nrows, ncols, h = 3000, 3000, 64
    rowptr, col = skewed_csr(rng, nrows, ncols, 30, 1.3, False)
    vals = np.ones(len(col), dtype=np.int32)
    vals[rng.choice(len(col), size=len(col) // 300, replace=False)] = 3
    parts = _col_split(rowptr, col, nrows, ncols, 3)
    x = driver_features(rng, ncols, h, np.int32)
    chunks = [np.ascontiguousarray(x[:, :40]), np.ascontiguousarray(x[:, 40:])]
    ref = oracle.spmm_csr(rowptr, col, vals, x)

    def once():
        out, _ = run_group_host("CSR", [p[0] for p in parts], [p[1] for p in parts], [vals[p[4]] for p in parts], [nrows] * 3,
                                [p[3] for p in parts], chunks, h)
        assert np.array_equal(out, ref)

    once()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(30):
        once()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (32 << 20), (free0, free1)


def test_more_streams_than_slice_major_buffers(rng):
    """the slice-major copies live in at most four per-(device, stream) buffers (least recently used goes first): products on
    six streams in turn, each on its own X, stay correct; then the explicit x_unchanged argument shares one stream's copy with
    another stream (ordered by an event) and a product on a REWRITTEN X without the flag sees the new contents"""
    n, h = 3000, 256
    rowptr, col = random_csr(rng, n, n, 40, long_rows=[(11, 5000)])
    rp, cl = torch.from_numpy(rowptr).cuda(), torch.from_numpy(col).cuda()
    if True:
        hd = _lib.group_create(_lib.CSR, _lib.INT32, [rp.data_ptr()], [cl.data_ptr()], None, [n], [n], [cl.numel()], [1], [h], h)
        streams = [torch.cuda.Stream() for _ in range(6)]
        xs = [torch.from_numpy(driver_features(rng, n, h, np.int32)).cuda() for _ in range(6)]
        refs = [oracle.spmm_csr(rowptr, col, None, x.cpu().numpy()) for x in xs]
        outs = [torch.empty((n, h), dtype=torch.int32, device="cuda") for _ in range(6)]
        torch.cuda.synchronize()
        for rnd in range(3):
            for k in range(6):
                with torch.cuda.stream(streams[k]):
                    _lib.spmm_run_group(hd, [xs[k].data_ptr()], outs[k].data_ptr(), streams[k].cuda_stream)
                torch.cuda.synchronize()  # one group serves one call at a time
                assert np.array_equal(outs[k].cpu().numpy(), refs[k]), (rnd, k)
        # share stream 0's copy of xs[0] with stream 1
        with torch.cuda.stream(streams[0]):
            _lib.spmm_run_group(hd, [xs[0].data_ptr()], outs[0].data_ptr(), streams[0].cuda_stream)
            ev = torch.cuda.Event()
            ev.record(streams[0])
        torch.cuda.synchronize()
        streams[1].wait_event(ev)
        with torch.cuda.stream(streams[1]):
            _lib.spmm_run_group(hd, [xs[0].data_ptr()], outs[1].data_ptr(), streams[1].cuda_stream, x_unchanged=True)
        torch.cuda.synchronize()
        assert np.array_equal(outs[1].cpu().numpy(), refs[0])
        # rewrite X in place: without the flag the copy is made again
        xs[0].copy_(xs[3])
        with torch.cuda.stream(streams[1]):
            _lib.spmm_run_group(hd, [xs[0].data_ptr()], outs[1].data_ptr(), streams[1].cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(outs[1].cpu().numpy(), refs[3])
        _lib.group_free(hd)


@pytest.mark.parametrize("seed", range(max(36, _EXTRA)))
def test_random_quantised_aggregations(seed):
    """the conv layers' quantise -> aggregate -> dequantise through every form of the C ABI -- one fused call, with and
    without the per-column epilogue, and quantise + product-with-dequantising-store -- on random shapes, skews, widths,
    element types, strides, sp_parts (merged groups) and plan knobs, against the oracle's statement of
    models/quantize.py:20-42 (integers bit-exact; FLT32 within 1e-5 of |A| . |x_q| . scale)"""
    from pygim_amd.pim_ops import DTYPE_CODE

    rng = np.random.default_rng(9000 + seed)
    tdt = [torch.int8, torch.int16, torch.int32, torch.float32][seed % 4]
    npdt = {torch.int8: np.int8, torch.int16: np.int16, torch.int32: np.int32, torch.float32: np.float32}[tdt]
    nrows = ncols = int(rng.integers(2, 1500))
    h = int(rng.choice([1, 3, 5, 16, 31, 32, 64, 100, 128, 255, 256]))
    rowptr, col = skewed_csr(rng, nrows, ncols, float(rng.choice([0.5, 3, 12, 60])), float(rng.choice([0.3, 1.0, 1.6])),
                             bool(rng.random() < 0.4))
    sp_parts = int(rng.choice([1, 1, 2, 3]))
    knobs = {"panel_mode": int(rng.choice([0, 1, 1, 2])), "panel_bytes": int(rng.choice([128 * 16, 128 * 200, 4 << 20])),
             "panel_coop": int(rng.choice([64, 512])), "long_row_threshold": int(rng.choice([128, 4096])),
             "panel_pack": int(rng.choice([0, 1, 1])), "panel_col16": int(rng.choice([0, 1])),
             "slice_group_bytes": int(rng.choice([0, 1, 640 << 20])), "merge_parts": int(rng.choice([0, 1, 1])),
             "fuse_windows": int(rng.choice([0, 1, 1])), "lds_mode": int(rng.choice([0, 1, 1, 2])),
             "lds_long_slots": int(rng.choice([0, 1, 128])), "lds_code": int(rng.choice([0, 1, 1]))}
    old = {k: _lib.set_tunable(k, v) for k, v in knobs.items()}
    try:
        parts = _col_split(rowptr, col, nrows, ncols, sp_parts)
        keep = []
        rps, cls, nnzs, ncs = [], [], [], []
        for rp, cl, _rows, width, _k in parts:
            a, b = torch.from_numpy(rp).cuda(), torch.from_numpy(cl).cuda()
            keep += [a, b]
            rps.append(a.data_ptr())
            cls.append(b.data_ptr())
            nnzs.append(len(cl))
            ncs.append(width)
        hd = _lib.group_create(_lib.CSR, DTYPE_CODE[tdt], rps, cls, None, [nrows] * sp_parts, ncs, nnzs, [1] * sp_parts,
                               [h] * sp_parts, h)
        ld = h + int(rng.integers(0, 4))
        xs = (rng.standard_normal((ncols, ld)) * float(rng.choice([0.01, 1.0, 300.0]))).astype(np.float32)
        x = torch.from_numpy(xs).cuda()
        xv = xs[:, :h]
        s_ref, xq = oracle.symmetric_quantize(xv, npdt)
        prod = oracle.spmm_csr(rowptr, col, None, xq)
        want = oracle.symmetric_dequantize(prod, 1.0, s_ref)
        bound = 1e-5 * oracle.spmm_csr(rowptr, col, None, np.abs(xq)).astype(np.float64) * float(s_ref)

        # FLT32 keeps the quantised values as floats and sums them in float32: the bar is 1e-5 of |A| . |x_q| . scale around the
        # EXACT sum (float64).  The sequential float32 loop of the CPU path is itself only that accurate -- on a 6000-entry row
        # whose partial sums pass 2^24 it sits 4 bounds away from exact while the device's segment-wise sum sits at 0.15
        # (seed 219).  Rows the device sums in stored order reproduce that loop (and its error) instead (seed 595: 88 column
        # panels, each continuing the row's running sum).  Every element must be within one bound of one of the two.
        import scipy.sparse as _sp

        exact = None
        if tdt == torch.float32:
            a64 = _sp.csr_matrix((np.ones(len(col)), col.astype(np.int64), rowptr.astype(np.int64)), shape=(nrows, ncols))
            exact = (a64 @ xq.astype(np.float64)) * float(s_ref)

        def same(got, ref, extra=0.0):
            if tdt != torch.float32:
                return np.array_equal(got, ref)
            g = got.astype(np.float64)
            near_exact = np.abs(g - exact) <= bound + 1e-30           # rows summed segment-wise / by a whole wave
            near_seq = np.abs(g - ref.astype(np.float64)) <= bound + 1e-30  # rows summed in stored order, like the CPU loop
            return bool(np.all(near_exact | near_seq))

        out = torch.full((nrows, h), float("nan"), dtype=torch.float32, device="cuda")
        scale = torch.zeros((), dtype=torch.float32, device="cuda")
        _lib.quant_spmm_run(hd, x.data_ptr(), ld, out.data_ptr(), scale.data_ptr())
        torch.cuda.synchronize()
        assert np.float32(scale.item()) == s_ref, (seed, "scale")
        assert same(out.cpu().numpy(), want), (seed, tdt, nrows, h, sp_parts, knobs, "fused")
        # epilogue: bit-identical to torch's a * out + b, relu
        a = (torch.rand(h, device="cuda") + 0.5)
        b = torch.randn(h, device="cuda")
        out2 = torch.full((nrows, h), float("nan"), dtype=torch.float32, device="cuda")
        _lib.quant_spmm_run(hd, x.data_ptr(), ld, out2.data_ptr(), 0, 0, a.data_ptr(), b.data_ptr(), True)
        torch.cuda.synchronize()
        assert torch.equal(out2, torch.relu(a * out + b)), (seed, "epilogue")
        # two steps: quantise, then the product whose last store dequantises
        bits = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.quant_absmax(x.data_ptr(), ld, ncols, h, bits.data_ptr())
        xqd = torch.empty((ncols, h), dtype=tdt, device="cuda")
        _lib.quantize(DTYPE_CODE[tdt], x.data_ptr(), ld, ncols, h, bits.data_ptr(), xqd.data_ptr())
        out3 = torch.full((nrows, h), float("nan"), dtype=torch.float32, device="cuda")
        _lib.spmm_run_dequant(hd, xqd.data_ptr(), h, out3.data_ptr(), bits.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(xqd.cpu().numpy(), xq), (seed, "quantise")
        assert same(out3.cpu().numpy(), want), (seed, tdt, nrows, h, sp_parts, knobs, "two-step")
        _lib.group_free(hd)
    finally:
        for k, v in old.items():
            _lib.set_tunable(k, v)


@pytest.mark.parametrize("seed", range(max(24, _EXTRA)))
def test_random_lds_spmv(seed):
    """k_spmv_lds on random shapes: rows of X of 1..4 elements, every element type, unit and real weights, panels from a handful
    of columns (more units than CUs) to the whole matrix, all four length classes, empty rows, accumulate through a second call"""
    rng = np.random.default_rng(12000 + seed)
    dt = ALL_DTYPES[seed % 6]
    npdt = NP_DTYPES[dt]
    nrows = int(rng.integers(1, 1500))
    ncols = int(rng.integers(1, 4000))
    w = int(rng.integers(1, 5))
    rowptr, col = skewed_csr(rng, nrows, ncols, float(rng.choice([2, 30, 150, 500])), float(rng.choice([0.3, 1.0, 1.6])),
                             bool(rng.random() < 0.3))
    weighted = bool(rng.random() < 0.4)
    knobs = {"panel_mode": 1, "vec_lds_min_seg": 0, "panel_bytes": int(rng.choice([128 * 4, 128 * 64, 128 * 700, 4 << 20])),
             "merge_parts": 0, "split_unit_pattern": int(rng.choice([0, 1]))}
    old = {k: _lib.set_tunable(k, v) for k, v in knobs.items()}
    try:
        x = driver_features(rng, ncols, w, npdt)
        vals = rng.integers(-3, 4, size=len(col)).astype(npdt) if weighted else None
        ref = oracle.spmm_csr(rowptr, col, vals, x)
        out, info = run_group_host("CSR", [rowptr], [col], None if vals is None else [vals], [nrows], [ncols], [x], w)
        assert np.array_equal(out, ref), (seed, dt, nrows, ncols, w, weighted, knobs, info)
    finally:
        for k, v in old.items():
            _lib.set_tunable(k, v)


def test_random_host_operand_pipelines():
    """the host-operand pipeline over random shapes, element types, sparse parts, caller feature blocks, window counts, direct stores on / off, page-locked or pageable
    results (scripts/stress_host_pipeline.py, in its own process: it changes tunables): every pipelined result byte-equal to the serial call's; forced float windows
    on the sweep inside the norm-wise bound"""
    import subprocess
    import sys

    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    cases = 40 + 4 * _EXTRA
    r = subprocess.run([sys.executable, _os.path.join(root, "scripts", "stress_host_pipeline.py"), str(cases), "11"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and f"host pipeline: {cases} cases" in r.stdout, (r.stdout[-600:], r.stderr[-1200:])
