"""BASELINE.json's full-size configurations on the GPU, checked through size-independent
properties (the oracle cannot run 58 GFLOP in seconds): column-count checksum, linearity,
idempotence of repeated runs, and a sampled-rows comparison with the oracle."""
import numpy as np
import pytest
import torch

import oracle
from pygim_amd import _lib, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def backend():
    assert torch.cuda.is_available()
    _lib.init_ranks(1)
    yield
    _lib.release()


def run(handle, x, n, h):
    out = torch.empty((n, h), dtype=x.dtype, device=x.device)
    _lib.spmm_run_group(handle, [x.data_ptr()], out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out


def sample_rows_vs_oracle(rowptr, col, vals, x, out, rows):
    rp = rowptr.cpu().numpy().astype(np.int64)
    xh = x.cpu().numpy()
    for r0, r1 in rows:
        lo, hi = rp[r0], rp[r1]
        sub_rp = (rp[r0:r1 + 1] - lo).astype(np.int32)
        sub_col = col[lo:hi].cpu().numpy()
        sub_val = None if vals is None else vals[lo:hi].cpu().numpy()
        ref = oracle.spmm_csr(sub_rp, sub_col, sub_val, xh)
        assert np.array_equal(out[r0:r1].cpu().numpy(), ref), (r0, r1)


def test_reddit_csr_f32_h256():
    """configs[1]: Reddit-shaped CSR, h = 256, FLT32 (the benchmark workload)"""
    dev = torch.device("cuda", 0)
    n, nnz, d_max = synth.SHAPES["reddit"]
    h = 256
    rowptr, col = synth.make_csr(n, nnz, d_max, seed=0, device=dev)
    x1 = synth.features(n, h, torch.float32, seed=0, device=dev)
    x2 = synth.features(n, h, torch.float32, seed=5, device=dev)
    hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
    try:
        # the product this workload takes: the LDS-staged schedule compiled into machine code (the sweep's plan is built beside it and
        # serves accumulate-into-C calls of other widths)
        assert _lib.group_lds_code(hd)["active"] == 1 and _lib.group_lds_plan(hd)["nnz"] == nnz
        geo = _lib.group_lds_geometry(hd)
        assert geo["waves"] * geo["acc_per_wave"] >= -(-n // _lib.group_lds_plan(hd)["tiles"]) and geo["chunk_cols"] * 256 * geo["buffers"] <= 163840, geo
        c1 = run(hd, x1, n, h)
        # checksum: sum_i C[i, :] = sum_j colcount[j] * X[j, :]   (small integers: exact in f64)
        colcount = torch.bincount(col.long(), minlength=n).double()
        assert torch.equal(c1.double().sum(0), colcount @ x1.double())
        # linearity (exact: every partial sum is a small integer)
        c2 = run(hd, x2, n, h)
        c12 = run(hd, x1 + x2, n, h)
        assert torch.equal(c12, c1 + c2)
        # idempotence / determinism
        assert torch.equal(run(hd, x1, n, h), c1)
        # real-valued features: deterministic, and within 1e-5 of an f64 evaluation of sampled rows
        xr = synth.features(n, h, torch.float32, seed=1, device=dev, kind="uniform")
        cr = run(hd, xr, n, h)
        assert torch.equal(run(hd, xr, n, h), cr)
        deg = (rowptr[1:] - rowptr[:-1]).long()
        longest = int(torch.argmax(deg))
        sample_rows_vs_oracle(rowptr, col, None, x1, c1, [(0, 300), (n - 200, n), (longest, longest + 1)])
        rows = torch.tensor([1, 1234, longest, n - 1], device=dev)
        for r in rows.tolist():
            cols = col[int(rowptr[r]):int(rowptr[r + 1])].long()
            ref = xr[cols].double().sum(0)
            scale = xr[cols].double().abs().sum(0)
            assert torch.all((cr[r].double() - ref).abs() <= 1e-5 * scale + 1e-30)
        # ... and EVERY row against the oracle's row-parallel CSR loop (all host cores): the bar is 1e-5 of |A| . |x|; relative
        # to the result itself the same bound holds wherever the terms do not cancel (|result| >= 10 % of |A| . |x|)
        ref = np.zeros((n, h), dtype=np.float32)
        oracle.spmm_csr_rowpar(rowptr.cpu().numpy().astype(np.uint32), col.cpu().numpy().astype(np.uint32), None, xr.cpu().numpy(),
                               nthreads=oracle.max_threads(), out=ref)
        ca = run(hd, xr.abs(), n, h).cpu().numpy().astype(np.float64)
        got = cr.cpu().numpy().astype(np.float64)
        err = np.abs(got - ref.astype(np.float64))
        assert np.all(err <= 1e-5 * ca + 1e-30)
        solid = np.abs(ref) >= 0.1 * ca
        assert np.all(err[solid] <= 1e-5 * np.abs(ref.astype(np.float64))[solid])
        assert np.mean(got == ref) > 0.98  # one lane group sums a row in stored order: mostly bit-identical
    finally:
        _lib.group_free(hd)


def test_reddit_default_call_with_cpu_tensors_is_a_pipeline_of_two_feature_windows():
    """the reference driver's default call at the benchmark's size (spmm_test.py:29-35: CPU tensors in, a CPU tensor back): through the Python surface the
    238.6 MB of X go up and the 238.6 MB of C come down as two windows of 128 features around two half-width products (run_group_windows) -- bit-identical
    to the device-resident product (real-valued features: a window keeps each row's stored order), for a pageable and for a page-locked X; the serial call
    (host_windows = 1) and the direct mode (host_direct = 2: no staged C, the kernel stores into the host tensor) give the same bits"""
    from pygim_amd import pim_ops
    from pygim_amd.backend_pim import spmm as spmm_mod
    from pygim_amd.sparse_tensor import SparseTensorShim

    dev = torch.device("cuda", 0)
    n, nnz, d_max = synth.SHAPES["reddit"]
    h = 256
    rowptr, col = synth.make_csr(n, nnz, d_max, seed=0, device=dev)
    pim_ops.load("spmm")
    A = spmm_mod.SparseTensorCOO(SparseTensorShim(rowptr=rowptr, col=col, sparse_sizes=(n, n)), dtype=torch.float32, format="CSR")
    A.to_pim_group(h, 1)
    try:
        x = synth.features(n, h, torch.float32, seed=1, kind="uniform")
        want = A.mul(x.to(dev)).cpu()
        for xin in (x, x.pin_memory()):
            out = A.mul(xin)
            assert out.device.type == "cpu" and _lib.group_host_windows(A.sp_info_ptr) == 2
            assert torch.equal(out, want)
        old = _lib.set_tunable("host_windows", 1)
        try:
            out = A.mul(x)
            assert _lib.group_host_windows(A.sp_info_ptr) == 1 and torch.equal(out, want)
        finally:
            _lib.set_tunable("host_windows", old)
        # the direct mode (the products' store stage writes the page-locked result itself: what a group turns to when the runtime's pitched copies to
        # the host run slow) and the staged-C copies, each forced
        for direct in (2, 0):
            old = _lib.set_tunable("host_direct", direct)
            try:
                out = A.mul(x)
                assert _lib.group_host_call(A.sp_info_ptr) == {"windows": 2, "direct": 1 if direct else 0}
                assert torch.equal(out, want)
            finally:
                _lib.set_tunable("host_direct", old)
    finally:
        A.free_group()


def test_products_coo_i32_h256():
    """configs[2]: ogbn-products-shaped COO, h = 256, INT32, bit-exact"""
    dev = torch.device("cuda", 0)
    n, nnz, d_max = synth.SHAPES["ogbn-products"]
    h = 256
    rowptr, col = synth.make_csr(n, nnz, d_max, seed=0, device=dev)
    row, ccol, val = synth.csr_to_coo_coalesced(rowptr, col, torch.int32)
    del col
    m = row.numel()
    x = synth.features(n, h, torch.int32, seed=0, device=dev)
    hd = _lib.group_create(_lib.COO, _lib.INT32, [row.data_ptr()], [ccol.data_ptr()], [val.data_ptr()], [n], [n], [m],
                           [1], [h], h)
    try:
        c = run(hd, x, n, h)
        # checksum modulo 2^32 == exact in int64 here (|sums| stay far below 2^31)
        w = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, ccol.long(), val.double())
        assert torch.equal(c.double().sum(0), w @ x.double())
        assert torch.equal(run(hd, x, n, h), c)
        # sampled rows against the oracle's COO loop
        rp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        rp[1:] = torch.cumsum(torch.bincount(row.long(), minlength=n), 0)
        xh = x.cpu().numpy()
        for r0, r1 in ((0, 400), (n - 300, n)):
            lo, hi = int(rp[r0]), int(rp[r1])
            ref = oracle.spmm_coo((row[lo:hi] - r0).cpu().numpy(), ccol[lo:hi].cpu().numpy(), val[lo:hi].cpu().numpy(), xh, r1 - r0)
            assert np.array_equal(c[r0:r1].cpu().numpy(), ref)
    finally:
        _lib.group_free(hd)


@pytest.mark.parametrize("name", ["INT8", "INT16", "INT64", "DBL64"])
def test_reddit_csr_other_types_h256(name):
    """the benchmark graph in the other element types: column-count checksum (modulo 2^bits for the integers: int8 /
    int16 sums of ~500 terms wrap, as val_dt arithmetic does), sampled rows against the oracle, determinism"""
    dev = torch.device("cuda", 0)
    tdt = {"INT8": torch.int8, "INT16": torch.int16, "INT64": torch.int64, "DBL64": torch.float64}[name]
    code = {"INT8": _lib.INT8, "INT16": _lib.INT16, "INT64": _lib.INT64, "DBL64": _lib.DBL64}[name]
    n, nnz, d_max = synth.SHAPES["reddit"]
    h = 256
    rowptr, col = synth.make_csr(n, nnz, d_max, seed=0, device=dev)
    x = synth.features(n, h, tdt, seed=3, device=dev)
    hd = _lib.group_create(_lib.CSR, code, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [h], h)
    try:
        c = run(hd, x, n, h)
        assert torch.equal(run(hd, x, n, h), c)
        colcount = torch.bincount(col.long(), minlength=n)
        if tdt.is_floating_point:
            assert torch.equal(c.sum(0), colcount.double() @ x)  # small integers: exact
        else:
            bits = torch.iinfo(tdt).bits
            # every row sum wraps modulo 2^bits; so does their sum: compare modulo 2^bits, computed exactly in float64
            # chunks (|colcount . x| < 2^53)
            want = (colcount.double() @ x.double())
            got = c.to(torch.int64).sum(0)
            if bits < 64:
                mod = float(2 ** bits)
                assert torch.equal(torch.remainder(got.double() - want, mod), torch.zeros_like(want))
            else:
                assert torch.equal(got.double(), want)
        deg = (rowptr[1:] - rowptr[:-1]).long()
        longest = int(torch.argmax(deg))
        sample_rows_vs_oracle(rowptr, col, None, x, c, [(0, 200), (n - 100, n), (longest, longest + 1)])
    finally:
        _lib.group_free(hd)


def sample_rows_compact(rowptr, col, x, out, rows):
    """sampled row ranges against the oracle, with only the gathered rows of X copied to the host (X may be many GB)"""
    for r0, r1 in rows:
        lo, hi = int(rowptr[r0]), int(rowptr[r1])
        sub_rp = (rowptr[r0:r1 + 1].to(torch.int64) - lo).cpu().numpy().astype(np.int32)
        uniq, inv = torch.unique(col[lo:hi].long(), return_inverse=True)
        ref = oracle.spmm_csr(sub_rp, inv.cpu().numpy().astype(np.int32), None, x[uniq].cpu().numpy())
        assert np.array_equal(out[r0:r1].cpu().numpy(), ref), (r0, r1)


def test_papers100m_per_gpu_slices_f32():
    """configs[4] (ogbn-papers100M CSR, h = 128, FLT32 over 8 GPUs), the work of ONE GPU at full size:
    (a) ds_parts = 8 feature split: all 111 M rows x 16 of the 128 features (64-byte rows: one panel, gathers straight
        from the row-major X, 7.1 GB -> 64-bit gather offsets, AMODE 0);
    (b) the 2 x 4 grid pygim_amd.autotune prefers: the first nnz-balanced half of the rows x 32 features (128-byte rows,
        14.2 GB of X -> 64-bit offsets again).
    Checked through size-independent properties: column-count checksum (exact: integer-valued features), determinism,
    sampled rows against the oracle (first / last / longest rows), real-valued features within 1e-5 on sampled rows."""
    from bench import nnz_balanced_row_split

    dev = torch.device("cuda", 0)
    n, nnz, d_max = synth.SHAPES["ogbn-papers100M"]
    rowptr, col = synth.make_csr(n, nnz, d_max, seed=0, device=dev)
    deg = (rowptr[1:] - rowptr[:-1])
    longest = int(torch.argmax(deg))
    colcount = torch.bincount(col.long(), minlength=n).double()
    quarters = nnz_balanced_row_split(rowptr.cpu(), 4)
    half, quarter = quarters[2], quarters[1]
    # (c) a 4 x 2 grid share: a quarter of the rows x 64 features = two 128-byte slices; the slice-major copy (28 GB) is
    #     far beyond the 640 MiB slice group, so the slices are swept one launch at a time -- with 64-bit offsets
    for name, nrows, h in (("1x8 feature split", n, 16), ("2x4 grid", half, 32), ("4x2 grid", quarter, 64)):
        m = int(rowptr[nrows])
        x = synth.features(n, h, torch.float32, seed=0, device=dev)
        assert n * h * 4 > 2 ** 32, "this case is about operands beyond 4 GiB"
        hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [nrows], [n], [m], [1], [h], h)
        try:
            plan = _lib.group_plan(hd)
            assert plan["n_panels"] == 1 and plan["n_segment_tasks"] == 0, (name, plan)  # 14.5 entries per row: one-panel sweep
            c = run(hd, x, nrows, h)
            cc = colcount if nrows == n else torch.bincount(col[:m].long(), minlength=n).double()
            assert torch.equal(c.double().sum(0), cc @ x.double()), name
            assert torch.equal(run(hd, x, nrows, h), c), name
            picks = [(0, 200), (nrows - 200, nrows)] + ([(longest, longest + 1)] if longest < nrows else [])
            sample_rows_compact(rowptr, col, x, c, picks)
            del c
            xr = synth.features(n, h, torch.float32, seed=1, device=dev, kind="uniform")
            cr = run(hd, xr, nrows, h)
            for r in (0, nrows // 3, nrows - 1) + ((longest,) if longest < nrows else ()):
                cols = col[int(rowptr[r]):int(rowptr[r + 1])].long()
                ref = xr[cols].double().sum(0)
                assert torch.all((cr[r].double() - ref).abs() <= 1e-5 * ref.abs() + 1e-6 * xr[cols].double().abs().sum(0)), (name, r)
            del xr, cr
        finally:
            _lib.group_free(hd)
        del x


@pytest.mark.parametrize("w,tdt,code,weighted", [(1, torch.int32, "INT32", False), (2, torch.float32, "FLT32", False),
                                                 (4, torch.int32, "INT32", True)])
def test_reddit_spmv_end_full_size(w, tdt, code, weighted):
    """the SpMV end of the path (spmv_sparseP: rows of X of 1..4 elements) on the Reddit-shaped graph at full size: the LDS-staged
    kernel's plan (several panels, all four length classes), column-count checksum, determinism, sampled rows against the oracle"""
    dev = torch.device("cuda", 0)
    n, nnz, d_max = synth.SHAPES["reddit"]
    rowptr, col = synth.make_csr(n, nnz, d_max, seed=0, device=dev)
    x = synth.features(n, w, tdt, seed=3, device=dev)
    vals = torch.randint(-3, 4, (nnz,), device=dev, dtype=tdt) if weighted else None
    hd = _lib.group_create(_lib.CSR, getattr(_lib, code), [rowptr.data_ptr()], [col.data_ptr()], None if vals is None else [vals.data_ptr()],
                           [n], [n], [nnz], [1], [w], w)
    try:
        plan = _lib.group_plan(hd)
        assert plan["n_panels"] >= 8 and plan["col16"] == 1, plan   # panels sized for a workgroup's LDS
        c1 = run(hd, x, n, w)
        if vals is None:
            colw = torch.bincount(col.long(), minlength=n).double()
        else:
            colw = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, col.long(), vals.double())
        assert torch.equal(c1.double().sum(0), colw @ x.double())
        assert torch.equal(run(hd, x, n, w), c1)
        sample_rows_vs_oracle(rowptr, col, vals, x, c1, [(0, 64), (n // 2, n // 2 + 64), (n - 64, n)])
        longest = int((rowptr[1:] - rowptr[:-1]).argmax())
        sample_rows_vs_oracle(rowptr, col, vals, x, c1, [(longest, longest + 1)])
    finally:
        _lib.group_free(hd)


@pytest.mark.parametrize("clustered", [False, True])
def test_reddit_lds_staged_product_is_the_cpu_loop_bit_for_bit(clustered):
    """configs[1] on the LDS-staged kernel (the default for this shape since round 3), uniform and community-like columns:
    every output of the real-valued product equals the oracle's sequential loop bit for bit (with and without entry values),
    and the driver-feature product equals the L2 sweep's."""
    dev = torch.device("cuda", 0)
    n, nnz, d_max = synth.SHAPES["reddit"]
    h = 256
    rowptr, col = synth.make_csr(n, nnz, d_max, seed=0, device=dev, clustered=clustered)
    rp_h, col_h = rowptr.cpu().numpy().astype(np.uint32), col.cpu().numpy().astype(np.uint32)
    xr = synth.features(n, h, torch.float32, seed=1, device=dev, kind="uniform")
    x1 = synth.features(n, h, torch.float32, seed=0, device=dev)
    vals = (torch.rand(nnz, device=dev) * 2 - 1).float()
    for v in (None, vals):
        hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None if v is None else [v.data_ptr()],
                               [n], [n], [nnz], [1], [h], h)
        try:
            plan = _lib.group_lds_plan(hd)
            assert plan["tiles"] > 0 and plan["nnz"] == nnz, plan
            if clustered:
                assert plan["chunk_fills"] * 5 < plan["tiles"] * -(-n // _lib.group_lds_geometry(hd)["chunk_cols"])   # a tile streams only the chunks it touches
            got = run(hd, xr, n, h).cpu().numpy()
            ref = np.zeros((n, h), dtype=np.float32)
            oracle.spmm_csr_rowpar(rp_h, col_h, None if v is None else v.cpu().numpy(), xr.cpu().numpy(),
                                   nthreads=oracle.max_threads(), out=ref)
            assert got.tobytes() == ref.tobytes()
            if v is None:
                c_lds = run(hd, x1, n, h)
                old = _lib.set_tunable("lds_mode", 2)      # the same group, the sweep
                try:
                    c_sweep = run(hd, x1, n, h)
                finally:
                    _lib.set_tunable("lds_mode", old)
                assert torch.equal(c_lds, c_sweep)
        finally:
            _lib.group_free(hd)


@pytest.mark.parametrize("fail", [8, 12])
def test_products_scale_group_survives_a_failed_transient_allocation(fail):
    """creating a products-scale group needs gigabytes of transient device memory for the device code generator
    (sort keys, column tables).  lds_fail bit 8 makes its first large allocation "run out of memory" on the products-shaped community graph
    (2.4 M rows, 123.7 M entries, the density split's dense half); the rung below is the host encoder, which either cannot hold a part of this
    size (8: its slot headers overflow) or fails outright (8 | 4): the split is dropped and the part's own sweep serves the group, every step
    down named in pygim_group_lds_note.  Either way the product is right (column-count checksum + sampled rows against the oracle), nothing of
    the failed attempt stays allocated, and the next group is created normally."""
    dev = torch.device("cuda", 0)
    n, nnz, _ = synth.SHAPES["ogbn-products"]
    h = 128
    rowptr, col = synth.make_shape("ogbn-products", seed=0, device=dev, kind="sbm")
    x = synth.features(n, h, torch.int32, seed=0, device=dev)
    colcount = torch.bincount(col.long(), minlength=n).double()
    want_sum = colcount @ x.double()

    out = torch.empty((n, h), dtype=torch.int32, device=dev)

    def used():   # bytes in use on the device beyond torch's live tensors (its cache handed back first)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        free, total = torch.cuda.mem_get_info()
        return total - free - torch.cuda.memory_allocated()

    def create():
        return _lib.group_create(_lib.CSR, _lib.INT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [h], h)

    def product(hd, dst):
        _lib.spmm_run_group(hd, [x.data_ptr()], dst.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()

    base = used()
    old = _lib.set_tunable("lds_fail", fail)
    try:
        hd = create()
    finally:
        _lib.set_tunable("lds_fail", old)
    try:
        note = _lib.group_lds_note(hd)
        # (the host encoder is the rung below the device's -- but a dense half of this size does not fit its slot headers, so either way the
        # split is dropped, the part's own sweep serves the group and the note names every step down)
        assert "hybrid: no code-stream plan for the dense cells" in note and "sweep" in note and "device code generation not used" in note, note
        assert ("could not be built" in note) if fail == 12 else ("does not fit its slot headers" in note and "out of device memory" in note), note
        product(hd, out)
        assert torch.equal(out.double().sum(0), want_sum)
        sample_rows_vs_oracle(rowptr, col, None, x, out, [(0, 300), (n // 2, n // 2 + 300), (n - 300, n)])
        assert _lib.group_lds_runs(hd) == 0
    finally:
        _lib.group_free(hd)
    assert used() - base < (64 << 20), f"{(used() - base) >> 20} MiB still allocated after the group was freed"
    hd = create()   # undisturbed: the device writes the stream
    try:
        note = _lib.group_lds_note(hd)
        assert "density split" in note and "host encoder" not in note, note
        again = torch.empty_like(out)
        product(hd, again)
        assert torch.equal(again, out)
    finally:
        _lib.group_free(hd)
