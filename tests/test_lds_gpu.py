"""GPU parity of the LDS-staged product (k_lds_spmm_*: X chunks in LDS, a tile's running sums in registers, the
reference's scratchpad loop spmm_default/dpu_kernels/spmm_mul_csr_dpu.c:108-126) against the CPU oracle, through the C ABI.

Every row is summed by one wave in stored order, so FLOAT results must equal the oracle's sequential loop bit for bit
(BASELINE.json asks for 1e-5 relative; this path gives 0), integers are two's-complement modular.
"""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import coalesce, random_csr
from pygim_amd import _lib

pytestmark = pytest.mark.gpu
CODE = {np.dtype(np.float32): _lib.FLT32, np.dtype(np.int32): _lib.INT32}


@pytest.fixture(scope="module", autouse=True)
def backend():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.init_ranks(1)
    yield
    _lib.release()


@pytest.fixture()
def lds_forced():
    old = _lib.set_tunable("lds_mode", 1)
    yield
    _lib.set_tunable("lds_mode", old)
    _lib.set_tunable("lds_waves", 16)
    _lib.set_tunable("lds_long_slots", 128)
    _lib.set_tunable("lds_round_tiles", 1)
    _lib.set_tunable("lds_code", 1)
    for k in ("lds_code_waves", "lds_code_nbuf", "lds_code_kc", "lds_code_gsize", "lds_code_nsets", "lds_code_boundary"):
        _lib.set_tunable(k, 0)


def features(rng, n, h, dt):
    if dt == np.float32:
        return (rng.random((n, h), dtype=np.float32) * 2 - 1).astype(np.float32)   # sums round at every step
    return rng.integers(-2**31, 2**31 - 1, size=(n, h), dtype=np.int64).astype(np.int32)  # sums wrap


def product(rowptr, col, x, ncols=None, fmt="CSR", row=None, vals=None, want_plan=True):
    n = len(rowptr) - 1
    ncols = x.shape[0] if ncols is None else ncols
    h = x.shape[1]
    rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
    idx0 = rp if fmt == "CSR" else np.ascontiguousarray(row, np.int32)
    v = None if vals is None else [np.ascontiguousarray(vals, x.dtype).ctypes.data]
    hd = _lib.group_create(_lib.CSR if fmt == "CSR" else _lib.COO, CODE[x.dtype], [idx0.ctypes.data], [ci.ctypes.data], v,
                           [n], [ncols], [len(ci)], [1], [h], h)
    try:
        plan = _lib.group_lds_plan(hd)
        if want_plan is not None:
            assert (plan["tiles"] > 0) == want_plan, plan
        out = np.full((n, h), 77, dtype=x.dtype)
        xx = np.ascontiguousarray(x)
        _lib.spmm_run_group(hd, [xx.ctypes.data], out.ctypes.data)
    finally:
        _lib.group_free(hd)
    return out, plan


@pytest.mark.parametrize("waves", ["16-code", 16, 8, "16-long"])
@pytest.mark.parametrize("dt", [np.float32, np.int32])
def test_lds_product_is_bit_exact(rng, lds_forced, waves, dt):
    # "16-code": the schedule compiled into machine code (k_lds_code_*, the default form); 16 / 8 / "16-long": the token kernels
    # ("16-long": the 16-token-batch geometry the plan picks for long slots, forced here for every shape)
    old_long = _lib.set_tunable("lds_long_slots", 1 if waves == "16-long" else 0)
    old_code = _lib.set_tunable("lds_code", 1 if waves == "16-code" else 0)
    _lib.set_tunable("lds_waves", 16 if waves in ("16-long", "16-code") else waves)
    # (rows, cols, h, mean degree): widths around the 64-feature slice, ragged tiles, one and many chunks, a 5 000-entry row
    for n, ncols, h, avg in ((1, 1, 64, 1), (300, 700, 64, 12), (3000, 2500, 100, 12), (1700, 5000, 256, 11), (5000, 300, 65, 40),
                             (4000, 4000, 33, 30), (2000, 9000, 300, 25), (1500, 300, 64, 250)):   # (the last: hundreds of entries per row and chunk)
        rowptr, col = random_csr(rng, n, ncols, avg, long_rows=[(0, 5000)] if n > 100 else ())
        x = features(rng, ncols, h, dt)
        want = oracle.spmm_csr(rowptr, col, None, x)
        got, plan = product(rowptr, col, x)
        assert got.tobytes() == want.tobytes(), (waves, dt, n, ncols, h)
        assert plan["nnz"] == len(col) and plan["tokens"] >= len(col)
    _lib.set_tunable("lds_long_slots", old_long)
    _lib.set_tunable("lds_code", old_code)


def test_empty_rows_and_empty_matrix(rng, lds_forced):
    rowptr, col = random_csr(rng, 2000, 1500, 9, empty_frac=0.6)
    x = features(rng, 1500, 128, np.float32)
    got, _ = product(rowptr, col, x)
    assert got.tobytes() == oracle.spmm_csr(rowptr, col, None, x).tobytes()
    z = np.zeros(51, dtype=np.int64)
    got, plan = product(z, np.zeros(0, dtype=np.int64), features(rng, 9, 64, np.float32), want_plan=False)  # nothing stored: no plan
    assert not got.any()


def test_accumulate_strides_and_device_pointers(rng, lds_forced):
    """pygim_block_run: C += A.X into a wider matrix (ldc > width), X a window of a wider matrix (ldx > width)"""
    n, ncols, w = 3000, 2000, 96
    rowptr, col = random_csr(rng, n, ncols, 20)
    rp, ci = torch.from_numpy(rowptr.astype(np.int32)).cuda(), torch.from_numpy(col.astype(np.int32)).cuda()
    for dt, code in ((np.float32, _lib.FLT32), (np.int32, _lib.INT32)):
        xw = features(rng, ncols, w + 40, dt)
        c0 = features(rng, n, w + 24, dt)
        hd = _lib.group_create(_lib.CSR, code, [rp.data_ptr()], [ci.data_ptr()], None, [n], [ncols], [len(col)], [1], [w], w)
        try:
            assert _lib.group_lds_plan(hd)["tiles"] > 0
            xd, cd = torch.from_numpy(xw).cuda(), torch.from_numpy(c0).cuda()
            es = xd.element_size()
            _lib.block_run(hd, 0, xd.data_ptr() + 8 * es, w + 40, cd.data_ptr() + 16 * es, w + 24, w, accumulate=True)
            torch.cuda.synchronize()
            got = cd.cpu().numpy()
        finally:
            _lib.group_free(hd)
        prod = oracle.spmm_csr(rowptr, col, None, np.ascontiguousarray(xw[:, 8:8 + w]))
        want = c0.copy()
        if dt == np.float32:
            want[:, 16:16 + w] = c0[:, 16:16 + w] + prod     # the kernel adds its finished sum to what C held
        else:
            want[:, 16:16 + w] = (c0[:, 16:16 + w].astype(np.int64) + prod).astype(np.int32)
        assert got.tobytes() == want.tobytes(), dt


def test_coo_groups_and_merged_column_blocks(rng, lds_forced):
    """COO through the derived row pointers (unit weights), and sp_parts column blocks merged into one matrix"""
    n, ncols, h = 2500, 3000, 128
    rowptr, col = random_csr(rng, n, ncols, 15)
    r, c, v = coalesce(rowptr, col, np.float32)
    keep = np.ones(len(c), dtype=bool)  # a simple graph: all weights 1 after dropping duplicates
    rp2 = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rp2, r.astype(np.int64) + 1, 1)
    rp2 = np.cumsum(rp2)
    x = features(rng, ncols, h, np.float32)
    want = oracle.spmm_csr(rp2, c, None, x)
    got, _ = product(rp2, c, x, fmt="COO", row=r, vals=np.ones(len(c), np.float32))
    assert got.tobytes() == want.tobytes() and keep.all()
    # three column blocks with local ids (backend_pim/spmm.py:127-136): the group's merged matrix carries the plan
    import scipy.sparse as sp
    a = sp.csr_matrix((np.ones(len(c), np.float32), c.copy(), rp2.copy()), shape=(n, ncols))
    step = (ncols + 2) // 3
    blocks = [a[:, i * step:min(ncols, (i + 1) * step)].tocsr() for i in range(3)]
    for b in blocks:
        b.sort_indices()
    rps = [np.ascontiguousarray(b.indptr, np.int32) for b in blocks]
    cis = [np.ascontiguousarray(b.indices, np.int32) for b in blocks]
    hd = _lib.group_create(_lib.CSR, _lib.FLT32, [p.ctypes.data for p in rps], [q.ctypes.data for q in cis], None, [n] * 3,
                           [b.shape[1] for b in blocks], [b.nnz for b in blocks], [1] * 3, [h] * 3, h)
    try:
        assert _lib.group_lds_plan(hd)["tiles"] > 0 and _lib.group_plan(hd)["merged"] == 1
        out = np.empty((n, h), np.float32)
        _lib.spmm_run_group(hd, [x.ctypes.data], out.ctypes.data)
    finally:
        _lib.group_free(hd)
    assert out.tobytes() == want.tobytes()


@pytest.mark.parametrize("dt", [np.float32, np.int32])
def test_valued_entries_are_bit_exact(rng, lds_forced, dt):
    """acc += val * x in stored order, product and sum rounded separately (the valued device loop spmm_mul_csr_dpu.c:113;
    the oracle's spmm_grande/spmm_mul_csr.c:119-136): equal to the CPU loop bit for bit, floats too"""
    for n, ncols, h in ((3000, 2500, 100), (1700, 5000, 256), (20000, 20000, 64)):
        rowptr, col = random_csr(rng, n, ncols, 25, long_rows=[(1, 4000)])
        x = features(rng, ncols, h, dt)
        vals = features(rng, len(col), 1, dt)[:, 0]
        for code in (1, 0):   # FLT32: the code-stream form (the value is the literal of a v_mul_f32 in the stream), then the token kernel
            old_code = _lib.set_tunable("lds_code", code)
            got, plan = product(rowptr, col, x, vals=vals)
            _lib.set_tunable("lds_code", old_code)
            assert got.tobytes() == oracle.spmm_csr(rowptr, col, vals, x).tobytes(), (dt, n, h, code)
    _lib.set_tunable("lds_waves", 8)   # no valued kernel in the 8-wave geometry: the sweep answers (floats within the bound)
    got, plan = product(rowptr, col, x, vals=vals, want_plan=False)
    want = oracle.spmm_csr(rowptr, col, vals, x)
    assert np.array_equal(got, want) if dt == np.int32 else np.allclose(got, want, rtol=0, atol=1e-3)


def test_auto_rule_and_unsorted_rows_keep_the_sweep(rng):
    """lds_mode = 0 (default): planned only where a staged column of X serves enough stored entries; rows whose stored order
    is not column order never take it (and still give the oracle's result)"""
    assert _lib.set_tunable("lds_mode", 0) == 0
    xd = features(rng, 1000, 64, np.float32)
    rowptr, col = random_csr(rng, 4000, 1000, 200)         # ~2.9 stored entries per staged column of X and tile: planned
    got, plan = product(rowptr, col, xd, want_plan=True)
    assert got.tobytes() == oracle.spmm_csr(rowptr, col, None, xd).tobytes()
    x = features(rng, 3000, 64, np.float32)
    rowptr, col = random_csr(rng, 3000, 3000, 0.5)         # ~0.2: not worth staging
    got, plan = product(rowptr, col, x, want_plan=False)
    assert got.tobytes() == oracle.spmm_csr(rowptr, col, None, x).tobytes()
    rowptr, col = random_csr(rng, 3000, 3000, 20)
    col2 = col.copy()
    s, e = rowptr[5], rowptr[6]
    if e - s > 1:
        col2[s:e] = col2[s:e][::-1]                         # one row stored in descending column order
    got, plan = product(rowptr, col2, x, want_plan=False)
    assert got.tobytes() == oracle.spmm_csr(rowptr, col2, None, x).tobytes()


def test_same_features_reuse_the_slice_major_copy(rng, lds_forced):
    n, h = 3000, 128
    rowptr, col = random_csr(rng, n, n, 12)
    rp, ci = torch.from_numpy(rowptr.astype(np.int32)).cuda(), torch.from_numpy(col.astype(np.int32)).cuda()
    x = torch.from_numpy(features(rng, n, h, np.float32)).cuda()
    hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rp.data_ptr()], [ci.data_ptr()], None, [n], [n], [len(col)], [1], [h], h)
    try:
        a, b = torch.empty((n, h), device="cuda"), torch.empty((n, h), device="cuda")
        _lib.spmm_run_group(hd, [x.data_ptr()], a.data_ptr())
        _lib.spmm_run_group(hd, [x.data_ptr()], b.data_ptr(), x_unchanged=True)
        torch.cuda.synchronize()
    finally:
        _lib.group_free(hd)
    assert torch.equal(a, b) and a.cpu().numpy().tobytes() == oracle.spmm_csr(rowptr, col, None, x.cpu().numpy()).tobytes()


def test_shared_copy_is_not_reused_across_ring_geometries(rng, lds_forced):
    """the staged copy's slice stride depends on the part's chunk size (rows padded to whole chunks).  Two
    groups that share X with x_unchanged=True -- the first a code stream (128-column chunks), the second fallen to the token form
    (320-column chunks, lds_fail = 2) -- must not read each other's copy: slices 1.. would come back wrong (or the DMA run past
    the buffer).  ncols is chosen so that the two paddings differ."""
    n, ncols, h = 2500, 2000, 256          # 2000 -> 2048 rows per slice (kc = 128) against 2240 (kc = 320)
    rowptr, col = random_csr(rng, n, ncols, 40)
    x = features(rng, ncols, h, np.float32)
    want = oracle.spmm_csr(rowptr, col, None, x)
    rp, ci = torch.from_numpy(rowptr.astype(np.int32)).cuda(), torch.from_numpy(col.astype(np.int32)).cuda()
    xd = torch.from_numpy(x).cuda()
    mk = lambda: _lib.group_create(_lib.CSR, _lib.FLT32, [rp.data_ptr()], [ci.data_ptr()], None, [n], [ncols], [len(col)], [1], [h], h)
    h1 = mk()
    old = _lib.set_tunable("lds_fail", 2)
    try:
        h2 = mk()
    finally:
        _lib.set_tunable("lds_fail", old)
    try:
        g1, g2 = _lib.group_lds_geometry(h1), _lib.group_lds_geometry(h2)
        assert _lib.group_lds_code(h1)["active"] == 1 and _lib.group_lds_code(h2)["active"] == 0, (_lib.group_lds_note(h1), _lib.group_lds_note(h2))
        assert g1["chunk_cols"] != g2["chunk_cols"], (g1, g2)
        outs = [torch.full((n, h), 77.0, device="cuda") for _ in range(4)]
        for hd, o, same in ((h1, outs[0], False), (h2, outs[1], True), (h1, outs[2], True), (h2, outs[3], True)):
            _lib.spmm_run_group(hd, [xd.data_ptr()], o.data_ptr(), x_unchanged=same)
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            assert o.cpu().numpy().tobytes() == want.tobytes(), k
    finally:
        _lib.group_free(h1)
        _lib.group_free(h2)


@pytest.mark.parametrize("seed", range(max(12, int(os.environ.get("PYGIM_STRESS_SEEDS", "0")))))  # (a one-off soak: PYGIM_STRESS_SEEDS=500)
def test_random_shapes(seed, lds_forced):
    rng = np.random.default_rng(1000 + seed)
    _lib.set_tunable("lds_waves", int(rng.choice([8, 16])))
    _lib.set_tunable("lds_long_slots", int(rng.choice([0, 1, 128])))
    _lib.set_tunable("lds_round_tiles", int(rng.choice([0, 1])))
    _lib.set_tunable("lds_code", int(rng.choice([0, 1, 1])))
    dt = [np.float32, np.int32][seed % 2]
    n, ncols = int(rng.integers(1, 6000)), int(rng.integers(1, 6000))
    h = int(rng.integers(33, 320))
    rowptr, col = random_csr(rng, n, ncols, float(rng.uniform(0.5, 60)), empty_frac=float(rng.uniform(0, 0.5)),
                             long_rows=[(int(rng.integers(0, n)), int(rng.integers(1, 20000)))])
    x = features(rng, ncols, h, dt)
    got, _ = product(rowptr, col, x, want_plan=None)
    assert got.tobytes() == oracle.spmm_csr(rowptr, col, None, x).tobytes(), (seed, n, ncols, h)


@pytest.mark.parametrize("dt", ["INT8", "INT16", "INT32", "INT64", "FLT32", "DBL64"])
def test_all_six_types_with_the_lds_path_forced(rng, lds_forced, dt):
    """lds_mode = 1 with every val_dt (support/common.h:39-60), unit weights and values: the LDS-staged kernel wherever a plan exists,
    the sweep otherwise -- all equal the oracle (integers bit-exact; floats bit-exact here too: real-valued features, rows of <= 512 entries per
    panel for the sweep's types, empty rows, one long row)"""
    from conftest import NP_DTYPES

    npdt = NP_DTYPES[dt]
    code = {"INT8": _lib.INT8, "INT16": _lib.INT16, "INT32": _lib.INT32, "INT64": _lib.INT64, "FLT32": _lib.FLT32, "DBL64": _lib.DBL64}[dt]
    n, ncols, h = 2500, 1800, 96
    rowptr, col = random_csr(rng, n, ncols, 14, empty_frac=0.2, long_rows=[(9, 400)])
    if np.issubdtype(npdt, np.floating):
        x = (rng.random((ncols, h)) * 2 - 1).astype(npdt)
        vals = (rng.random(len(col)) * 2 - 1).astype(npdt)
    else:
        info = np.iinfo(npdt)
        x = rng.integers(info.min, info.max, size=(ncols, h), dtype=np.int64, endpoint=True).astype(npdt)
        vals = rng.integers(info.min, info.max, size=len(col), dtype=np.int64, endpoint=True).astype(npdt)
    rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
    for v in (None, vals):
        hd = _lib.group_create(_lib.CSR, code, [rp.ctypes.data], [ci.ctypes.data], None if v is None else [v.ctypes.data], [n], [ncols],
                               [len(ci)], [1], [h], h)
        try:
            # (round 4: unit-weight INT8 rides the INT16 code stream, its features widened to 16 bits in the staged copy)
            #  and unit-weight INT64 / DBL64 their own 8-byte code stream: rows of 512 bytes in LDS, a register pair per running sum)
            #  (round 5: every valued type too -- INT32 / INT16 / INT8 v_mul_lo_u32 / v_pk_mul_lo_u16 with the value inline or in an SGPR, INT64 / DBL64 through an SGPR pair)
            assert _lib.group_lds_plan(hd)["tiles"] > 0, dt
            out = np.full((n, h), 77, dtype=npdt)
            _lib.spmm_run_group(hd, [x.ctypes.data], out.ctypes.data)
        finally:
            _lib.group_free(hd)
        assert out.tobytes() == oracle.spmm_csr(rowptr, col, v, x).tobytes(), (dt, v is not None)


@pytest.mark.parametrize("dt", ["INT8", "INT16", "INT32", "INT64", "FLT32", "DBL64"])
def test_host_windows_store_straight_into_the_page_locked_result(rng, lds_forced, dt):
    """the host-operand pipeline's DIRECT mode (rt_run.inc run_group_windows, tunable host_direct = 2): each feature window's product is one pass of the
    LDS-staged kernel whose store stage writes the caller's page-locked tensor itself -- every val_dt, unit weights and values, column-split shares (the
    reduce kernel then does the writing) -- byte-equal to the oracle and to the staged-C pipeline; a result in pageable memory, or a group without an
    LDS-staged plan, keeps the copies (and says so)"""
    from conftest import NP_DTYPES

    npdt = NP_DTYPES[dt]
    es = np.dtype(npdt).itemsize
    code = {"INT8": _lib.INT8, "INT16": _lib.INT16, "INT32": _lib.INT32, "INT64": _lib.INT64, "FLT32": _lib.FLT32, "DBL64": _lib.DBL64}[dt]
    n, ncols = 2100, 1900
    h = 1024 // es if es < 8 else 192          # 4 slices of 256 bytes (8-byte types: 3 of 512)
    rowptr, col = random_csr(rng, n, ncols, 14, empty_frac=0.2, long_rows=[(9, 400)])
    if np.issubdtype(npdt, np.floating):
        x = (rng.random((ncols, h)) * 2 - 1).astype(npdt)
        vals = (rng.random(len(col)) * 2 - 1).astype(npdt)
    else:
        info = np.iinfo(npdt)
        x = rng.integers(info.min, info.max, size=(ncols, h), dtype=np.int64, endpoint=True).astype(npdt)
        vals = rng.integers(info.min, info.max, size=len(col), dtype=np.int64, endpoint=True).astype(npdt)
    rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
    tdt = torch.from_numpy(x).dtype
    old_d = _lib.set_tunable("host_direct", 2)
    try:
        for v, split in ((None, 0), (vals, 0), (None, 3)):
            if split and dt in ("INT64", "DBL64"):
                continue   # (8-byte plans have no column ranges)
            old_s = _lib.set_tunable("lds_col_split", split)
            old_f = _lib.set_tunable("lds_col_split_f32", 2 if split else 1)
            try:
                hd = _lib.group_create(_lib.CSR, code, [rp.ctypes.data], [ci.ctypes.data], None if v is None else [v.ctypes.data], [n], [ncols],
                                       [len(ci)], [1], [h], h)
            finally:
                _lib.set_tunable("lds_col_split", old_s)
                _lib.set_tunable("lds_col_split_f32", old_f)
            try:
                assert _lib.group_lds_plan(hd)["tiles"] > 0, dt
                ref = oracle.spmm_csr(rowptr, col, v, x)
                got = {}
                for hw in ((2, 3) if es == 8 else (2, 4)):   # (8-byte types: a window of one 32-feature slice is below the LDS kernel's width, lds_min_width8)
                    old_w = _lib.set_tunable("host_windows", hw)
                    try:
                        for direct in (2, 0):
                            _lib.set_tunable("host_direct", direct)
                            out = torch.full((n, h), 77, dtype=tdt, pin_memory=True)
                            runs = _lib.group_lds_runs(hd)
                            _lib.spmm_run_group(hd, [x.ctypes.data], out.data_ptr())
                            call = _lib.group_host_call(hd)
                            assert call["windows"] == min(hw, (h * es + 255) // 256) and call["direct"] == (1 if direct else 0), (dt, hw, direct, call)
                            assert _lib.group_lds_runs(hd) == runs + call["windows"]   # (every window took the LDS-staged kernel)
                            got[(hw, direct)] = out.numpy().copy()
                        _lib.set_tunable("host_direct", 2)
                        out = np.full((n, h), 77, dtype=npdt)   # pageable: the copies
                        _lib.spmm_run_group(hd, [x.ctypes.data], out.ctypes.data)
                        assert _lib.group_host_call(hd) == {"windows": min(hw, (h * es + 255) // 256), "direct": 0}
                        got[(hw, "pageable")] = out
                    finally:
                        _lib.set_tunable("host_windows", old_w)
                for key, o in got.items():
                    if split and np.issubdtype(npdt, np.floating):   # (a row's sum = the sum of its ranges' sums: the norm-wise contract)
                        assert o.tobytes() == got[(2, 0)].tobytes(), (dt, key)
                        scale = np.abs(oracle.spmm_csr(rowptr, col, None, np.abs(x)).astype(np.float64))
                        assert np.all(np.abs(o.astype(np.float64) - ref.astype(np.float64)) <= 1e-5 * scale + 1e-30)
                    else:
                        assert o.tobytes() == ref.tobytes(), (dt, v is not None, split, key)
            finally:
                _lib.group_free(hd)
        # a group the sweep serves: nothing to store directly
        old_m = _lib.set_tunable("lds_mode", 2)
        try:
            hd = _lib.group_create(_lib.CSR, code, [rp.ctypes.data], [ci.ctypes.data], None, [n], [ncols], [len(ci)], [1], [h], h)
        finally:
            _lib.set_tunable("lds_mode", old_m)
        try:
            old_w = _lib.set_tunable("host_windows", 2)
            out = torch.full((n, h), 77, dtype=tdt, pin_memory=True)
            _lib.spmm_run_group(hd, [x.ctypes.data], out.data_ptr())
            _lib.set_tunable("host_windows", old_w)
            assert _lib.group_host_call(hd) == {"windows": 2, "direct": 0}
            assert out.numpy().tobytes() == oracle.spmm_csr(rowptr, col, None, x).tobytes()
        finally:
            _lib.group_free(hd)
    finally:
        _lib.set_tunable("host_direct", old_d)


@pytest.mark.parametrize("dt,h", [(np.float32, 64), (np.int32, 64), (np.int16, 128), (np.float64, 64)])
def test_one_slice_products_read_the_callers_matrix_in_place(rng, lds_forced, dt, h):
    """a product of ONE slice whose rows of X are exactly the staged row (256 bytes; 512 for the 8-byte types) reads the caller's matrix instead of a slice-major copy
    (launch_lds, tunable lds_direct_x): same bytes as with the copy and as the oracle, for a row count that is no multiple of the chunk -- the last chunk's padding rows
    are read (never used) when they lie inside the caller's allocation, and the copy is made when they do not (X at the very end of its allocation)"""
    code = {np.dtype(np.float32): _lib.FLT32, np.dtype(np.int32): _lib.INT32, np.dtype(np.int16): _lib.INT16, np.dtype(np.float64): _lib.DBL64}[np.dtype(dt)]
    n, ncols = 3000, 2001
    rowptr, col = random_csr(rng, n, ncols, 20, empty_frac=0.1, long_rows=[(7, 1500)])
    rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
    x = (rng.random((ncols, h)) * 2 - 1).astype(dt) if np.issubdtype(dt, np.floating) else rng.integers(-1000, 1000, size=(ncols, h)).astype(dt)
    want = oracle.spmm_csr(rowptr, col, None, x)
    hd = _lib.group_create(_lib.CSR, code, [rp.ctypes.data], [ci.ctypes.data], None, [n], [ncols], [len(ci)], [1], [h], h)
    try:
        assert _lib.group_lds_plan(hd)["tiles"] > 0
        tdt = torch.from_numpy(x).dtype
        row_bytes = h * x.itemsize
        # (a) X somewhere inside a larger allocation; (b) X as the LAST rows of its allocation: the padding rows would lie outside
        big = torch.zeros((ncols + 4096) * row_bytes, dtype=torch.uint8, device="cuda")
        tail = torch.zeros(ncols * row_bytes + 256, dtype=torch.uint8, device="cuda")
        off = (-tail.data_ptr()) % 256
        views = [big[:ncols * row_bytes].view(tdt).view(ncols, h), tail[off:off + ncols * row_bytes].view(tdt).view(ncols, h)]
        for xv in views:
            xv.copy_(torch.from_numpy(x))
            for direct in (1, 0):
                old = _lib.set_tunable("lds_direct_x", direct)
                try:
                    out = torch.full((n, h), 77, dtype=tdt, device="cuda")
                    runs = _lib.group_lds_runs(hd)
                    _lib.spmm_run_group(hd, [xv.data_ptr()], out.data_ptr(), torch.cuda.current_stream().cuda_stream)
                    torch.cuda.synchronize()
                    assert _lib.group_lds_runs(hd) == runs + 1
                finally:
                    _lib.set_tunable("lds_direct_x", old)
                assert out.cpu().numpy().tobytes() == want.tobytes(), (np.dtype(dt).name, direct)
    finally:
        _lib.group_free(hd)


def test_denormals_and_infinities_follow_the_cpu_loop(rng, lds_forced):
    """the assembly inherits the kernel's float mode (denormals kept, round to nearest even): sums of subnormal features stay
    subnormal exactly as on the CPU, an infinite feature makes its rows infinite"""
    n, ncols, h = 1500, 1200, 64
    rowptr, col = random_csr(rng, n, ncols, 10)
    x = (rng.random((ncols, h), dtype=np.float32) * np.float32(3e-42)).astype(np.float32)   # all subnormal
    assert np.all(np.abs(x) < np.finfo(np.float32).tiny)
    x[7, :] = np.inf
    x[9, 3] = -np.inf
    got, _ = product(rowptr, col, x)
    want = oracle.spmm_csr(rowptr, col, None, x)
    both_nan = np.isnan(got) & np.isnan(want)        # inf - inf: a NaN on both sides (its payload is not part of the contract)
    assert np.array_equal(got.view(np.uint32)[~both_nan], want.view(np.uint32)[~both_nan])
    assert np.any((want != 0) & (np.abs(want) < np.finfo(np.float32).tiny)) and np.any(np.isinf(want))


def test_int16_two_features_to_a_lane(rng, lds_forced):
    """INT16 on the LDS-staged kernel: a lane holds two features (a 256-byte slice is 128 of them), packed 16-bit adds and
    multiplies wrap each half on its own -- modular at the element width, as support/common.h:39-60 asks; odd widths or rows of C
    that are not dword-aligned keep the sweep (and the same result)"""
    info = np.iinfo(np.int16)
    for n, ncols, h in ((3000, 2500, 100), (1700, 5000, 256), (2000, 3000, 66), (2500, 2000, 300), (2000, 1500, 101)):
        rowptr, col = random_csr(rng, n, ncols, 20, empty_frac=0.2, long_rows=[(1, 3000)])
        x = rng.integers(info.min, info.max, size=(ncols, h), endpoint=True).astype(np.int16)
        vals = rng.integers(info.min, info.max, size=len(col), endpoint=True).astype(np.int16)
        rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
        # the code-stream form (v_pk_add_u16 in the stream; round 5: valued too -- v_pk_mul_lo_u16) and the token kernels (lds_code = 0)
        for v, code in ((None, 1), (None, 0), (vals, 1), (vals, 0)):
            old_code = _lib.set_tunable("lds_code", code)
            hd = _lib.group_create(_lib.CSR, _lib.INT16, [rp.ctypes.data], [ci.ctypes.data], None if v is None else [v.ctypes.data],
                                   [n], [ncols], [len(ci)], [1], [h], h)
            try:
                assert _lib.group_lds_plan(hd)["tiles"] > 0
                assert _lib.group_lds_code(hd)["active"] == code
                out = np.full((n, h), 77, dtype=np.int16)
                _lib.spmm_run_group(hd, [x.ctypes.data], out.ctypes.data)
            finally:
                _lib.group_free(hd)
                _lib.set_tunable("lds_code", old_code)
            assert out.tobytes() == oracle.spmm_csr(rowptr, col, v, x).tobytes(), (n, h, v is not None, code)


@pytest.mark.parametrize("dt,code", [(np.int32, "INT32"), (np.float32, "FLT32"), (np.int8, "INT8"), (np.int16, "INT16")])
def test_quantised_aggregation_with_the_dequantising_store(rng, lds_forced, dt, code):
    """the conv layers' quantise -> aggregate -> dequantise (models/quantize.py:20-42, pyg_gcn_conv.py:130-137) in one device call on
    the LDS-staged kernel: the slice-major copy is written quantised, the kernel's store writes float(sum) * scale -- equal to the
    oracle's statement of the three steps bit for bit.  INT8 (round 4, the type models/quantize.py:22-23 quantises to): the quantised
    features are staged as 16-bit numbers, summed by the INT16 code stream, and the store sign-extends each sum's low byte = the
    modular int8 sum; INT16 likewise with 16-bit halves."""
    n, h = 3000, 256
    rowptr, col = random_csr(rng, n, n, 25, long_rows=[(5, 2500)])
    xf = rng.standard_normal((n, h)).astype(np.float32)
    rp, ci = torch.from_numpy(rowptr.astype(np.int32)).cuda(), torch.from_numpy(col.astype(np.int32)).cuda()
    xd = torch.from_numpy(xf).cuda()
    hd = _lib.group_create(_lib.CSR, getattr(_lib, code), [rp.data_ptr()], [ci.data_ptr()], None, [n], [n], [len(col)], [1], [h], h)
    try:
        assert _lib.group_lds_plan(hd)["tiles"] > 0
        out = torch.empty((n, h), dtype=torch.float32, device="cuda")
        scale = torch.empty(1, dtype=torch.float32, device="cuda")
        _lib.quant_spmm_run(hd, xd.data_ptr(), h, out.data_ptr(), scale.data_ptr())
        torch.cuda.synchronize()
        # the per-column epilogue of a GCN layer (bias + eval-mode BatchNorm as one affine map, then ReLU) rides the same store:
        # product and sum rounded separately, bit-identical to torch's a * out + b.  Widths that end inside a slice too.
        for hh, relu in ((h, True), (h, False)):
            a = torch.rand(hh, device="cuda") + 0.5
            b = torch.randn(hh, device="cuda")
            out2 = torch.full((n, hh), float("nan"), dtype=torch.float32, device="cuda")
            _lib.quant_spmm_run(hd, xd.data_ptr(), h, out2.data_ptr(), 0, 0, a.data_ptr(), b.data_ptr(), relu)
            torch.cuda.synchronize()
            want2 = a * out + b
            assert torch.equal(out2, torch.relu(want2) if relu else want2), (code, hh, relu)
    finally:
        _lib.group_free(hd)
    s_ref, xq = oracle.symmetric_quantize(xf, dt)
    want = oracle.symmetric_dequantize(oracle.spmm_csr(rowptr, col, None, xq), 1.0, s_ref)
    assert np.float32(scale.item()) == s_ref
    assert out.cpu().numpy().tobytes() == want.tobytes(), code


@pytest.mark.parametrize("code", [1, 0])
def test_short_row_shares_split_their_tiles_into_column_ranges(rng, lds_forced, code):
    """a rank's row share on N GPUs is a few tall tiles: too few workgroups to fill the chip when each streams a whole slice of X.
    The plan then splits every row tile into S column ranges (S x the workgroups, 1/S of X each), partial sums land in a scratch
    block and are added in range order.  INT32 / INT16: exact, automatic.  FLT32: automatic from a million entries up (round 6), any part with lds_col_split_f32 = 2, never
    with 0 -- and then a row's sum is the sum of its ranges' sequential sums (compared here on integer-valued features, where every order is exact, and
    on real ones against the norm-wise bound)."""
    old_code = _lib.set_tunable("lds_code", code)
    try:
        n, ncols, h = 2500, 30000, 256
        rowptr, col = random_csr(rng, n, ncols, 150, empty_frac=0.1, long_rows=[(3, 9000)])
        xi = features(rng, ncols, h, np.int32)
        got, plan = product(rowptr, col, xi)
        assert plan["tiles"] >= 4 * 2 and plan["tiles"] % 2 == 0          # 2 tall row tiles x S >= 4 column ranges
        assert got.tobytes() == oracle.spmm_csr(rowptr, col, None, xi).tobytes()
        # FLT32: a part of this size (375 k entries) is not split unless asked
        xf = features(rng, ncols, h, np.float32)
        got, plan1 = product(rowptr, col, xf)
        assert got.tobytes() == oracle.spmm_csr(rowptr, col, None, xf).tobytes()   # (bit-identical: whole rows, stored order)
        old = _lib.set_tunable("lds_col_split_f32", 2)
        try:
            got, plan2 = product(rowptr, col, xf)
            xint = rng.integers(-8, 8, size=(ncols, h)).astype(np.float32)
            got_i, _ = product(rowptr, col, xint)
        finally:
            _lib.set_tunable("lds_col_split_f32", old)
        assert plan2["tiles"] == plan["tiles"]
        assert np.array_equal(got_i, oracle.spmm_csr(rowptr, col, None, xint))
        want = oracle.spmm_csr(rowptr, col, None, xf)
        scale = oracle.spmm_csr(rowptr, col, None, np.abs(xf))
        assert np.all(np.abs(got.astype(np.float64) - want) <= 1e-5 * scale + 1e-30)
        # a forced split count, and never
        for s, tiles in ((3, 2 * 3), (1, None)):
            o = _lib.set_tunable("lds_col_split", s)
            try:
                got, pl = product(rowptr, col, xi)
            finally:
                _lib.set_tunable("lds_col_split", o)
            assert got.tobytes() == oracle.spmm_csr(rowptr, col, None, xi).tobytes()
            assert tiles is None or pl["tiles"] == tiles
    finally:
        _lib.set_tunable("lds_code", old_code)


# (waves, ring buffers, columns per chunk, staged columns per group, x-register sets) -- the knobs of the code-stream plan, round 4:
# 8 waves x 228 accumulators (k_lds_code8_*), rings of 3 / 4 / 5 buffers with the barrier in the middle of a slot, deeper read pipelines
# (+ where the workgroup meets when the ring has three or more buffers: 0 = at the slot boundary, the default; 2 = in the middle of a slot)
CODE_GEOS = [(16, 2, 0, 0, 0, 0), (16, 3, 0, 0, 0, 0), (16, 3, 192, 6, 3, 2), (8, 2, 0, 0, 0, 0), (8, 3, 0, 0, 0, 2), (8, 4, 0, 0, 0, 0), (8, 4, 160, 6, 3, 2),
             (8, 5, 0, 0, 0, 0), (8, 5, 0, 0, 0, 2), (8, 6, 0, 0, 0, 0), (8, 3, 64, 2, 2, 0), (16, 4, 128, 8, 2, 2), (8, 10, 64, 4, 3, 0)]


@pytest.mark.parametrize("geo", CODE_GEOS)
def test_code_stream_geometries_are_bit_exact(rng, lds_forced, geo):
    """tests/test_lds_plan.py runs these plans through the CPU interpreter; here the same plans run on the GPU (k_lds_code_* / k_lds_code8_*)
    against the oracle's loop: FLT32 bit-identical (every row summed by one wave in stored order, whatever the geometry), INT32 modular,
    INT16 two features to a lane, and the valued FLT32 form"""
    for k, v in zip(("lds_code_waves", "lds_code_nbuf", "lds_code_kc", "lds_code_gsize", "lds_code_nsets", "lds_code_boundary"), geo):
        _lib.set_tunable(k, v)
    for dt in (np.float32, np.int32):
        for n, ncols, h, avg in ((1, 1, 64, 1), (300, 700, 64, 12), (3000, 2500, 100, 12), (1700, 5000, 256, 11), (5000, 300, 65, 40),
                                 (2000, 9000, 300, 25), (1500, 300, 64, 250), (6000, 6000, 128, 60)):
            rowptr, col = random_csr(rng, n, ncols, avg, long_rows=[(0, 5000)] if n > 100 else ())
            x = features(rng, ncols, h, dt)
            got, plan = product(rowptr, col, x, want_plan=len(col) > 0)
            assert got.tobytes() == oracle.spmm_csr(rowptr, col, None, x).tobytes(), (geo, dt, n, ncols, h)
    # the geometry the library reports is the one that was asked for
    rowptr, col = random_csr(rng, 3000, 4000, 30)
    x = features(rng, 4000, 128, np.float32)
    rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
    hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rp.ctypes.data], [ci.ctypes.data], None, [3000], [4000], [len(ci)], [1], [128], 128)
    try:
        g = _lib.group_lds_geometry(hd)
        assert g["waves"] == geo[0] and g["buffers"] == geo[1] and g["acc_per_wave"] == (228 if geo[0] == 8 else 96), g
        assert g["chunk_cols"] * 256 * g["buffers"] <= 163840 and g["chunk_cols"] % (4 * geo[0]) == 0
        assert _lib.group_lds_code(hd)["active"] == 1
    finally:
        _lib.group_free(hd)
    # valued FLT32: the value is the literal of a v_mul_f32 in front of the add
    rowptr, col = random_csr(rng, 2500, 3000, 20, long_rows=[(5, 2800)])
    x = features(rng, 3000, 96, np.float32)
    vals = (rng.random(len(col), dtype=np.float32) * 2 - 1).astype(np.float32)
    got, _ = product(rowptr, col, x, vals=vals)
    assert got.tobytes() == oracle.spmm_csr(rowptr, col, vals, x).tobytes(), geo


@pytest.mark.parametrize("fail,want", [(1, "token"), (2, "token"), (4, "sweep")])
def test_fallback_ladder_says_what_it_did(rng, lds_forced, fail, want):
    """when the code-stream form cannot be set up (its generation fails: 1; no executable memory: 2) the group takes
    the TOKEN form of the same product -- not the sweep -- and when the schedule itself cannot be built (4) the sweep; every step down is
    recorded as text with the group (pygim_group_lds_note), nothing falls back silently, and the product is the oracle's either way"""
    rowptr, col = random_csr(rng, 3000, 2500, 20, long_rows=[(0, 2400)])
    x = features(rng, 2500, 128, np.float32)
    want_c = oracle.spmm_csr(rowptr, col, None, x)
    rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
    old = _lib.set_tunable("lds_fail", fail)
    try:
        hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rp.ctypes.data], [ci.ctypes.data], None, [3000], [2500], [len(ci)], [1], [128], 128)
    finally:
        _lib.set_tunable("lds_fail", old)
    try:
        note, plan, code = _lib.group_lds_note(hd), _lib.group_lds_plan(hd), _lib.group_lds_code(hd)
        assert code["active"] == 0 and code["code_bytes"] == 0
        if want == "token":
            assert plan["tiles"] > 0 and "code-stream form not available" in note and "token form" in note, note
            assert ("lds_fail" in note) and (("generation" in note) if fail == 1 else ("executable" in note)), note
            assert _lib.group_lds_geometry(hd)["waves"] == 16
        else:
            assert plan["tiles"] == 0 and "could not be built" in note and "sweep" in note, note
        out = np.full((3000, 128), 77, dtype=np.float32)
        _lib.spmm_run_group(hd, [x.ctypes.data], out.ctypes.data)
        if want == "token":
            assert out.tobytes() == want_c.tobytes()          # the token kernels sum in stored order too
        else:
            bound = oracle.spmm_csr(rowptr, col, None, np.abs(x))
            assert np.all(np.abs(out.astype(np.float64) - want_c) <= 1e-5 * bound + 1e-30)
    finally:
        _lib.group_free(hd)
    # ... and the undisturbed group says it is a code stream
    hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rp.ctypes.data], [ci.ctypes.data], None, [3000], [2500], [len(ci)], [1], [128], 128)
    try:
        assert _lib.group_lds_note(hd) == "code-stream form" and _lib.group_lds_code(hd)["active"] == 1
    finally:
        _lib.group_free(hd)


def test_int8_rides_the_int16_code_stream(rng, lds_forced):
    """INT8, unit weights (round 4): features widened to 16 bits in the slice-major copy (128 to a slice), packed 16-bit sums in the
    8-wave code stream, the store keeps each sum's low byte -- the two's-complement modular int8 sum of the oracle's loop, bit for bit;
    widths that end inside a lane or a slice (odd widths too), rows of C at any byte stride"""
    for n, ncols, h, avg in ((2000, 1500, 256, 30), (1000, 3000, 100, 12), (1500, 900, 101, 40), (700, 700, 255, 9), (1200, 5000, 129, 25), (600, 600, 67, 200)):
        rowptr, col = random_csr(rng, n, ncols, avg, empty_frac=0.1, long_rows=[(3, min(ncols, 2000))])
        x = rng.integers(-128, 127, size=(ncols, h), dtype=np.int64, endpoint=True).astype(np.int8)
        rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
        hd = _lib.group_create(_lib.CSR, _lib.INT8, [rp.ctypes.data], [ci.ctypes.data], None, [n], [ncols], [len(ci)], [1], [h], h)
        try:
            assert _lib.group_lds_plan(hd)["tiles"] > 0 and _lib.group_lds_code(hd)["active"] == 1 and _lib.group_lds_geometry(hd)["waves"] == 8
            out = np.full((n, h), 77, dtype=np.int8)
            _lib.spmm_run_group(hd, [x.ctypes.data], out.ctypes.data)
        finally:
            _lib.group_free(hd)
        assert out.tobytes() == oracle.spmm_csr(rowptr, col, None, x).tobytes(), (n, ncols, h)


@pytest.mark.parametrize("dt", [np.float64, np.int64])
def test_8_byte_elements_on_their_code_stream(rng, lds_forced, dt):
    """INT64 / DBL64, unit weights (round 4; support/common.h:39-60 makes all six val_dt first-class): slices of 64 features = 512-byte
    rows in LDS, one ds_read_b64 per staged column, a register pair per running sum (8 waves x 114 rows), v_add_f64 or a 64-bit
    integer add.  DBL64 sums are bit-identical to the oracle's loop (stored order), INT64 modular; widths that end inside a slice,
    C += A.X, ring geometries"""
    code = _lib.DBL64 if dt == np.float64 else _lib.INT64
    for geo in ((0, 0, 0, 0), (4, 64, 5, 2), (6, 48, 3, 3), (3, 96, 2, 2)):
        for k, v in zip(("lds_code_nbuf", "lds_code_kc", "lds_code_gsize", "lds_code_nsets"), geo):
            _lib.set_tunable(k, v)
        for n, ncols, h, avg in ((1, 1, 64, 1), (2000, 1500, 256, 30), (1000, 3000, 100, 12), (1500, 900, 65, 40), (900, 5000, 33, 25), (600, 600, 64, 200)):
            rowptr, col = random_csr(rng, n, ncols, avg, empty_frac=0.1, long_rows=[(0, min(ncols, 2500))] if n > 100 else ())
            if dt == np.float64:
                x = rng.random((ncols, h)) * 2 - 1
            else:
                x = rng.integers(-2**63, 2**63 - 1, size=(ncols, h), dtype=np.int64)
            rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
            hd = _lib.group_create(_lib.CSR, code, [rp.ctypes.data], [ci.ctypes.data], None, [n], [ncols], [len(ci)], [1], [h], h)
            try:
                if len(col):
                    g = _lib.group_lds_geometry(hd)
                    assert _lib.group_lds_code(hd)["active"] == 1 and g["waves"] == 8 and g["acc_per_wave"] == 114 and g["chunk_cols"] * 512 * g["buffers"] <= 163840, g
                out = np.full((n, h), 77, dtype=dt)
                _lib.spmm_run_group(hd, [x.ctypes.data], out.ctypes.data)
                want = oracle.spmm_csr(rowptr, col, None, x)
                assert out.tobytes() == want.tobytes(), (dt, geo, n, ncols, h)
                if n > 100:   # C += A.X into a wider matrix (pygim_block_run's accumulate form; device pointers)
                    before = np.ascontiguousarray(rng.integers(-5, 5, size=(n, h + 7)).astype(dt))
                    wide_d, x_d = torch.from_numpy(before.copy()).cuda(), torch.from_numpy(np.ascontiguousarray(x)).cuda()
                    _lib.block_run(hd, 0, x_d.data_ptr(), h, wide_d.data_ptr(), h + 7, h, True)
                    torch.cuda.synchronize()
                    wide = wide_d.cpu().numpy()
                    if dt == np.float64:
                        assert np.array_equal(wide[:, :h], before[:, :h] + want) and np.array_equal(wide[:, h:], before[:, h:])
                    else:
                        assert np.array_equal(wide[:, :h], (before[:, :h].astype(np.uint64) + want.astype(np.uint64)).astype(np.int64)) and np.array_equal(wide[:, h:], before[:, h:])
            finally:
                _lib.group_free(hd)


def test_the_note_is_truncated_to_the_callers_buffer(rng, lds_forced):
    """pygim_group_lds_note writes at most cap - 1 characters and a terminating NUL"""
    import ctypes

    rowptr, col = random_csr(rng, 500, 400, 10)
    rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
    hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rp.ctypes.data], [ci.ctypes.data], None, [500], [400], [len(ci)], [1], [64], 64)
    try:
        full = _lib.group_lds_note(hd)
        assert full == "code-stream form"
        buf = (ctypes.c_char * 8)(*([b"\xff"] * 8))
        assert _lib.lib().pygim_group_lds_note(ctypes.c_int64(hd), buf, ctypes.c_int64(5)) == 0
        assert buf.raw[:5] == b"code\x00" and buf.raw[5:] == b"\xff" * 3
    finally:
        _lib.group_free(hd)


@pytest.mark.parametrize("verify", [0, 2])
def test_a_share_that_all_but_fits_one_workgroup_per_cu_leaves_its_last_rows_to_the_tail_kernel(rng, verify):
    """Round 6: 16 full-height row tiles x 4 slices x 4 column ranges are exactly the 256 workgroups of the chip; a share of 16 x 1 824 + 200 rows would
    need 17 tiles and then only 3 ranges fit (204 workgroups, each streaming a third of X).  The plan takes the 29 184 rows that fit, k_lds_tail the last
    200 from the same staged copy.  INT32 exact against the oracle, FLT32 inside the path's 1e-5 (column-split sums), a ragged width, accumulation into C,
    and the conv layers' quantised aggregation (the dequantisation where the ranges' partial sums meet) -- all on the LDS-staged kernel (lds_runs says so);
    verify = 2: the device-written code stream compared word for word with the host encoder's for the plan that sees fewer rows."""
    n, ncols = 16 * 1824 + 200, 40000
    rowptr, col = random_csr(rng, n, ncols, 40, empty_frac=0.05, long_rows=[(n - 7, 3000), (11, 5000), (n - 150, 0)])
    rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
    old = _lib.set_tunable("lds_col_split_f32", 2), _lib.set_tunable("lds_codegen", verify if verify else 1)
    try:
        for dt, code, h in ((np.int32, _lib.INT32, 256), (np.float32, _lib.FLT32, 256), (np.int32, _lib.INT32, 200)):
            hd = _lib.group_create(_lib.CSR, code, [rp.ctypes.data], [ci.ctypes.data], None, [n], [ncols], [len(ci)], [1], [h], h)
            try:
                geo, note = _lib.group_lds_geometry(hd), _lib.group_lds_note(hd)
                assert geo["col_splits"] == 4 and "k_lds_tail" in note and str(16 * 1824) in note, (geo, note)
                assert _lib.group_lds_plan(hd)["tiles"] == 16 * 4
                x = features(rng, ncols, h, dt) if dt == np.float32 else rng.integers(-1000, 1000, size=(ncols, h)).astype(np.int32)
                out = np.full((n, h), 77, dtype=dt)
                _lib.spmm_run_group(hd, [np.ascontiguousarray(x).ctypes.data], out.ctypes.data)
                want = oracle.spmm_csr(rowptr, col, None, x)
                if dt == np.int32:
                    assert np.array_equal(out, want), h
                    # accumulate into C through the block entry point: what was there + the product
                    base = rng.integers(-50, 50, size=(n, h)).astype(np.int32)
                    acc = base.copy()
                    xd_, accd_ = torch.from_numpy(x).cuda(), torch.from_numpy(acc).cuda()
                    _lib.block_run(hd, 0, xd_.data_ptr(), h, accd_.data_ptr(), h, h, True)
                    torch.cuda.synchronize()
                    acc = accd_.cpu().numpy()
                    assert np.array_equal(acc, (base.astype(np.int64) + want.astype(np.int64)).astype(np.int32)), h
                else:
                    bound = oracle.spmm_csr(rowptr, col, None, np.abs(x)).astype(np.float64)
                    assert np.all(np.abs(out.astype(np.float64) - want.astype(np.float64)) <= 1e-5 * bound + 1e-30)
                    assert np.array_equal(out[-200:][:, :8], out[-200:][:, :8])   # (finite)
                runs = _lib.group_lds_runs(hd)
                assert runs >= 1
                if h == 256:   # quantise -> aggregate -> dequantise in one device call
                    xf = (rng.standard_normal((ncols, h)) * 3).astype(np.float32)
                    xd, od, sd = torch.from_numpy(xf).cuda(), torch.empty((n, h), dtype=torch.float32, device="cuda"), torch.empty((), dtype=torch.float32, device="cuda")
                    _lib.quant_spmm_run(hd, xd.data_ptr(), h, od.data_ptr(), sd.data_ptr(), 0)
                    torch.cuda.synchronize()
                    assert _lib.group_lds_runs(hd) == runs + 1, "the quantised aggregation of a column-split share left the LDS-staged kernel"
                    s_ref, xq = oracle.symmetric_quantize(xf, dt)
                    assert np.float32(sd.item()) == s_ref
                    want_q = oracle.spmm_csr(rowptr, col, None, xq)
                    want_f = oracle.symmetric_dequantize(want_q, 1.0, s_ref)
                    got = od.cpu().numpy()
                    if dt == np.int32:
                        assert got.tobytes() == want_f.tobytes()
                    else:
                        bq = oracle.spmm_csr(rowptr, col, None, np.abs(xq)).astype(np.float64) * float(s_ref)
                        assert np.all(np.abs(got.astype(np.float64) - want_f.astype(np.float64)) <= 1e-5 * bq + 1e-30)
            finally:
                _lib.group_free(hd)
        # a share whose last rows hold no entries at all: no segment, no gather launch -- the rows are written as zeros all the same
        rowptr0, col0 = random_csr(rng, n, ncols, 40, long_rows=[(r, 0) for r in range(n - 200, n)])
        rp0, ci0 = np.ascontiguousarray(rowptr0, np.int32), np.ascontiguousarray(col0, np.int32)
        hd = _lib.group_create(_lib.CSR, _lib.INT32, [rp0.ctypes.data], [ci0.ctypes.data], None, [n], [ncols], [len(ci0)], [1], [128], 128)
        try:
            assert "k_lds_tail" in _lib.group_lds_note(hd)
            x = rng.integers(-1000, 1000, size=(ncols, 128)).astype(np.int32)
            out = np.full((n, 128), 77, dtype=np.int32)
            _lib.spmm_run_group(hd, [x.ctypes.data], out.ctypes.data)
            assert np.array_equal(out, oracle.spmm_csr(rowptr0, col0, None, x)) and not out[-200:].any()
        finally:
            _lib.group_free(hd)
        # never: the share keeps three ranges and all its rows inside the plan
        _lib.set_tunable("lds_row_tail", 0)
        hd = _lib.group_create(_lib.CSR, _lib.INT32, [rp.ctypes.data], [ci.ctypes.data], None, [n], [ncols], [len(ci)], [1], [256], 256)
        try:
            assert _lib.group_lds_geometry(hd)["col_splits"] == 3 and "k_lds_tail" not in _lib.group_lds_note(hd)
        finally:
            _lib.group_free(hd)
    finally:
        _lib.set_tunable("lds_row_tail", 3)
        _lib.set_tunable("lds_col_split_f32", old[0])
        _lib.set_tunable("lds_codegen", old[1])


@pytest.mark.parametrize("name,npdt,code", [("INT8", np.int8, 0), ("INT16", np.int16, 1)])
def test_column_split_shares_of_the_16_bit_streams(rng, name, npdt, code):
    """Round 6: a short row share of an INT8 / INT16 graph is split into column ranges too.  The ranges' partial sums are the INT16 stream's packed 16-bit
    numbers; k_lds_reduce16 adds them (modulo 2^16) and makes the result: the int8 sum's byte (plain product, also accumulating into C), or -- the conv layers'
    quantised aggregation, models/pyg_gcn_conv.py:130-137 -- float(sum) * scale.  All exact against the oracle; lds_runs says the LDS-staged kernel ran."""
    n, ncols, h = 2500, 30000, 256
    rowptr, col = random_csr(rng, n, ncols, 150, empty_frac=0.1, long_rows=[(3, 9000)])
    rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
    assert code == (_lib.INT8 if name == "INT8" else _lib.INT16)
    hd = _lib.group_create(_lib.CSR, code, [rp.ctypes.data], [ci.ctypes.data], None, [n], [ncols], [len(ci)], [1], [h], h)
    try:
        geo = _lib.group_lds_geometry(hd)
        assert geo["col_splits"] >= 4 and _lib.group_lds_code(hd)["active"] == 1, (geo, _lib.group_lds_note(hd))
        lim = 127 if name == "INT8" else 30000
        x = rng.integers(-lim, lim, size=(ncols, h)).astype(npdt)      # sums wrap
        want = oracle.spmm_csr(rowptr, col, None, x)
        out = np.full((n, h), 77, dtype=npdt)
        _lib.spmm_run_group(hd, [x.ctypes.data], out.ctypes.data)
        assert np.array_equal(out, want)
        runs = _lib.group_lds_runs(hd)
        assert runs >= 1, _lib.group_lds_note(hd)
        if name == "INT8":   # what was in C + the product
            base = rng.integers(-100, 100, size=(n, h)).astype(npdt)
            xd, cd = torch.from_numpy(x).cuda(), torch.from_numpy(base.copy()).cuda()
            _lib.block_run(hd, 0, xd.data_ptr(), h, cd.data_ptr(), h, h, True)
            torch.cuda.synchronize()
            assert np.array_equal(cd.cpu().numpy(), (base.astype(np.int64) + want.astype(np.int64)).astype(npdt))
            assert _lib.group_lds_runs(hd) == runs + 1
            runs += 1
        xf = (rng.standard_normal((ncols, h)) * 3).astype(np.float32)
        xd, od, sd = torch.from_numpy(xf).cuda(), torch.empty((n, h), dtype=torch.float32, device="cuda"), torch.empty((), dtype=torch.float32, device="cuda")
        _lib.quant_spmm_run(hd, xd.data_ptr(), h, od.data_ptr(), sd.data_ptr(), 0)
        torch.cuda.synchronize()
        assert _lib.group_lds_runs(hd) == runs + 1, "the quantised aggregation of a column-split share left the LDS-staged kernel"
        s_ref, xq = oracle.symmetric_quantize(xf, npdt)
        want_f = oracle.symmetric_dequantize(oracle.spmm_csr(rowptr, col, None, xq), 1.0, s_ref)
        assert np.float32(sd.item()) == s_ref and od.cpu().numpy().tobytes() == want_f.tobytes()
    finally:
        _lib.group_free(hd)


@pytest.mark.parametrize("h", [32, 17, 24])
def test_half_split_plans_for_products_of_at_most_32_lanes(rng, h):
    """Round 6: a product of 17..32 lanes (all rows x 32 FLT32 features: a feature-split rank) would leave half of every 64-lane slice empty.  The half-split plan
    folds TWO column ranges into the halves of a wave instead -- a staged row is [X[c] | X[c + H]], the stream adds under the lower / upper half of EXEC, the store
    adds the halves -- so every tile stages half the bytes.  INT32 exact against the oracle (odd column count, empty rows, a long row, accumulation into C, a ragged
    width), FLT32 inside 1e-5 of |A|.|x|; lds_half_split = 0 keeps the plain plan with the same results.  The plan is written by the DEVICE code generator (the
    default since the round's last session) and compared word for word with the host encoder's inside the library (lds_codegen = 2); lds_codegen = 0 takes the host
    encoder alone."""
    n, ncols = 5000, 30001
    rowptr, col = random_csr(rng, n, ncols, 60, empty_frac=0.1, long_rows=[(3, 9000), (n - 1, 4000)])
    rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
    old = _lib.set_tunable("lds_col_split_f32", 2)
    old_split = _lib.set_tunable("lds_col_split", 1 if h == 24 else 0)   # (h = 24: whole-X tiles -- the kernel's own store adds the halves, and adds into C)
    old_mode = _lib.set_tunable("lds_mode", 1 if h == 24 else 0)         # (... which the reuse rule would not plan for a matrix this small)
    try:
        for dt, code in ((np.int32, _lib.INT32), (np.float32, _lib.FLT32)):
            outs = {}
            for hs, cg in ((1, 2), (1, 0), (0, 2)):
                _lib.set_tunable("lds_half_split", hs)
                old_cg = _lib.set_tunable("lds_codegen", cg)
                try:
                    hd = _lib.group_create(_lib.CSR, code, [rp.ctypes.data], [ci.ctypes.data], None, [n], [ncols], [len(ci)], [1], [h], h)
                finally:
                    _lib.set_tunable("lds_codegen", old_cg)
                try:
                    note = _lib.group_lds_note(hd)
                    assert ("half-split" in note) == bool(hs) and _lib.group_lds_code(hd)["active"] == 1, note
                    assert _lib.group_lds_code(hd)["device_generated"] == (1 if cg else 0), note
                    x = features(rng, ncols, h, dt) if dt == np.float32 else rng.integers(-1000, 1000, size=(ncols, h)).astype(np.int32)
                    if "x" not in outs:
                        outs["x"] = x
                    x = outs["x"]
                    out = np.full((n, h), 77, dtype=dt)
                    _lib.spmm_run_group(hd, [np.ascontiguousarray(x).ctypes.data], out.ctypes.data)
                    assert _lib.group_lds_runs(hd) == 1, note
                    want = oracle.spmm_csr(rowptr, col, None, x)
                    if dt == np.int32:
                        assert np.array_equal(out, want), (h, hs)
                        base = rng.integers(-50, 50, size=(n, h)).astype(np.int32)
                        xd_, accd_ = torch.from_numpy(x).cuda(), torch.from_numpy(base.copy()).cuda()
                        _lib.block_run(hd, 0, xd_.data_ptr(), h, accd_.data_ptr(), h, h, True)
                        torch.cuda.synchronize()
                        assert np.array_equal(accd_.cpu().numpy(), (base.astype(np.int64) + want.astype(np.int64)).astype(np.int32)), (h, hs)
                    else:
                        bound = oracle.spmm_csr(rowptr, col, None, np.abs(x)).astype(np.float64)
                        assert np.all(np.abs(out.astype(np.float64) - want.astype(np.float64)) <= 1e-5 * bound + 1e-30), (h, hs)
                finally:
                    _lib.group_free(hd)
    finally:
        _lib.set_tunable("lds_half_split", 1)
        _lib.set_tunable("lds_col_split", old_split)
        _lib.set_tunable("lds_mode", old_mode)
        _lib.set_tunable("lds_col_split_f32", old)


@pytest.mark.parametrize("seed", range(6))
def test_random_shares_around_the_tile_multiples(seed):
    """Round 6, randomised: row counts just above a whole number of full-height tiles (where the plan leaves the last rows to the tail kernels), random degrees (long rows in
    the tail included), 256 / 192 / 64 features, INT32 exact and FLT32 inside 1e-5 of |A|.|x| against the oracle; every code stream also compared word for word with the
    host encoder's (lds_codegen = 2)."""
    r = np.random.default_rng(9000 + seed)
    h = int(r.choice([256, 192, 64]))
    nsl = (h + 63) // 64
    S2 = int(r.choice([8, 4])) if nsl > 1 else 8
    tall2 = 256 // (nsl * S2)
    n = tall2 * 1824 + int(r.integers(1, max(2, int(0.028 * tall2 * 1824))))
    ncols = int(r.integers(6000, 20000))
    deg = float(r.uniform(4, 12)) if nsl == 1 else float(r.uniform(8, 40))
    longs = [(int(n - 1 - r.integers(0, 20)), int(r.integers(1500, 5000))), (int(r.integers(0, 100)), 3000)]
    rowptr, col = random_csr(r, n, ncols, deg, empty_frac=0.1, long_rows=longs)
    rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
    old = _lib.set_tunable("lds_col_split_f32", 2), _lib.set_tunable("lds_codegen", 2), _lib.set_tunable("lds_mode", 1)
    try:
        for dt, code in ((np.int32, _lib.INT32), (np.float32, _lib.FLT32)):
            hd = _lib.group_create(_lib.CSR, code, [rp.ctypes.data], [ci.ctypes.data], None, [n], [ncols], [len(ci)], [1], [h], h)
            try:
                note = _lib.group_lds_note(hd)
                assert "k_lds_tail" in note, (n, h, note)
                x = features(r, ncols, h, dt) if dt == np.float32 else r.integers(-10000, 10000, size=(ncols, h)).astype(np.int32)
                out = np.full((n, h), 77, dtype=dt)
                _lib.spmm_run_group(hd, [np.ascontiguousarray(x).ctypes.data], out.ctypes.data)
                assert _lib.group_lds_runs(hd) == 1
                want = oracle.spmm_csr(rowptr, col, None, x)
                if dt == np.int32:
                    assert np.array_equal(out, want), (seed, n, h)
                else:
                    bound = oracle.spmm_csr(rowptr, col, None, np.abs(x)).astype(np.float64)
                    assert np.all(np.abs(out.astype(np.float64) - want.astype(np.float64)) <= 1e-5 * bound + 1e-30), (seed, n, h)
            finally:
                _lib.group_free(hd)
    finally:
        _lib.set_tunable("lds_col_split_f32", old[0])
        _lib.set_tunable("lds_codegen", old[1])
        _lib.set_tunable("lds_mode", old[2])
