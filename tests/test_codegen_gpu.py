"""GPU: the code stream GENERATED ON THE DEVICE (pygim_amd/csrc/lds_codegen_dev.hpp, round 5) against the host encoder
(lds_plan.hpp lds_code_from_plan), word for word, and against the oracle.

``lds_codegen = 2`` makes ``pygim_group_create`` build the stream both ways and compare the instruction words, the stream offsets, the row
map, the tile table and the statistics inside the library (codegen_verify); a difference is an error of the call.  The CPU form of the same
comparison (the bodies run as plain loops) is tests/test_lds_plan.py::test_data_parallel_encoder_equals_host_encoder.
Reference cost being matched: the reference's one-time step is a partition walk and a copy (spmm_default/spmm_mul_csr.c:118-330).
"""
import time

import numpy as np
import pytest
import torch

import oracle
from conftest import random_csr
from pygim_amd import _lib, synth

pytestmark = pytest.mark.gpu
CODE = {np.dtype(np.float32): _lib.FLT32, np.dtype(np.int32): _lib.INT32, np.dtype(np.int16): _lib.INT16, np.dtype(np.int8): _lib.INT8,
        np.dtype(np.float64): _lib.DBL64, np.dtype(np.int64): _lib.INT64}
GEO_KNOBS = ("lds_code_waves", "lds_code_nbuf", "lds_code_kc", "lds_code_gsize", "lds_code_nsets", "lds_code_boundary")


@pytest.fixture(scope="module", autouse=True)
def backend():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.init_ranks(1)
    yield
    _lib.release()


@pytest.fixture()
def checked():
    old = _lib.set_tunable("lds_mode", 1), _lib.set_tunable("lds_codegen", 2)
    yield
    _lib.set_tunable("lds_mode", old[0])
    _lib.set_tunable("lds_codegen", old[1])
    for k in GEO_KNOBS + ("lds_col_split",):
        _lib.set_tunable(k, 0)
    _lib.set_tunable("lds_round_tiles", 1)


def feats(rng, n, h, dt):
    dt = np.dtype(dt)
    if dt.kind == "f":
        return (rng.random((n, h)) * 2 - 1).astype(dt)
    info = np.iinfo(dt)
    return rng.integers(info.min, info.max, size=(n, h), dtype=np.int64).astype(dt)


def run(rowptr, col, x, vals=None, device=False):
    n, ncols, h = len(rowptr) - 1, x.shape[0], x.shape[1]
    rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
    v = None if vals is None else [np.ascontiguousarray(vals, x.dtype).ctypes.data]
    hd = _lib.group_create(_lib.CSR, CODE[x.dtype], [rp.ctypes.data], [ci.ctypes.data], v, [n], [ncols], [len(ci)], [1], [h], h)
    try:
        info, note = _lib.group_lds_code(hd), _lib.group_lds_note(hd)
        out = np.full((n, h), 77, dtype=x.dtype)
        xx = np.ascontiguousarray(x)
        _lib.spmm_run_group(hd, [xx.ctypes.data], out.ctypes.data)
        info["runs"] = _lib.group_lds_runs(hd)   # (a plan existing does not say that this call took it)
    finally:
        _lib.group_free(hd)
    return out, info, note


@pytest.mark.parametrize("dt", [np.float32, np.int32, np.int16, np.int8, np.float64, np.int64])
def test_device_stream_equals_host_stream_every_type(rng, checked, dt):
    for n, ncols, h, avg in ((1, 1, 64, 1), (300, 700, 64, 12), (3000, 2500, 100, 12), (5000, 300, 65, 40), (2000, 9000, 200, 25), (1500, 300, 64, 250)):
        rowptr, col = random_csr(rng, n, ncols, avg, long_rows=[(0, 5000)] if n > 100 else ())
        h = (h * max(1, 4 // np.dtype(dt).itemsize) + 1) // 2 * 2   # (narrow types: as many BYTES per row; INT16 rows hold whole lanes)
        x = feats(rng, ncols, h, dt)
        got, info, note = run(rowptr, col, x)
        assert info["active"] == 1 and info["device_generated"] == 1 and info["runs"] >= 1 and note == "code-stream form", (info, note)
        assert got.tobytes() == oracle.spmm_csr(rowptr, col, None, x).tobytes(), (dt, n, ncols, h)


def test_valued_float_entries(rng, checked):
    rowptr, col = random_csr(rng, 2500, 1800, 18, long_rows=[(7, 2600)])
    x = feats(rng, 1800, 96, np.float32)
    vals = (rng.random(len(col)) * 2 - 1).astype(np.float32)
    got, info, note = run(rowptr, col, x, vals=vals)
    assert info["device_generated"] == 1, note
    assert got.tobytes() == oracle.spmm_csr(rowptr, col, vals, x).tobytes()   # products and sums rounded separately, stored order


@pytest.mark.parametrize("small", [True, False])
def test_valued_int32_entries(rng, checked, small):
    """valued INT32 on the code stream (round 5; the reference's grande loop multiplies by the stored value for every type,
    spmm_grande/spmm_mul_csr.c:131): v_mul_lo_u32 takes no literal on gfx9, so the value rides as an INLINE CONSTANT when every value of
    the matrix lies in [-16, 64], else through an SGPR (s_mov_b32 + v_mul_lo_u32); products and sums wrap modulo 2^32 like the CPU loop's"""
    rowptr, col = random_csr(rng, 2500, 1800, 18, long_rows=[(7, 2600)])
    x = feats(rng, 1800, 96, np.int32)
    vals = rng.integers(-16, 65, size=len(col)).astype(np.int32) if small else rng.integers(-2**31, 2**31 - 1, size=len(col), dtype=np.int64).astype(np.int32)
    got, info, note = run(rowptr, col, x, vals=vals)
    assert info["active"] == 1 and info["device_generated"] == 1 and info["runs"] >= 1, (info, note)
    assert got.tobytes() == oracle.spmm_csr(rowptr, col, vals, x).tobytes()


@pytest.mark.parametrize("small", [True, False])
def test_valued_int64_entries(rng, checked, small):
    """valued INT64 on the code stream (round 5): the 64-bit product from 32-bit pieces (v_mul_lo_u32 on the high half, a correction for negative
    values, v_mad_u64_u32 for product and sum of the low half), the value inline ([-16, 64]) or through an SGPR; wraps modulo 2^64 like the CPU loop.
    A matrix with a value beyond 32 bits takes the full form: both halves through s[94:95], x_lo * v_lo + 2^32 (x_hi * v_lo + x_lo * v_hi)."""
    rowptr, col = random_csr(rng, 2200, 1500, 16, long_rows=[(3, 2100)])
    x = feats(rng, 1500, 70, np.int64)
    vals = (rng.integers(-16, 65, size=len(col)) if small else rng.integers(-2**31, 2**31 - 1, size=len(col), dtype=np.int64)).astype(np.int64)
    got, info, note = run(rowptr, col, x, vals=vals)
    assert info["active"] == 1 and info["device_generated"] == 1 and info["runs"] >= 1, (info, note)
    assert got.tobytes() == oracle.spmm_csr(rowptr, col, vals, x).tobytes()
    wide = vals.copy()
    wide[5] = 2**40 + 3                                     # one value beyond 32 bits: the full form for this matrix
    if not small:
        wide[::3] = rng.integers(-2**63, 2**63 - 1, size=len(wide[::3]), dtype=np.int64)
    got, info, note = run(rowptr, col, x, vals=wide)
    assert info["active"] == 1 and info["device_generated"] == 1 and info["runs"] >= 1, (info, note)
    assert got.tobytes() == oracle.spmm_csr(rowptr, col, wide, x).tobytes()


@pytest.mark.parametrize("dt", [np.int16, np.int8])
@pytest.mark.parametrize("small", [True, False])
def test_valued_int16_int8_entries(rng, checked, dt, small):
    """valued INT16 / INT8 on the code stream (round 5): v_pk_mul_lo_u16 x, V, x op_sel_hi:[0,1] -- both 16-bit features of a lane times the value, inline
    ([-16, 64]) or through s94 -- then v_pk_add_u16; INT8 on its features widened to 16 bits (the store keeps the low byte).  Wraps like the CPU loop."""
    info = np.iinfo(dt)
    rowptr, col = random_csr(rng, 2200, 1500, 16, long_rows=[(3, 2100)])
    x = feats(rng, 1500, 280, dt)
    vals = (rng.integers(-16, 65, size=len(col)) if small else rng.integers(info.min, info.max, size=len(col), endpoint=True)).astype(dt)
    got, inf, note = run(rowptr, col, x, vals=vals)
    assert inf["active"] == 1 and inf["device_generated"] == 1 and inf["runs"] >= 1, (inf, note)
    assert got.tobytes() == oracle.spmm_csr(rowptr, col, vals, x).tobytes()


def test_valued_double_entries(rng, checked):
    """valued DBL64 on the code stream (round 5): per entry s_mov_b32 x 2 + v_mul_f64 with the value in s[94:95], then v_add_f64 -- product and sum
    rounded separately, every row summed by one wave in stored order: the CPU loop's bits"""
    rowptr, col = random_csr(rng, 2200, 1500, 16, long_rows=[(3, 2100)])
    x = feats(rng, 1500, 70, np.float64)
    vals = (rng.random(len(col)) * 2 - 1).astype(np.float64)
    got, info, note = run(rowptr, col, x, vals=vals)
    assert info["active"] == 1 and info["device_generated"] == 1 and info["runs"] >= 1, (info, note)
    assert got.tobytes() == oracle.spmm_csr(rowptr, col, vals, x).tobytes()


@pytest.mark.parametrize("geo", [(8, 5, 0, 0, 0, 0), (16, 2, 0, 0, 0, 0), (16, 3, 0, 8, 2, 1), (8, 2, 0, 10, 2, 0), (8, 3, 0, 6, 3, 1), (8, 4, 160, 12, 2, 1), (8, 8, 64, 2, 2, 1), (8, 5, 32, 4, 3, 1)])
def test_device_stream_equals_host_stream_every_geometry(rng, checked, geo):
    for k, v in zip(GEO_KNOBS, geo):
        _lib.set_tunable(k, v)
    _lib.set_tunable("lds_round_tiles", int(rng.integers(0, 2)))
    for dt in (np.float32, np.int32):
        n, ncols, h = int(rng.integers(200, 6000)), int(rng.integers(100, 6000)), int(rng.integers(33, 300))
        rowptr, col = random_csr(rng, n, ncols, float(rng.uniform(2, 50)), empty_frac=0.3, long_rows=[(int(rng.integers(0, n)), int(rng.integers(1, 9000)))])
        x = feats(rng, ncols, h, dt)
        got, info, note = run(rowptr, col, x)
        assert info["active"] == 1 and info["device_generated"] == 1, (geo, info, note)
        assert got.tobytes() == oracle.spmm_csr(rowptr, col, None, x).tobytes(), (geo, dt, n, ncols, h)


@pytest.mark.parametrize("splits", [2, 3, 8])
def test_column_split_tiles(rng, checked, splits):
    """short row shares (a rank's share on N GPUs): every row tile becomes `splits` workgroup tiles, each with a range of the chunks, partial
    sums reduced in range order -- generated on the device like the rest (the entries of every (row tile, range) are counted there before
    the tiles are put in launch order), word for word the host encoder's"""
    old = _lib.set_tunable("lds_col_split", splits), _lib.set_tunable("lds_col_split_f32", 2)
    try:
        for dt in (np.int32, np.float32):
            n, ncols, h = int(rng.integers(500, 4000)), int(rng.integers(2000, 9000)), int(rng.integers(64, 260))
            rowptr, col = random_csr(rng, n, ncols, float(rng.uniform(5, 40)), empty_frac=0.2, long_rows=[(3, 4000)])
            x = feats(rng, ncols, h, dt) if dt == np.int32 else rng.integers(-8, 4, size=(ncols, h)).astype(np.float32)   # (exact float sums: the ranges reorder them)
            got, info, note = run(rowptr, col, x)
            assert info["active"] == 1 and info["device_generated"] == 1 and info["runs"] >= 1, (info, note)
            assert got.tobytes() == oracle.spmm_csr(rowptr, col, None, x).tobytes(), (splits, dt)
    finally:
        _lib.set_tunable("lds_col_split", old[0])
        _lib.set_tunable("lds_col_split_f32", old[1])


def test_plans_the_device_form_does_not_cover_say_so(rng):
    """the mid-slot hand-off (and lds_codegen = 0) is written by the host encoder, and the group says it (nothing silent)"""
    old = _lib.set_tunable("lds_mode", 1)
    try:
        rowptr, col = random_csr(rng, 3000, 2500, 20)
        x = feats(rng, 2500, 128, np.int32)
        want = oracle.spmm_csr(rowptr, col, None, x)
        for knob, val, why in (("lds_code_boundary", 2, "mid-slot"), ("lds_codegen", 0, "lds_codegen = 0")):
            prev = _lib.set_tunable(knob, val)
            prev_split = _lib.set_tunable("lds_col_split", 1) if knob != "lds_col_split" else None   # (no automatic column ranges)
            try:
                got, info, note = run(rowptr, col, x)
            finally:
                _lib.set_tunable(knob, prev)
                if prev_split is not None:
                    _lib.set_tunable("lds_col_split", prev_split)
            assert info["active"] == 1 and info["device_generated"] == 0 and "host encoder" in note and why in note, (knob, info, note)
            assert got.tobytes() == want.tobytes()
    finally:
        _lib.set_tunable("lds_mode", old)


def test_full_reddit_shape_word_for_word_and_creation_time():
    """the bench workload: the 1 GB code stream of the Reddit-shaped graph, generated on the device, is the host encoder's byte for byte
    (lds_codegen = 2), and creating the group with the device form alone takes a fraction of the host form's time"""
    dev = torch.device("cuda", 0)
    n, nnz, dmax = synth.SHAPES["reddit"]
    rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
    x = synth.features(n, 256, torch.float32, seed=0, device=dev)
    out = torch.empty((n, 256), dtype=torch.float32, device=dev)
    times = {}
    for mode in (2, 1, 0):
        old = _lib.set_tunable("lds_codegen", mode)
        try:
            torch.cuda.synchronize()
            t0 = time.time()
            hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [256], 256)
            torch.cuda.synchronize()
            times[mode] = time.time() - t0
        finally:
            _lib.set_tunable("lds_codegen", old)
        info = _lib.group_lds_code(hd)
        assert info["active"] == 1 and info["device_generated"] == (1 if mode else 0), (mode, info, _lib.group_lds_note(hd))
        _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
        torch.cuda.synchronize()
        colcount = torch.bincount(col.long(), minlength=n).double()
        assert torch.equal(out.double().sum(0), colcount @ x.double()), mode   # (integer-valued features: exact)
        _lib.group_free(hd)
    print(f"\n[codegen] group_create of the Reddit-shaped FLT32 group: device {times[1] * 1e3:.0f} ms, host encoder {times[0] * 1e3:.0f} ms, "
          f"device + word-for-word check {times[2] * 1e3:.0f} ms")
    assert times[1] < 0.6 * times[0], times
