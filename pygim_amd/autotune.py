"""Partition chooser for N GPUs.

The reference's utils/autotuner.py:263-343 picks ``(sp_parts, ds_parts, balance)`` by pricing, for every
candidate split, load + compute + retrieve + host-merge time from UPMEM calibration constants
(autotuner.py:23-89) -- and depends on an op (``prepare_tune_csr``) that is not in the repository.
Here the same idea is restated for one node of MI355Xs with constants MEASURED this round on the
hardware (profiles/): a candidate is a (row parts x feature parts) grid over the
GPUs; its price is the slowest GPU's product time plus the collective that re-assembles C.

    product  : gather-model bytes / rate, rate = L2-blocked sweep when a row has enough entries per
               column panel, else the Infinity-Cache/HBM row-gather rate
    all-gather of C over xGMI (row parts), nothing inside the product for feature parts, but a narrower
               feature block gathers less than a cache line per entry below 128 B.
Returns the reference's ``[sp_parts, ds_parts, balance_dpu, balance_tasklet, None]`` shape.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

# measured on MI355X this round (bytes/s)
RATE_PANEL = 17.0e12        # k_csr_panel, Reddit-shaped, h = 256 f32
RATE_GATHER_FAR = 4.8e12    # row-per-wave gathers from HBM (products-shaped, X = 2.5 GB)
RATE_GATHER_MALL = 7.0e12   # row-per-wave gathers, X about the size of the Infinity Cache (Reddit, panel off)
RATE_STREAM = 5.0e12        # streaming reads/writes (index arrays, C)
XGMI_IN = 7 * 45e9          # inbound bytes/s per GPU, 7 links (spec 7 x ~64 GB/s per direction; 70 % assumed)
PANEL_BYTES = 4 << 20
LAUNCH = 6e-6               # per kernel launch incl. ramp, back to back
# LDS-staged product, code-stream form (k_lds_code8_*, round 4; profiles/r04_lds_kernel.md): one 512-thread workgroup per CU and
# (row tile, 64-feature slice), 8 waves x 228 accumulators, X streamed through a ring of 5 x 128 columns
CUS = 256
LDS_ROWS_MAX = 8 * 228      # rows of a tile (waves x accumulators per wave)
# fitted to row shares 1/1 ... 1/16 of the Reddit-shaped graph at h = 64 / 128 / 256 (profiles/r04_exp_share.txt): a workgroup takes
#   columns it streams x 3.43 ns  (a 256-byte row of X through LDS-DMA with ~96 KiB in flight: ~75 GB/s per CU)
# + stored entries x 0.26 ns      (what the entries add on top: LDS reads beside the DMA's writes)
LDS_NS_PER_COLUMN = 3.43
LDS_NS_PER_ENTRY = 0.26
LDS_NS_PER_ENTRY_HALF = 0.5  # ... in a half-split plan (its adds run under half of EXEC, range by range: 0.94 us per chunk of twice the entries, profiles/r06_exp_shard.txt)
LDS_NS_PER_COLUMN_FLOOR = 4.4   # ... and never less than this per column, however few entries a tile has: with every CU streaming, the L2 -> LDS path
                                # lands ~57 GB/s per CU (round 6, profiles/r06_stamps.txt: lighter tiles leave the 0.57 us per 128-column chunk unchanged)
LDS_COL_SPLIT = True        # FLT32 shares of a million entries and more are split into column ranges (the library's default since round 6; tunable
                            # lds_col_split_f32 = 0 keeps the bit-identical stored-order form -- callers that set it set this to False)
LDS_ROW_TAIL = 3            # percent of a share's rows that may stay outside the plan (tunable lds_row_tail)
TAIL_S = 30e-6              # the two tail kernels
LDS_MIN_LANES = 5           # narrower products keep the sweep / the SpMV kernels (tunable lds_min_width; 17 until the round's last session)
LDS_MIN_REUSE_NARROW = 0.75 # products of at most 32 lanes: stored entries per staged column below which the sweep is kept (x 1.6 for a half-split plan, which stages half the columns);
                            # measured: the same crossing as the wide products' (scripts/exp_narrow_rule.py)
LDS_HALF_SPLIT = True       # products of at most 32 lanes fold two column ranges into the halves of a wave (tunable lds_half_split, on since the device generator writes these plans)
LDS_MIN_REUSE = 0.75        # stored entries per staged column below which the sweep is kept (pygim_hip.hip lds_min_reuse_x100)


@dataclass
class Choice:
    row_parts: int
    feat_parts: int
    seconds: float
    product_s: float
    collective_s: float
    panel: bool


def lds_rows_per_tile(nrows, nslices, rmax=LDS_ROWS_MAX, cus=CUS):
    """lds_plan.hpp lds_rows_per_tile: tile height such that tiles x slices workgroups fill whole rounds of CUs"""
    tiles = -(-nrows // rmax)
    wgs = tiles * nslices
    rounds = -(-wgs // cus)
    if rounds > 8:
        return rmax
    if wgs % cus == 0:
        return -(-nrows // tiles)      # whole rounds already: tiles of equal height
    tiles2 = rounds * cus // nslices
    if tiles2 <= tiles:
        return rmax
    return min(rmax, max(16, -(-nrows // tiles2)))


def lds_product_seconds(nrows, ncols, nnz, h, es):
    """the LDS-staged product on one GPU, or None where the library keeps the sweep (4-byte elements, rows of at least 17
    elements, enough stored entries per staged column); uniform columns assumed (every tile streams every chunk)"""
    if es != 4 or h < LDS_MIN_LANES or nrows == 0 or nnz == 0:
        return None
    nsl = -(-h // 64)
    pack_s = ncols * h * es * 2 / RATE_STREAM
    cols_x = ncols      # columns a row tile stages
    ns_entry = LDS_NS_PER_ENTRY
    if LDS_HALF_SPLIT and h * es <= 128 and ncols >= 4 * 128:
        # (round 6) products of at most 32 lanes: two column ranges in the halves of a wave -- a staged row holds X[c] and X[c + H], every tile stages half the rows
        # (and meets twice the entries per chunk): all rows x 32 features 0.59 -> 0.49 ms
        cols_x = -(-ncols // 2)
        ns_entry = LDS_NS_PER_ENTRY_HALF
    # (narrow products gather fewer bytes per entry on the sweep: their plans must serve more entries per staged column; rt_plans.inc)
    min_reuse = LDS_MIN_REUSE if h * es > 128 else max(LDS_MIN_REUSE, LDS_MIN_REUSE_NARROW * (1.6 if cols_x != ncols else 1.0))
    # short row shares (pygim_hip.hip build_lds_plan): full-height tiles split into S column ranges, each workgroup lands 1 / S of X
    tall = -(-int(nrows) // LDS_ROWS_MAX)
    split = min(8, CUS // (tall * nsl)) if tall * nsl * 2 <= CUS else 1
    tail_rows = 0
    if LDS_COL_SPLIT and nnz >= (1 << 20):
        # (round 6) a share whose rows all but fit tall' x slices x S' = one workgroup per compute unit for a larger S': the last few rows (at most
        # LDS_ROW_TAIL percent) stay outside the plan (k_lds_tail_seg / _fin, a few tens of microseconds)
        for s2 in range(8, split, -1):
            tall_s = CUS // (nsl * s2)
            if not tall_s:
                continue
            covered = tall_s * LDS_ROWS_MAX
            if covered >= nrows:
                break
            if (nrows - covered) * 100 <= nrows * LDS_ROW_TAIL and tall_s * nsl * s2 * 10 >= CUS * 9:
                split, tall, tail_rows = s2, tall_s, nrows - covered
                break
    if LDS_COL_SPLIT and nnz >= (1 << 20) and split > 1:
        if nnz / (tall * cols_x) < min_reuse:
            return None
        tall2 = CUS // (nsl * split)            # (round 6) more, lighter row tiles when tall x slices x ranges leaves compute units idle
        if not tail_rows and tall < tall2 <= 2 * tall:
            tall = tall2
        per_wg = max(cols_x / split * LDS_NS_PER_COLUMN + nnz / tall / split * ns_entry, cols_x / split * LDS_NS_PER_COLUMN_FLOOR) * 1e-9
        reduce_s = (split + 1) * nrows * h * es / RATE_STREAM
        return per_wg + reduce_s + pack_s + 2 * LAUNCH + (TAIL_S if tail_rows else 0.0)
    rpt = lds_rows_per_tile(int(nrows), nsl)
    tiles = -(-int(nrows) // rpt)
    if nnz / (tiles * cols_x) < min_reuse:
        return None
    per_wg = max(cols_x * LDS_NS_PER_COLUMN + nnz / tiles * ns_entry, cols_x * LDS_NS_PER_COLUMN_FLOOR) * 1e-9
    rounds = -(-tiles * nsl // CUS)
    return rounds * per_wg + pack_s + LAUNCH


def product_seconds(nrows, ncols, nnz, h, es):
    """one GPU: rows x all columns, h features"""
    if nrows == 0 or nnz == 0 or h == 0:
        return 0.0, False
    t_lds = lds_product_seconds(nrows, ncols, nnz, h, es)
    if t_lds is not None:
        return t_lds, True
    row_bytes = h * es
    line_bytes = max(row_bytes, 128) if row_bytes < 128 else row_bytes  # at least one cache line per entry
    gather = nnz * line_bytes
    npanels = max(1, math.ceil(ncols * 128 / PANEL_BYTES))
    panel = row_bytes >= 64 and (npanels == 1 or nnz / (nrows * npanels) >= 8)
    if panel:
        nsl = math.ceil(row_bytes / 128)
        t = gather / RATE_PANEL + npanels * (2 * nrows * row_bytes) / RATE_STREAM / 8 + nsl * 4 * nnz / RATE_STREAM / 8
        t += npanels * LAUNCH + ncols * row_bytes * 2 / RATE_STREAM
    else:
        x_bytes = ncols * row_bytes
        rate = RATE_GATHER_MALL if x_bytes <= 300e6 else RATE_GATHER_FAR
        t = gather / rate + LAUNCH
    return t, panel


def price(nrows, ncols, nnz, h, es, row_parts, feat_parts):
    hp = math.ceil(h / feat_parts)
    t, panel = product_seconds(math.ceil(nrows / row_parts), ncols, nnz / row_parts, hp, es)
    gpus = row_parts * feat_parts
    # every GPU ends with the full C: it receives everything it did not compute
    coll = 0.0 if gpus == 1 else (nrows * h * es) * (1 - 1 / gpus) / XGMI_IN
    return Choice(row_parts, feat_parts, t + coll, t, coll, panel)


def choose(nrows, ncols, nnz, h, elem_bytes, n_gpus):
    """best (row parts x feature parts) grid over n_gpus GPUs and the table of all candidates"""
    table = []
    for r in range(1, n_gpus + 1):
        if n_gpus % r:
            continue
        f = n_gpus // r
        if f > h:
            continue
        table.append(price(nrows, ncols, nnz, h, elem_bytes, r, f))
    best = min(table, key=lambda c: c.seconds)
    return best, table


def autotune(nrows, ncols, nnz, hidden_size, elem_bytes=4, n_gpus=8):
    """reference-shaped result: [sp_parts, ds_parts, balance over GPUs, balance inside a GPU, None]
    (utils/autotuner.py:333-343).  sp_parts here are ROW parts, balanced by nnz."""
    best, _ = choose(nrows, ncols, nnz, hidden_size, elem_bytes, n_gpus)
    return [best.row_parts, best.feat_parts, "nnz", "nnz", None]


def autotune_dataset(datadir, dataset, hidden_size, split_set, blnc_set=(0, 2), elem_bytes=4):
    """The reference's call shape (utils/autotuner.py:263: ``autotune(datadir, dataset, hidden_size, split_set, blnc_set)``):
    price every ``(sp_parts, ds_parts)`` of ``split_set`` for the named dataset -- read from ``datadir`` when its raw files are
    there, else the dataset's node / edge counts -- as a (row parts x feature parts) grid and return the cheapest as
    ``[sp_parts, ds_parts, balance, balance_tasklet, None]`` (autotuner.py:333-343).  ``blnc_set`` selected UPMEM balance
    schemes; every split here is nnz-balanced, so it only has to be non-empty."""
    from . import datasets, synth

    assert len(split_set) > 0 and len(blnc_set) > 0
    got = datasets.load_adjacency(datadir, dataset)
    if got is not None:
        nrows, nnz = int(got[2]), int(len(got[1]))
    else:
        nrows, nnz, _ = synth.DATASETS[dataset]
    best = None
    for sp, ds in split_set:
        if sp < 1 or ds < 1 or ds > hidden_size:
            continue
        c = price(nrows, nrows, nnz, hidden_size, elem_bytes, sp, ds)
        if best is None or c.seconds < best.seconds:
            best = c
    if best is None:
        return [None, None, None, None, None]
    return [best.row_parts, best.feat_parts, "nnz", "nnz", None]
