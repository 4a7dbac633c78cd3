"""pygim_amd/autotune.py on the CPU: the product model follows the library's plan rules of round 6 (column-split shares, the row-tail rule, the staging floor) and the grid
chooser prefers what one GPU measured (profiles/r06_exp_shard.txt).  The reference prices its candidates the same way from UPMEM constants (utils/autotuner.py:263-343)."""
from pygim_amd import autotune, synth


def test_row_tail_rule_buys_a_whole_column_range_for_shares_that_all_but_fit():
    n, nnz, _ = synth.SHAPES["reddit"]
    # 16 full-height tiles x 4 slices x 4 ranges = 256 workgroups hold 29 184 rows: 29 471 rows leave 287 (1 %) to the tail kernels ...
    with_tail, _ = autotune.product_seconds(29471, n, nnz // 8, 256, 4)
    old = autotune.LDS_ROW_TAIL
    try:
        autotune.LDS_ROW_TAIL = 0      # ... without the rule: 17 tiles, three ranges, more of X per workgroup
        without, _ = autotune.product_seconds(29471, n, nnz // 8, 256, 4)
    finally:
        autotune.LDS_ROW_TAIL = old
    assert with_tail < without and 0.35e-3 < with_tail < 0.55e-3, (with_tail, without)
    # a share far from a tile multiple keeps all its rows in the plan: the rule changes nothing
    a, _ = autotune.product_seconds(31000, n, nnz // 8, 256, 4)
    autotune.LDS_ROW_TAIL = 0
    try:
        b, _ = autotune.product_seconds(31000, n, nnz // 8, 256, 4)
    finally:
        autotune.LDS_ROW_TAIL = old
    assert a == b


def test_model_against_the_measured_shapes_of_round_6():
    """ms measured on one MI355X (profiles/r06_exp_shard.txt), model within 25 %"""
    n, nnz, _ = synth.SHAPES["reddit"]
    for rows, frac, h, measured in ((232965, 1, 256, 2.07), (116638, 2, 256, 1.11), (58490, 4, 256, 0.69), (29471, 8, 256, 0.45), (232965, 1, 128, 1.08),
                                    (232965, 1, 64, 0.65), (232965, 1, 32, 0.49), (116638, 2, 64, 0.365), (58490, 4, 128, 0.385)):   # (h = 32: the half-split plan, 0.60 before it)
        t, _ = autotune.product_seconds(rows, n, nnz // frac, h, 4)
        assert abs(t * 1e3 - measured) <= 0.25 * measured, (rows, h, t * 1e3, measured)


def test_products_too_narrow_for_the_lds_path_keep_the_sweep_price():
    n, nnz, _ = synth.SHAPES["reddit"]
    assert autotune.lds_product_seconds(n, n, nnz, 4, 4) is None and autotune.lds_product_seconds(n, n, nnz, 5, 4) is not None
    t4, _ = autotune.product_seconds(n, n, nnz, 4, 4)
    assert t4 > 0
    # a fifth of the entries (0.77 per staged column): the LDS-staged kernel still pays at 16 features as at 256 (measured: 0.26 against 0.38 ms on the sweep); a tenth: neither
    assert autotune.lds_product_seconds(n, n, nnz // 5, 16, 4) is not None and autotune.lds_product_seconds(n, n, nnz // 5, 256, 4) is not None
    assert autotune.lds_product_seconds(n, n, nnz // 10, 16, 4) is None and autotune.lds_product_seconds(n, n, nnz // 10, 256, 4) is None


def test_eight_ranks_prefer_the_grid_whose_product_is_cheapest():
    n, nnz, _ = synth.SHAPES["reddit"]
    best, table = autotune.choose(n, n, nnz, 256, 4, 8)
    by = {(c.row_parts, c.feat_parts): c for c in table}
    assert (best.row_parts, best.feat_parts) == (2, 4)
    assert by[(2, 4)].product_s < by[(8, 1)].product_s < by[(1, 8)].product_s      # 0.36 < 0.45 < 0.49 ms measured
    one, _ = autotune.choose(n, n, nnz, 256, 4, 1)
    assert one.product_s / by[(2, 4)].product_s > 5.0     # products alone: 5.6 x measured (the exchange over xGMI is the rest of the step)
