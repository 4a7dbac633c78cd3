"""pygim_amd/experiment.py against the reference harness' contract (utils/experiment.py:157-491, helpers.py:44-103):
result-file names, the done / failed / todo states, the driver command line, and the [DATA] reduction.  CPU only:
the runs use the reference's own CPU-runnable leg (--version cpu, BASELINE configs[0])."""
import argparse
import logging
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pygim_amd.experiment import Experiment, make_argument_parser, parse_stdout, results_to_csv, run_experiments  # noqa: E402


def _exp(**kw):
    base = dict(dataset="Cora", sp_part=2, ds_part=4, sp_format="CSR", dense_size=32, dtype="FLT32", balance="nnz",
                balance_tsklt="row", nr_tasklets=16, cg_lock=False, cache_size=32, backend="spmm_default")
    base.update(kw)
    return Experiment(**base)


def test_result_and_build_names_follow_the_reference_scheme():
    # experiment.py:202-275 read by hand: build keys in order, cache_size moved behind the run keys, optional keys only
    # away from their defaults, '.failed' appended after the extension
    e = _exp()
    stem = ("backend=spmm_default-spf=CSR-dtype=FLT32-blnc=nnz-blnc_tsklt=row-nr_tasklets=16-cg_lock=False"
            "-spds=2x4-dataset=Cora-dense_size=32-nr_dpus=None-cache_size=32")
    assert e.stdout_path("/res") == f"/res/{stem}.out"
    assert e.stderr_path("/res") == f"/res/{stem}.err"
    assert e.stdout_path("/res", failed=True) == f"/res/{stem}.out.failed"
    assert e.build_path("/b") == ("/b/backend=spmm_default-spf=CSR-dtype=FLT32-blnc=nnz-blnc_tsklt=row-nr_tasklets=16"
                                  "-cg_lock=False-cache_size=32")
    e = _exp(backend="spmm_multigroup", groups_per_rank=2, merge="row", sync=False, nr_dpus=512, model="gcn", num_layers=3)
    assert e.stdout_path("r") == ("r/backend=spmm_multigroup-spf=CSR-dtype=FLT32-blnc=nnz-blnc_tsklt=row-nr_tasklets=16"
                                  "-cg_lock=False-gpr=2-merge=row-sync=False-spds=2x4-dataset=Cora-dense_size=32"
                                  "-nr_dpus=512-cache_size=32-model=gcn-num_layers=3.out")
    assert _exp(backend=None).backend == "backend_pim_group" and _exp(backend=None, ds_part=0).backend == "backend_pim_grande"
    assert _exp(backend=None, groups_per_rank=4).backend == "backend_pim_multigroup"
    assert set(e.build_params) == {"backend", "sp_format", "dtype", "balance", "balance_tsklt", "nr_tasklets", "cg_lock",
                                   "cache_size", "merge", "sync"}


def test_command_line_is_the_one_the_reference_builds():
    e = _exp(nr_dpus=64)
    cmd = e.command("/src", "/data", "/b", repeat=5)
    assert cmd[1] == "/src/spmm_test.py"
    assert cmd[2:] == ["--dataset=Cora", "--datadir=/data", "--sp_format=CSR", "--data_type=FLT32", "--hidden_size=32",
                       "--sp_part=2", "--ds_part=4", "--repeat=5", f"--lib_path={e.build_path('/b')}/libbackend_pim.so",
                       "--version=spmm", "--nr_dpus=64"]
    g = _exp(backend="spmm_grande", model="gin", num_layers=2, dataset="ogbnproteins").command("/src", "/d", "/b")
    assert g[1] == "/src/inference.py" and "--version=grande" in g and g[-2:] == ["--model=gin", "--num_layers=2"]
    assert "--dataset=ogbn-proteins" in g
    assert "--version=spmv" in _exp(backend="spmv_sparseP").command("/s", "/d", "/b")
    c = _exp(backend="cpu").command("/s", "/d", "/b")
    assert "--version=cpu" in c and "--lib_path=None" in c
    with pytest.raises(NotImplementedError):
        _exp(backend=None).command("/s", "/d", "/b")   # the default names are not runnable in the reference either
    m = _exp(backend="spmm_multigroup", groups_per_rank=2).command("/s", "/d", "/b")
    assert m[-1] == "--group_per_rank=2" and "--version=spmm" in m    # experiment.py:432-434


def test_data_lines_reduce_like_the_reference():
    text = """Namespace(...)
-------------------- Model=spmm_test Repeat=0 --------------------
[DATA]torch_time(ms):  10.0
[DATA]load_time(ms): 1.0
[DATA]load_time(ms): 2.0
-------------------- Model=spmm_test Repeat=1 --------------------
[DATA]torch_time(ms):  30.0
[DATA]load_time(ms): 3.0
[DATA]load_time(ms): 6.0
noise [DATA]not_at_line_start: 5
"""
    got = parse_stdout(text.splitlines(keepends=True))
    # per key: fields of one repeat are summed, repeats averaged (experiment.py:484-488)
    assert got["repeat"] == 2 and got["torch_time(ms)"] == 20.0 and got["load_time(ms)"] == 6.0
    assert "not_at_line_start" not in got


def test_run_writes_named_files_and_failures_are_marked(tmp_path):
    res = str(tmp_path / "results")
    env_keep = {k: os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")}
    os.environ["HIP_VISIBLE_DEVICES"] = os.environ["CUDA_VISIBLE_DEVICES"] = ""   # the cpu leg needs no device
    try:
        ok = _exp(backend="cpu")
        assert ok.status_at(res) == "todo"
        assert ok.build(ROOT, str(tmp_path / "b")) is None
        assert ok.run(ROOT, "./data", str(tmp_path / "b"), result_root=res, repeat=2) == 0
        assert ok.status_at(res) == "done" and os.path.exists(ok.stderr_path(res))
        parsed = ok.parse_result(res)
        assert parsed["repeat"] == 2 and parsed["torch_time(ms)"] >= 0
        bad = _exp(backend="cpu", dtype="FLT16")            # not a data type the driver knows: non-zero exit
        with pytest.raises(RuntimeError):
            bad.run(ROOT, "./data", str(tmp_path / "b"), result_root=res, repeat=1)
        assert bad.status_at(res) == "failed"
        assert os.path.exists(bad.stdout_path(res, failed=True)) and os.path.getsize(bad.stderr_path(res, failed=True)) > 0
        os.remove(bad.stdout_path(res, failed=True))
        assert bad.run(ROOT, "./data", str(tmp_path / "b"), result_root=res, repeat=1, silent=True) != 0

        # helpers.py:83-99: done runs are skipped; failed ones are retried unless failures are accepted
        log = logging.getLogger("test_experiment")
        args = argparse.Namespace(result_root=res, src_root=ROOT, data_root="./data", build_root=str(tmp_path / "b"),
                                  dry_run=False, skip_failed=True, force_rebuild=False)
        stamp = os.path.getmtime(ok.stdout_path(res))
        third = _exp(backend="cpu", dense_size=16)
        run_experiments(args, [], [ok, bad, third], log, accept_failures=True, repeat=1)
        assert os.path.getmtime(ok.stdout_path(res)) == stamp and third.status_at(res) == "done"
    finally:
        for k, v in env_keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_dry_run_and_build_link(tmp_path):
    e = _exp()
    assert e.run(ROOT, "./data", str(tmp_path), result_root=str(tmp_path / "r"), dry_run=True) == 0
    assert e.status_at(str(tmp_path / "r")) == "todo"
    built = os.path.join(ROOT, "backend_pim", "spmm_default", "build", "libbackend_pim.so")
    if os.path.exists(built):       # (a checkout without built shims would compile them here: left to build())
        target = e.build(ROOT, str(tmp_path / "b"))
        assert os.path.samefile(target, built) and target == os.path.join(e.build_path(str(tmp_path / "b")), "libbackend_pim.so")
        assert e.build(ROOT, str(tmp_path / "b")) == target   # second call: nothing to do
    with pytest.raises(NotImplementedError):
        _exp(backend="backend_pim_group").build(ROOT, str(tmp_path / "b"))
    ns = make_argument_parser("exp1").parse_args(["run", "--skip_failed"])
    assert ns.action == "run" and ns.result_root == "./results/exp1" and ns.skip_failed and not ns.dry_run


def test_out_files_to_csv(tmp_path):
    """the per-run CSVs of backend_pim/spmv_sparseP/parse_results.py: a row per repeat, the avg row, and the derived
    pim_time_dense(ms) = pim_time_spmm(ms) - load_sparse_time"""
    out = tmp_path / "res"
    out.mkdir()
    (out / "a.out").write_text("-------------------- Repeat 0\n[DATA]load_sparse_time: 2.0\n[DATA]pim_time_spmm(ms): 10.0\n"
                               "-------------------- Repeat 1\n[DATA]load_sparse_time: 4.0\n[DATA]pim_time_spmm(ms): 20.0\n")
    (out / "b.out").write_text("-------------------- Model=gcn Repeat=0\n[DATA]infer_time(ms): 3.0\n")
    (out / "b.err").write_text("ignored")
    avg = results_to_csv(str(out))
    assert avg["a"] == {"load_sparse_time": 3.0, "pim_time_spmm(ms)": 15.0, "pim_time_dense(ms)": 12.0}
    assert avg["b"] == {"infer_time(ms)": 3.0}
    rows = (out / "csv_result" / "a.csv").read_text().splitlines()
    assert rows[0] == "Repeat, load_sparse_time, pim_time_spmm(ms), pim_time_dense(ms)"
    assert rows[1] == "0, 2.0, 10.0, 8.0" and rows[2] == "1, 4.0, 20.0, 16.0" and rows[3] == "avg, 3.0, 15.0, 12.0"
    table = (out / "csv_result" / "average_all.csv").read_text().splitlines()
    assert table[0].startswith("run, load_sparse_time") and table[1].startswith("a, 3.0, 15.0, 12.0") and table[2].startswith("b, ")


def test_tuned_partition_and_reference_shaped_autotune(monkeypatch):
    """utils/autotuner.py:263's call shape over the chooser, and Experiment(tune=...) taking its answer (experiment.py:402-405)"""
    from pygim_amd import autotune

    got = autotune.autotune_dataset("./data", "Reddit", 256, [(1, 8), (2, 4), (4, 2), (8, 1)])
    assert got[2:] == ["nnz", "nnz", None] and got[0] * got[1] == 8
    table = {(c.row_parts, c.feat_parts): c.seconds for c in autotune.choose(232965, 232965, 114615892, 256, 4, 8)[1]}
    assert table[(got[0], got[1])] == min(table.values())
    assert autotune.autotune_dataset("./data", "PubMed", 64, [(1, 128)]) == [None, None, None, None, None]  # more parts than columns
    monkeypatch.setenv("WORLD_SIZE", "8")
    e = _exp(dataset="Reddit", dense_size=256, tune=True, backend="spmm_default")
    e._apply_tuned_partition("./data")
    assert (e.sp_part, e.ds_part) == (got[0], got[1]) and e.balance == "nnz"
    off = _exp(tune="FALSE")
    assert off.command(ROOT, "./data", "/b")[7:9] == ["--sp_part=2", "--ds_part=4"]   # 'FALSE' leaves the split alone
