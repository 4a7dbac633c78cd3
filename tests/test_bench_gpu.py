"""bench.py's output contract (one JSON line with the roofline and cpu_baseline objects), on the GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_emits_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--cpu-rows", "4096"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and rf["kernel_ms"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["check"].startswith("bit-exact")
    # value = whole-job GFLOP/s of the stated workload
    flops = 2 * d["config"]["nnz"] * d["config"]["h"]
    assert abs(d["value"] - flops / (d["ms_per_step"] * 1e-3) / 1e9) / d["value"] < 1e-3
    # round 6: BASELINE's other configurations beside the headline, each with its own roofline object and a check that passed
    bc = d["baseline_configs"]
    c3, c4, c5 = bc["configs[2]_products_coo_i32"], bc["configs[3]_reddit_gcn_3_layers_one_gpu"], bc["configs[4]_papers100m_per_gpu"]
    for cfg in (c3, c4, c5["feature_split_1x8"], c5["grid_2x4"]):
        assert "error" not in cfg, cfg
        r6 = cfg["roofline"]
        assert r6["bound"] == "hbm" and r6["peak"] == 8000.0 and r6["kernel_ms"] > 0 and abs(r6["frac"] - r6["achieved"] / r6["peak"]) < 1e-3
        assert "MISMATCH" not in cfg["check"], cfg["check"]
    assert c3["dtype"] == "i32" and c3["N"] == 2449029 and c3["check"].startswith("bit-exact")
    assert c5["feature_split_1x8"]["h"] == 16 and c5["grid_2x4"]["h"] == 32 and c4["aggregations_per_forward"] == 3
    assert d["extra"]["end_to_end_cpu_tensors_ms"] > d["ms_per_step"] and "MISMATCH" not in d["extra"]["end_to_end_cpu_tensors"]["check"]
    assert "NOT BASELINE" in d["extra"]["note_on_structured_graphs"]


def _parse_like_the_harness(stdout):
    """the reference harness' result grammar (utils/experiment.py:466-491): `[DATA]key: value` lines, grouped by the
    'Repeat' / 'Model' banner lines, mean over repeats of the per-repeat sums"""
    import collections
    import re

    import numpy as np

    engine = re.compile(r"^\[DATA](.*?): (.*)")
    results, repeat = collections.defaultdict(list), 0
    for line in stdout.splitlines():
        if line.startswith("-------------------- Repeat") or line.startswith("-------------------- Model"):
            repeat += 1
        m = engine.findall(line.strip())
        if m:
            results[m[0][0]].append(float(m[0][1]))
    out = {k: np.asarray(v).reshape(repeat, -1).mean(axis=0).sum(axis=-1) for k, v in results.items()}
    out["repeat"] = repeat
    return out


@pytest.mark.parametrize("version,lib,fmt", [("spmm", "spmm_default", "COO"), ("spmm", "spmm_default", "CSR"),
                                             ("grande", "spmm_grande", "CSR"), ("spmv", "spmv_sparseP", "COO")])
def test_spmm_test_driver_under_the_harness_command_line(version, lib, fmt):
    """the command line utils/experiment.py:408-424 builds (singular --sp_part/--ds_part, quoted --lib_path, --nr_dpus)
    drives spmm_test.py, and its stdout parses with the harness grammar"""
    lib_path = os.path.join(ROOT, "backend_pim", lib, "build", "libbackend_pim.so")
    cmd = [sys.executable, os.path.join(ROOT, "spmm_test.py"), "--dataset=PubMed", "--datadir=./data", f"--sp_format={fmt}",
           "--data_type=INT32", "--hidden_size=64", "--sp_part=2", "--ds_part=2", "--repeat=2", f"--lib_path={lib_path}",
           f"--version={version}",
           # grande wants one rank (= 8 compute units here, 64 DPUs on UPMEM) per sparse part (grande.py:57)
           "--nr_dpus=16" if version == "grande" else "--nr_dpus=64"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    res = _parse_like_the_harness(r.stdout)
    assert res["repeat"] == 2
    assert res["pim_time_spmm(ms)"] > 0 and res["torch_time(ms)"] > 0 and res["outputs_equal"] == 1.0


def test_inference_driver_under_the_harness_command_line():
    lib_path = os.path.join(ROOT, "backend_pim", "spmm_grande", "build", "libbackend_pim.so")
    cmd = [sys.executable, os.path.join(ROOT, "inference.py"), "--dataset=PubMed", "--datadir=./data", "--sp_format=CSR",
           "--data_type=INT32", "--hidden_size=64", "--sp_part=1", "--ds_part=2", "--repeat=2", f"--lib_path={lib_path}",
           "--version=grande", "--model=gcn", "--num_layers=3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    res = _parse_like_the_harness(r.stdout)
    assert res["repeat"] == 2 and res["infer_time(ms)"] > 0


@pytest.mark.parametrize("backend,model", [("spmm_default", None), ("spmm_grande", "gcn"), ("spmv_sparseP", None)])
def test_experiment_harness_builds_runs_and_parses(backend, model, tmp_path):
    """pygim_amd/experiment.py end to end on the GPU: Experiment.build links the variant's library where run() looks for
    it, run() writes the named .out / .err, parse_result reduces them (utils/experiment.py:277-491)"""
    sys.path.insert(0, ROOT)
    from pygim_amd.experiment import Experiment

    e = Experiment(dataset="PubMed", sp_part=2 if model is None else 1, ds_part=2, sp_format="COO" if "spmv" in backend else "CSR",
                   dense_size=64, dtype="INT32", balance="nnz", balance_tsklt="nnz", nr_tasklets=16, cg_lock=False, cache_size=32,
                   backend=backend, nr_dpus=64 if model is None else None, model=model, num_layers=3 if model else None)
    build_root, res = str(tmp_path / "build"), str(tmp_path / "results")
    lib = e.build(ROOT, build_root)
    assert os.path.isfile(lib)
    assert e.run(ROOT, "./data", build_root, result_root=res, repeat=2) == 0
    assert e.status_at(res) == "done"
    got = e.parse_result(res)
    assert got["repeat"] == 2
    with open(e.stdout_path(res)) as f:
        same = _parse_like_the_harness(f.read())
    assert set(same) == set(got) and all(np.allclose(same[k], got[k]) for k in got)
    if model is None:
        assert got["pim_time_spmm(ms)"] > 0 and got["outputs_equal"] == 1.0
    else:
        assert got["infer_time(ms)"] > 0
    # the library that ran is the one linked into the per-configuration build directory
    with open(e.stderr_path(res)) as f:
        assert "Traceback" not in f.read()


@pytest.mark.parametrize("partition", ["auto", "pipelined", "row", "feature", "pipelined-feature", "push", "push-feature"])
def test_bench_two_ranks_logic_check(partition):
    """the N > 1 paths of bench.py with 2 ranks over gloo on this one GPU (a logic check, not a measurement): every
    partition assembles the exact C on every rank (column-count checksum) and prints one JSON line from rank 0"""
    env = dict(os.environ, PYGIM_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr=127.0.0.1",
           f"--master-port={_free_port()}", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--shape", "products-mini",
           "--partition", partition]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["check"].startswith("column-count checksum")
    assert d["config"]["ms_per_step_products_only"] > 0


def test_bench_launches_its_own_ranks_as_a_plain_command():
    """`python bench.py --gpus 2 ...` WITHOUT torch.distributed.run (the driver's command form at N > 1): bench.py starts its own
    ranks as child processes, relays their output and exits with their code -- one JSON line, n_gpus 2, rc 0 (2 ranks over
    gloo on this one GPU: a logic check, not a measurement).  A rank that fails must make the command fail."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYGIM_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--shape", "products-mini"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["check"].startswith("column-count checksum")
    bad = subprocess.run(cmd + ["--shape", "no-such-shape"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]


def test_a_rank_that_fails_in_set_up_ends_every_rank_promptly():
    """no rank may sit in a fence while another has raised.  One rank is made to fail while it sets its
    partition up (PYGIM_BENCH_FAIL_RANK); every candidate's set-up ends in a status word that all ranks reduce, so the launch exits
    non-zero within seconds -- not at the collective time-out, not at the watchdog -- and prints no JSON line.  Every rank also
    reports what it built (kernel family, creation time, threads) before the timed region."""
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYGIM_BENCH_BACKEND="gloo", PYGIM_RANK_TIMEOUT="600", PYGIM_COLLECTIVE_TIMEOUT="300")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--shape", "products-mini"]
    for part in ("auto", "row", "pipelined"):
        t0 = time.time()
        bad = subprocess.run(cmd + ["--partition", part], capture_output=True, text=True, timeout=900, env={**env, "PYGIM_BENCH_FAIL_RANK": "1"}, cwd=ROOT)
        took = time.time() - t0
        assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")], (part, bad.stdout[-500:])
        assert "forced set-up failure" in bad.stderr and "no candidate could be set up on every rank" in bad.stderr, bad.stderr[-3000:]
        assert took < 240, f"--partition {part}: the failing launch took {took:.0f} s (hung until a time-out?)"
    ok = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert ok.returncode == 0, ok.stderr[-3000:]
    d = json.loads([ln for ln in ok.stdout.splitlines() if ln.startswith("{")][0])
    assert len(d["config"]["per_rank"]) == 2 and all(pr["groups"] and pr["group_create_ms"] > 0 for pr in d["config"]["per_rank"])
    assert "contract verified" in d["check"]
    assert ok.stderr.count("[bench] rank ") >= 2


def test_bench_eight_ranks_logic_check():
    """the driver's largest launch, `--gpus 8`, as 8 ranks over gloo on this ONE GPU with the small shape (a logic check: eight processes
    share the device): one JSON line, every rank reports what it built, the gathered C is exact on every rank.  (With the full
    Reddit shape eight processes on one GPU starve one another for minutes -- profiles/r04_multirank.txt -- so the small shape here.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYGIM_BENCH_BACKEND="gloo", PYGIM_RANK_TIMEOUT="500", PYGIM_COLLECTIVE_TIMEOUT="400", PYGIM_LAUNCH_TIMEOUT="560")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--shape", "products-mini", "--partition", "pipelined"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 8 and len(d["config"]["per_rank"]) == 8 and d["check"].startswith("column-count checksum")
    assert all(pr["plan_threads"] >= 1 and pr["groups"] for pr in d["config"]["per_rank"])
