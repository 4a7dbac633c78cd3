"""Experiment harness counterpart: the reference's sweep tooling describes one run as a
``utils.experiment.Experiment`` (experiment.py:157-491), names its result files after the frozen parameters
(``<k=v-...>.out`` / ``.err``, ``.failed`` suffix on a non-zero exit, experiment.py:250-275), skips runs whose
``.out`` exists (experiment.py:350-356, helpers.py:83-89), starts ``spmm_test.py`` / ``inference.py`` with one fixed
command line (experiment.py:408-433) and reduces the ``[DATA]key: value`` lines of the stdout file
(experiment.py:466-491).  This module keeps that contract so sweep and plot scripts written against the reference
drive this backend; what differs is ``build``: there is no per-configuration UPMEM binary to compile -- the UPMEM
knobs (balance, tasklets, cache size, lock, merge, sync) stay in the names and select nothing -- so a "build" links
the one prebuilt ``libbackend_pim.so`` of the variant into the directory the run expects to find it in.
"""
from __future__ import annotations

import collections
import dataclasses
import logging
import os
import re
import shutil
import subprocess
import sys
from typing import Optional

import numpy as np

__all__ = ["Experiment", "run_experiments", "results_to_csv", "make_argument_parser", "make_logger"]

# backend name -> (driver --version, variant directory under backend_pim/)      (experiment.py:386-395)
BACKENDS = {
    "spmm_default": ("spmm", "spmm_default"),
    "spmm_multigroup": ("spmm", "spmm_default"),   # groups_per_rank has no meaning on one GPU: same library
    "spmm_grande": ("grande", "spmm_grande"),
    "spmv_sparseP": ("spmv", "spmv_sparseP"),
}
CPU_BACKENDS = ("cpu", "CPU", None)
_DATA_LINE = re.compile(r"^\[DATA](.*?): (.*)")
_BANNERS = ("-------------------- Repeat", "-------------------- Model")


def _joined(pairs) -> str:
    return "-".join(f"{k}={v}" for k, v in pairs)


@dataclasses.dataclass
class Experiment:
    """Same fields, defaults and derived names as the reference's dataclass (experiment.py:157-181)."""
    dataset: str
    sp_part: int
    ds_part: int
    sp_format: str
    dense_size: int
    dtype: str
    balance: str
    balance_tsklt: str
    nr_tasklets: int
    cg_lock: bool
    cache_size: int
    backend: Optional[str] = None
    merge: str = "block"
    groups_per_rank: Optional[int] = None
    sync: bool = True
    tune: str = "FALSE"
    nr_dpus: Optional[int] = None
    model: Optional[str] = None
    num_layers: Optional[int] = None

    def __post_init__(self):
        if self.backend is None:
            self.backend = self.default_backend

    # -- names ---------------------------------------------------------------------------------------------
    @property
    def default_backend(self):
        """experiment.py:237-242 (these names are not ones ``run`` accepts there either: pass ``backend``)."""
        if self.groups_per_rank not in (None, 1):
            return "backend_pim_multigroup"
        return "backend_pim_grande" if self.ds_part == 0 else "backend_pim_group"

    @property
    def build_params(self):
        keys = ("backend", "sp_format", "dtype", "balance", "balance_tsklt", "nr_tasklets", "cg_lock", "cache_size",
                "merge", "sync")
        return {k: getattr(self, k) for k in keys}

    @property
    def frozen_build_params(self):
        """experiment.py:202-219: the optional keys appear only away from their defaults."""
        p = collections.OrderedDict(backend=f"{self.backend}", spf=f"{self.sp_format}", dtype=f"{self.dtype}",
                                    blnc=f"{self.balance}", blnc_tsklt=f"{self.balance_tsklt}",
                                    nr_tasklets=f"{self.nr_tasklets}", cg_lock=f"{self.cg_lock}",
                                    cache_size=f"{self.cache_size}")
        if self.groups_per_rank not in (None, 1):
            p["gpr"] = f"{self.groups_per_rank}"
        if self.merge != "block":
            p["merge"] = f"{self.merge}"
        if not self.sync:
            p["sync"] = f"{self.sync}"
        return p

    @property
    def frozen_run_params(self):
        """experiment.py:221-235: build keys with cache_size moved behind the run keys."""
        p = self.frozen_build_params
        cache = p.pop("cache_size")
        p["spds"] = f"{self.sp_part}x{self.ds_part}"
        p["dataset"] = f"{self.dataset}"
        p["dense_size"] = f"{self.dense_size}"
        p["nr_dpus"] = f"{self.nr_dpus}"
        p["cache_size"] = cache
        if self.model is not None:
            p["model"] = f"{self.model}"
        if self.num_layers is not None:
            p["num_layers"] = f"{self.num_layers}"
        return p

    def build_path(self, build_root: str):
        return os.path.join(build_root, _joined(self.frozen_build_params.items()))

    def _result_path(self, result_root, ext, failed):
        return os.path.join(result_root, _joined(self.frozen_run_params.items()) + ext + (".failed" if failed else ""))

    def stdout_path(self, result_root: str, failed: bool = False):
        return self._result_path(result_root, ".out", failed)

    def stderr_path(self, result_root: str, failed: bool = False):
        return self._result_path(result_root, ".err", failed)

    def status_at(self, result_root: str):
        """experiment.py:350-356"""
        out = self.stdout_path(result_root)
        if os.path.exists(out):
            return "done"
        return "failed" if os.path.exists(out + ".failed") else "todo"

    # -- build ---------------------------------------------------------------------------------------------
    def build(self, src_root: str, build_root: str, force_rebuild: bool = False, dry_run: bool = False,
              logger: Optional[logging.Logger] = None):
        """Make ``build_path(build_root)/libbackend_pim.so`` exist (experiment.py:277-348 compiles one there).  The shim
        libraries are built once by ``make -C pygim_amd/csrc shims``; this links the variant's into the per-configuration
        directory, building the shims first when they are missing."""
        if self.backend in CPU_BACKENDS or str(self.backend).endswith("@cpu"):
            if logger is not None:
                logger.debug("==> Skipping building a CPU backend")
            return None
        if self.backend not in BACKENDS:
            raise NotImplementedError(self.backend)
        path = self.build_path(build_root)
        target = os.path.join(path, "libbackend_pim.so")
        if os.path.exists(target) and not force_rebuild:
            if logger is not None:
                logger.debug(f"==> Skipping build at {path}")
            return target
        if logger is not None:
            logger.info(f"==> Building for {self}")
        if dry_run:
            return target
        built = os.path.join(os.path.abspath(src_root), "backend_pim", BACKENDS[self.backend][1], "build", "libbackend_pim.so")
        if not os.path.exists(built) or force_rebuild:
            subprocess.check_call(["make", "-C", os.path.join(os.path.abspath(src_root), "pygim_amd", "csrc"), "all"],
                                  stdout=subprocess.DEVNULL)
        if os.path.isdir(path) and force_rebuild:
            shutil.rmtree(path)
        os.makedirs(path, exist_ok=True)
        if os.path.lexists(target):
            os.remove(target)
        os.symlink(built, target)
        return target

    # -- run -----------------------------------------------------------------------------------------------
    def command(self, src_root: str, data_root: str, build_root: str, repeat: int = 3):
        """The argument vector of experiment.py:408-433 (``--sp_part`` / ``--ds_part`` singular, as there)."""
        if self.backend in CPU_BACKENDS:
            version, lib_path = "cpu", None
        elif self.backend in BACKENDS:
            version = BACKENDS[self.backend][0]
            lib_path = os.path.join(self.build_path(build_root), "libbackend_pim.so")
        else:
            raise NotImplementedError(self.backend)
        dataset = "ogbn-proteins" if self.dataset == "ogbnproteins" else self.dataset   # experiment.py:384
        whole_model = self.model is not None and self.num_layers is not None
        script = os.path.join(src_root, "inference.py" if whole_model else "spmm_test.py")
        cmd = [sys.executable, script, f"--dataset={dataset}", f"--datadir={data_root}", f"--sp_format={self.sp_format}",
               f"--data_type={self.dtype}", f"--hidden_size={self.dense_size}", f"--sp_part={self.sp_part}",
               f"--ds_part={self.ds_part}", f"--repeat={repeat}", f"--lib_path={lib_path}", f"--version={version}"]
        if self.nr_dpus is not None:
            cmd.append(f"--nr_dpus={self.nr_dpus}")
        if whole_model:
            cmd += [f"--model={self.model}", f"--num_layers={self.num_layers}"]
        if self.backend == "spmm_multigroup" and self.groups_per_rank is not None:
            cmd.append(f"--group_per_rank={self.groups_per_rank}")   # experiment.py:432-434; accepted and ignored here
        return cmd

    def run(self, src_root: str, data_root: str, build_root: str, result_root: Optional[str] = None, repeat: int = 3,
            silent: bool = False, dry_run: bool = False, logger: Optional[logging.Logger] = None):
        """experiment.py:361-464: run the driver; with ``result_root`` its stdout / stderr land in the named files
        (``.failed`` appended on a non-zero exit, and -- unless ``silent`` -- a RuntimeError after they are written)."""
        # (the reference tests the truth of `tune`, whose default is the non-empty string 'FALSE': experiment.py:176, 402;
        #  here 'FALSE' means off)
        # result files carry the DECLARED parameters (the reference computes its stdout path before it tunes and keeps that name on
        # success, experiment.py:396-405): status_at / run_experiments / the plot scripts look a run up by what was asked for
        declared_out = declared_err = None
        if result_root is not None:
            declared_out = {f: self.stdout_path(result_root, failed=f) for f in (False, True)}
            declared_err = {f: self.stderr_path(result_root, failed=f) for f in (False, True)}
        if self.tune not in (None, False, "", "FALSE", "False"):
            self._apply_tuned_partition(data_root)
        cmd = self.command(src_root, data_root, build_root, repeat)
        if logger is not None:
            logger.info(f"==> {self}")
            logger.debug(f"Running: {cmd}")
        if dry_run:
            return 0
        if result_root is None:
            rc = subprocess.run(cmd).returncode
        else:
            os.makedirs(result_root, exist_ok=True)
            pipe = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            rc = pipe.returncode
            with open(declared_out[rc != 0], "wb") as w:
                w.write(pipe.stdout)
            with open(declared_err[rc != 0], "wb") as w:
                w.write(pipe.stderr)
        if rc != 0:
            message = f"The following command failed with return code {rc}\n==> {' '.join(cmd)}\n"
            if silent:
                if logger is not None:
                    logger.critical(message)
            else:
                raise RuntimeError(message)
        return rc

    def _apply_tuned_partition(self, data_root):
        """experiment.py:402-405 asks the autotuner for (sp_part, ds_part, balance, balance_tsklt, groups_per_rank) among
        sp_ds_set = [(1, 32), (2, 16)]; here the candidates are the (row parts x feature parts) grids over the GPUs of this
        job (WORLD_SIZE, one when not under torch.distributed.run), priced by pygim_amd/autotune.py."""
        from . import autotune

        name = "ogbn-proteins" if self.dataset == "ogbnproteins" else self.dataset
        es = {"INT64": 8, "INT32": 4, "INT16": 2, "INT8": 1, "FLT32": 4, "DBL64": 8}[self.dtype]
        n_gpus = max(int(os.environ.get("WORLD_SIZE", "1")), 1)
        grids = [(r, n_gpus // r) for r in range(1, n_gpus + 1) if n_gpus % r == 0]
        tuned = autotune.autotune_dataset(data_root, name, self.dense_size, grids, elem_bytes=es)
        if tuned[0] is not None:  # (no admissible grid, e.g. more feature parts than columns: keep the split as given)
            self.sp_part, self.ds_part, self.balance, self.balance_tsklt, self.groups_per_rank = tuned

    # -- results -------------------------------------------------------------------------------------------
    def parse_result(self, result_root: str):
        """experiment.py:466-491: per key, the values of one repeat are summed and the repeats averaged; ``repeat``
        counts the banner lines."""
        with open(self.stdout_path(result_root), "r") as reader:
            return parse_stdout(reader)


def parse_stdout(lines):
    values, repeat = collections.defaultdict(list), 0
    for line in lines:
        if line.startswith(_BANNERS):
            repeat += 1
        found = _DATA_LINE.findall(line.strip())
        if found:
            values[found[0][0]].append(float(found[0][1]))
    out = {k: np.asarray(v).reshape(repeat, -1).mean(axis=0).sum(axis=-1) for k, v in values.items()}
    out["repeat"] = repeat
    return out


def results_to_csv(out_dir: str, csv_dir: Optional[str] = None):
    """backend_pim/spmv_sparseP/parse_results.py:24-76 for a directory of ``.out`` files: per file one CSV with a row per
    repeat and an ``avg`` row, the derived column ``pim_time_dense(ms) = pim_time_spmm(ms) - load_sparse_time`` when the
    run logged both (the shims print the per-run timer lines under PYGIM_DATA_LOG=1), and ``average_all.csv`` with one
    line per file.  Returns {file stem: {key: average}}."""
    csv_dir = csv_dir or os.path.join(out_dir, "csv_result")
    os.makedirs(csv_dir, exist_ok=True)
    averages = {}
    for name in sorted(os.listdir(out_dir)):
        if not name.endswith(".out"):
            continue
        cols, repeats = collections.OrderedDict(), 0
        with open(os.path.join(out_dir, name), "r") as reader:
            for line in reader:
                if line.startswith("-------------------- Repeat") or line.startswith("-------------------- Model"):
                    repeats += 1
                if line.startswith("[DATA]") and ": " in line:
                    label, value = line[6:].rstrip("\n").split(": ", 1)
                    cols.setdefault(label, []).append(float(value))
        if repeats == 0:
            continue
        # several lines of one key inside a repeat (one per layer, say) are summed, as parse_result does
        table = collections.OrderedDict((k, np.asarray(v).reshape(repeats, -1).sum(axis=1)) for k, v in cols.items()
                                        if len(v) % repeats == 0)
        if "pim_time_spmm(ms)" in table and "load_sparse_time" in table:
            table["pim_time_dense(ms)"] = table["pim_time_spmm(ms)"] - table["load_sparse_time"]
        keys = list(table)
        with open(os.path.join(csv_dir, name[:-4] + ".csv"), "w") as w:
            w.write(", ".join(["Repeat"] + keys) + "\n")
            for rep in range(repeats):
                w.write(", ".join([str(rep)] + [repr(float(table[k][rep])) for k in keys]) + "\n")
            w.write(", ".join(["avg"] + [repr(float(table[k].mean())) for k in keys]) + "\n")
        averages[name[:-4]] = {k: float(table[k].mean()) for k in keys}
    all_keys = []
    for avg in averages.values():
        all_keys += [k for k in avg if k not in all_keys]
    with open(os.path.join(csv_dir, "average_all.csv"), "w") as w:
        w.write(", ".join(["run"] + all_keys) + "\n")
        for stem, avg in averages.items():
            w.write(", ".join([stem] + [repr(avg[k]) if k in avg else "" for k in all_keys]) + "\n")
    return averages


def run_experiments(args, build_set, experiments, logger: logging.Logger, accept_failures: bool = False, repeat: int = 1):
    """helpers.py:44-103: build what the build set names, then run every experiment that is still to do (a run whose
    ``.out`` exists is done; one whose ``.out.failed`` exists is retried unless failures are accepted)."""
    os.makedirs(args.result_root, exist_ok=True)
    for experiment in build_set:
        experiment.build(src_root=args.src_root, build_root=args.build_root,
                         force_rebuild=getattr(args, "force_rebuild", False), dry_run=args.dry_run, logger=logger)
    settled = ("done", "failed") if accept_failures else ("done",)
    for experiment in experiments:
        status = experiment.status_at(args.result_root)
        if status in settled:
            logger.info(f"==> Skipping {'failed ' if status == 'failed' else ''}{experiment}")
            continue
        experiment.run(src_root=args.src_root, data_root=args.data_root, build_root=args.build_root,
                       result_root=args.result_root, repeat=repeat, silent=getattr(args, "skip_failed", False),
                       dry_run=args.dry_run, logger=logger)


def make_argument_parser(result_name):
    """helpers.py:13-41, the ``run`` and ``plot`` actions (``migrate`` renamed result files of an older naming scheme
    that never existed here)."""
    import argparse

    root = argparse.ArgumentParser()
    root.add_argument("--log_file", type=str, default="./logs.log")
    root.add_argument("--append", action="store_true")
    root.add_argument("--verbose", action="store_true")
    root.add_argument("--dry_run", action="store_true")
    root.add_argument("--figure_root", type=str, default="./")
    actions = root.add_subparsers(dest="action")
    run = actions.add_parser("run")
    run.add_argument("--force_rebuild", action="store_true")
    run.add_argument("--skip_failed", action="store_true")
    run.add_argument("--src_root", type=str, default="./")
    run.add_argument("--data_root", type=str, default="./data")
    run.add_argument("--build_root", type=str, default="./build")
    run.add_argument("--result_root", type=str, default=f"./results/{result_name}")
    plot = actions.add_parser("plot")
    plot.add_argument("--result_root", type=str, default=f"./results/{result_name}")
    return root


def make_logger(name, args):
    """helpers.py:139-153"""
    logger = logging.getLogger(name)
    logger.propagate = False
    logger.setLevel(logging.DEBUG)
    console = logging.StreamHandler(stream=sys.stdout)
    console.setLevel(logging.DEBUG if args.verbose else logging.INFO)
    logger.addHandler(console)
    if args.log_file:
        to_file = logging.FileHandler(filename=args.log_file, mode="a" if args.append else "w")
        to_file.setLevel(logging.DEBUG)
        logger.addHandler(to_file)
    return logger
