"""libbackend_pim.so (TORCH_LIBRARY shim over the C ABI) loaded the way the reference loads its
backend: torch.ops.load_library(args.lib_path).  One subprocess per variant."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def run(variant, *extra):
    r = subprocess.run([sys.executable, os.path.join(HERE, "shim_driver.py"), variant, *extra], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("variant", ["spmm", "grande", "spmv"])
def test_shim_registers_reference_ops(variant):
    import torch
    if torch.cuda.is_available():
        pytest.skip("no-device behaviour")
    out = run(variant)
    assert "OK no-device" in out and (variant == "spmv" or "OK mtx" in out)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["spmm", "grande", "spmv"])
def test_shim_end_to_end_gpu(variant):
    assert "OK gpu" in run(variant, "gpu")
