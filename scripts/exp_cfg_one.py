#!/usr/bin/env python3
"""ONE BASELINE configuration outside bench.py, for rocprofv3 (kernel trace or PMC passes): exp_cfg_one.py c3 | c5 | c5a | c5b | c4
(pygim_amd/bench_configs.py: configs[2] products COO INT32, configs[4] papers100M per-GPU shares, configs[3] one-GPU GCN)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pygim_amd import _lib, bench_configs

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
_lib.init_ranks(1)
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
if which == "c3":
    r = bench_configs.config3_products_coo(dev, 0, 256, steps=3)
    products = 2 + 3 + 3            # warm-up + timed + the kernel-event pass (bench_configs._median_ms / _kernel_ms)
elif which in ("c5", "c5a", "c5b"):
    only = {"c5a": "feature_split_1x8", "c5b": "grid_2x4"}.get(which)
    r = bench_configs.config5_papers_slices(dev, 0, steps=2, only=only)
    products = (1 + 2 + 2) * (1 if only else 2)
else:
    r = bench_configs.config4_gcn_one_gpu(dev, 256, steps=2)
    products = 3 * (1 + 1 + 2 + 2)
print(json.dumps(r))
print("PRODUCTS", products)
_lib.release()
