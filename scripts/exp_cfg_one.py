#!/usr/bin/env python3
"""ONE BASELINE configuration outside bench.py, for rocprofv3 (kernel trace or PMC passes): exp_cfg_one.py c3 | c5 | c4
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
elif which == "c5":
    r = bench_configs.config5_papers_slices(dev, 0, steps=2)
else:
    r = bench_configs.config4_gcn_one_gpu(dev, 256, steps=2)
print(json.dumps(r))
_lib.release()
