#!/usr/bin/env python3
"""Does the host-operand pipeline's time depend on what the process did before?  (bench.py measured 11.9 ms per call after the BASELINE-config legs; a fresh
process 8.2.)  end_to_end_cpu_tensors first, then after each of the legs bench.py runs before it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from pygim_amd import _lib, bench_configs, synth

dev = torch.device("cuda", 0)
_lib.init_ranks(1)
stream = torch.cuda.current_stream().cuda_stream
n, nnz, dmax = synth.SHAPES["reddit"]
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)


def cpu_now():
    try:   # field 39 of /proc/self/stat: the processor this thread last ran on
        return int(open("/proc/thread-self/stat").read().rsplit(")", 1)[1].split()[36])
    except Exception:  # noqa: BLE001
        return -1


def e2e(tag):
    r = bench_configs.end_to_end_cpu_tensors(rowptr, col, n, 256)
    print(f"{tag:40s} pipelined {r['ms_per_mul']:.2f} ms [{'direct stores' if r['direct_stores'] else 'copies'}; calls: {r['ms_each_call']}] (up {r['ms_until_x_is_up']:.2f} + {r['ms_last_product_after_that']:.2f} + {r['ms_last_download_after_that']:.2f})  serial {r['ms_per_mul_serial']:.2f}  "
          f"cpu of this thread: {cpu_now()}", flush=True)


e2e("fresh process")
e2e("again")
for name, fn in (("after configs[2] (products COO)", lambda: bench_configs.config3_products_coo(dev, stream, 256)),
                 ("after configs[3] (GCN one GPU)", lambda: bench_configs.config4_gcn_one_gpu(dev, 256)),
                 ("after configs[4] (papers100M shares)", lambda: bench_configs.config5_papers_slices(dev, stream))):
    t0 = time.perf_counter()
    fn()
    torch.cuda.empty_cache()
    print(f"  ({name[6:]}: {time.perf_counter() - t0:.1f} s)", flush=True)
    e2e(name)
    e2e(name + ", again")
