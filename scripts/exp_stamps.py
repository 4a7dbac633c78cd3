#!/usr/bin/env python3
"""Where a workgroup of the code-stream kernel spends its life (round 6): k_lds_code8_f32 run as its measurement build (tunable lds_stamp), which
stamps the 100 MHz clock when a wave starts, enters its instruction stream, leaves it, and has stored its rows.  Prints, per case, the kernel's
span, the workgroups' start / end distribution (rounds, tail), and the phases: set-up (zeroing, pointer arithmetic), stream, store stage.
usage: exp_stamps.py [tunables] [cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from pygim_amd import _lib, synth
from pygim_amd.bench_plans import nnz_balanced_row_split

dev = torch.device("cuda", 0)
_lib.init_ranks(1)
tun = sys.argv[1] if len(sys.argv) > 1 else ""
for kv in filter(None, tun.split(",")):
    k, v = kv.split("=")
    assert _lib.set_tunable(k, int(v)) != -1, k
want = (sys.argv[2] if len(sys.argv) > 2 else "full,r8").split(",")
print("#", tun or "defaults", flush=True)
n, nnz, dmax = synth.SHAPES["reddit"]
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
rp_cpu = rowptr.cpu()
CASES = {"full": (1, 256), "r8": (8, 256), "h64": (1, 64), "h32": (1, 32), "g24": (2, 64), "g42": (4, 128), "r2": (2, 256), "r4": (4, 256), "h128": (1, 128)}
MAXB = 4096
stamps = torch.zeros((MAXB, 8, 4), dtype=torch.int64, device=dev)
for name in want:
    frac, h = CASES[name]
    top = nnz_balanced_row_split(rp_cpu, frac)[1]
    m = int(rp_cpu[top])
    x = synth.features(n, h, torch.float32, seed=1, device=dev, kind="uniform")
    out = torch.empty((top, h), dtype=torch.float32, device=dev)
    hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [top], [n], [m], [1], [h], h)
    for _ in range(3):
        _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
    torch.cuda.synchronize()
    ref = out.clone()
    stamps.zero_()
    _lib.set_tunable("lds_stamp", stamps.data_ptr())
    _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
    torch.cuda.synchronize()
    _lib.set_tunable("lds_stamp", 0)
    assert torch.equal(out, ref)
    geo, plan = _lib.group_lds_geometry(hd), _lib.group_lds_plan(hd)
    _lib.group_free(hd)
    st = stamps.cpu().numpy().astype(np.float64)
    used = st[:, :, 0].min(axis=1) > 0
    st = st[used] * 0.01            # microseconds (100 MHz)
    t0 = st[:, :, 0].min()
    st -= t0
    nb = st.shape[0]
    wg_start, wg_stream0, wg_stream1, wg_end = st[:, :, 0].min(1), st[:, :, 1].max(1), st[:, :, 2].max(1), st[:, :, 3].max(1)
    span = wg_end.max()
    setup = (st[:, :, 1] - st[:, :, 0])
    stream = (st[:, :, 2] - st[:, :, 1])
    store = (st[:, :, 3] - st[:, :, 2])
    life = wg_end - wg_start
    first_round = np.sort(wg_start)[:256]
    print(f"{name}: {nb} workgroups (tiles {plan['tiles']}, splits {geo['col_splits']}); kernel span {span:8.1f} us")
    print(f"   a wave's set-up  : mean {setup.mean():7.2f}  max {setup.max():7.2f} us")
    print(f"   a wave's stream  : mean {stream.mean():7.2f}  p5 {np.percentile(stream, 5):7.2f}  p95 {np.percentile(stream, 95):7.2f}  max {stream.max():7.2f} us")
    print(f"   a wave's store   : mean {store.mean():7.2f}  p95 {np.percentile(store, 95):7.2f}  max {store.max():7.2f} us")
    print(f"   workgroup life   : mean {life.mean():7.2f}  min {life.min():7.2f}  max {life.max():7.2f} us;  waves of one workgroup leave the stream within "
          f"{(st[:, :, 2].max(1) - st[:, :, 2].min(1)).mean():6.2f} us of each other (mean), max {(st[:, :, 2].max(1) - st[:, :, 2].min(1)).max():6.2f}")
    print(f"   first 256 starts : {first_round.min():6.2f} .. {first_round.max():6.2f} us;  last workgroup starts at {wg_start.max():8.1f}, ends at {wg_end.max():8.1f};  "
          f"earliest end {wg_end.min():8.1f}; idle CU-time at the tail (256 CUs x span - sum of lives) {256 * span - life.sum():10.0f} us = {100 * (1 - life.sum() / (256 * span)):4.1f} %")
_lib.release()
