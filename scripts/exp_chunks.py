#!/usr/bin/env python3
"""Cost of cutting one rank's row block into K pieces (needed to overlap the all-gather with compute):
sequential pieces on one stream vs pieces alternating over two streams."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import _lib, synth
sys.path.insert(0, ROOT)
from bench import nnz_balanced_row_split

dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES["reddit"]
h = 256
frac = int(sys.argv[1]) if len(sys.argv) > 1 else 1   # emulate a rank holding 1/frac of the rows
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
x = synth.features(n, h, torch.float32, seed=0, device=dev)
rp_cpu = rowptr.cpu()
top = nnz_balanced_row_split(rp_cpu, frac)[1]
for K, nstreams in ((1, 1), (2, 1), (4, 1), (2, 2), (4, 2), (4, 4)):
    b = nnz_balanced_row_split(rp_cpu[: top + 1], K)
    hs, outs, keep = [], [], []
    for c in range(K):
        lo, hi = int(rp_cpu[b[c]]), int(rp_cpu[b[c + 1]])
        rp_c = (rowptr[b[c]:b[c + 1] + 1] - lo).contiguous(); cc = col[lo:hi].contiguous(); keep += [rp_c, cc]
        hs.append(_lib.group_create(_lib.CSR, _lib.FLT32, [rp_c.data_ptr()], [cc.data_ptr()], None, [b[c + 1] - b[c]], [n], [hi - lo], [1], [h], h))
        outs.append(torch.empty((b[c + 1] - b[c], h), dtype=torch.float32, device=dev))
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    main = torch.cuda.current_stream()
    def step():
        if nstreams == 1:
            for c in range(K):
                _lib.spmm_run_group(hs[c], [x.data_ptr()], outs[c].data_ptr(), main.cuda_stream, x_unchanged=c > 0)
            return
        # piece 0 packs X on the main stream; the other pieces wait for it, then run on side streams
        _lib.spmm_run_group(hs[0], [x.data_ptr()], outs[0].data_ptr(), main.cuda_stream)
        ev = torch.cuda.Event(); ev.record(main)   # (conservative: after piece 0 entirely)
        for c in range(1, K):
            s = streams[c % nstreams]
            s.wait_event(ev)
            _lib.spmm_run_group(hs[c], [x.data_ptr()], outs[c].data_ptr(), s.cuda_stream, x_unchanged=True)
        for s in streams:
            main.wait_stream(s)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(6):
        a.record(); step(); e.record(); e.synchronize(); ts.append(a.elapsed_time(e))
    print(f"rows 1/{frac}: K={K} streams={nstreams}: {min(ts):.3f} ms", flush=True)
    for hd in hs:
        _lib.group_free(hd)
