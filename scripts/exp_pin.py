import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pygim_amd import _lib, synth
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES["ogbn-products"]
x = synth.features(n, 256, torch.int32, seed=0, device=dev)
out = torch.empty((n, 256), dtype=torch.int32, device=dev)
for p_in in (0.8, 0.95, 1.0):
    rowptr, col = synth.make_sbm(n, nnz, dmax, 1200, p_in=p_in, seed=0, device=dev, shuffle=False)
    for loc in (0, 1):
        _lib.set_tunable("panel_locality", loc)
        hd = _lib.group_create(_lib.CSR, _lib.INT32, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [256], 256)
        for _ in range(2): _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
        torch.cuda.synchronize()
        ts = []
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
        print(f"p_in {p_in} locality {loc}: {min(ts):.3f} ms  tiles {_lib.group_lds_tiles(hd)} plan {_lib.group_lds_plan(hd)['tiles']}", flush=True)
        _lib.group_free(hd)
