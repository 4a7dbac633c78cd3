import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pygim_amd import _lib, synth
from pygim_amd.bench_plans import nnz_balanced_row_split
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES["reddit"]
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
rp_cpu = rowptr.cpu()
def timed(hd, x, out):
    for _ in range(2): _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(5):
        a.record(); _lib.spmm_run_group(hd, [x.data_ptr()], out.data_ptr(), 0); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
_lib.set_tunable("lds_mode", 1)
for frac, h, rt in ((16, 256, 1), (16, 256, 0), (16, 64, 0), (64, 64, 0), (8, 256, 0), (3, 256, 1)):
    _lib.set_tunable("lds_round_tiles", rt)
    top = nnz_balanced_row_split(rp_cpu, frac)[1]
    m = int(rp_cpu[top])
    x = synth.features(n, h, torch.float32, seed=0, device=dev)
    out = torch.empty((top, h), dtype=torch.float32, device=dev)
    hd = _lib.group_create(_lib.CSR, _lib.FLT32, [rowptr.data_ptr()], [col.data_ptr()], None, [top], [n], [m], [1], [h], h)
    t = timed(hd, x, out); pl = _lib.group_lds_plan(hd)
    print(f"rows 1/{frac} h={h} round_tiles={rt}: {t:.3f} ms  tiles {pl['tiles']} x {(h+63)//64} slices = {pl['tiles']*((h+63)//64)} workgroups", flush=True)
    _lib.group_free(hd)
