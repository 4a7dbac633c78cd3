import sys, time
sys.path.insert(0, '/root/repo')
import torch
from pygim_amd import _lib, synth
dev = torch.device("cuda", 0)
_lib.init_ranks(1)
n, nnz, dmax = synth.SHAPES["reddit"]
rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
for hs in (1, 0, 1):
    _lib.set_tunable("lds_half_split", hs)
    for dt, code in (("FLT32", _lib.FLT32), ("INT32", _lib.INT32)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hd = _lib.group_create(_lib.CSR, code, [rowptr.data_ptr()], [col.data_ptr()], None, [n], [n], [nnz], [1], [32], 32)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) * 1e3
        print(f"lds_half_split={hs} {dt} h=32: group_create {t:.1f} ms; code {_lib.group_lds_code(hd)}; {_lib.group_lds_note(hd)[:140]}", flush=True)
        _lib.group_free(hd)
