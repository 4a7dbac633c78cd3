"""MI355X-native GNN aggregation backend with PyGim's ``backend_pim`` surface.

Layout:
  csrc/            hand-written HIP kernels (gfx950) + the C ABI (include/pygim_hip.h)
  _lib.py          ctypes binding of that ABI (fails loudly when the .so is missing)
  pim_ops.py       ``torch.ops.pim_ops`` registration (the reference's custom-op names)
  backend_pim/     ``prepare_pim_*`` / ``SparseTensorCOO.mul`` wrappers (reference surface)
  sparse_tensor.py stand-in for torch_sparse.SparseTensor when that package is absent
  synth.py         seeded synthetic graphs in the shapes BASELINE.json names
  dist.py          sp_parts / ds_parts across the GPUs of one node (torch.distributed/RCCL)
"""
__version__ = "0.1.0"
