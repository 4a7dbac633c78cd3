"""Python operator surface of the reference (backend_pim/{spmm,grande,spmv}.py), on HIP."""
from .spmm import prepare_pim_spmm, pim_spmm  # noqa: F401
from .grande import prepare_pim_spmm_grande, pim_spmm_grande  # noqa: F401
from .spmv import prepare_pim_spmv, pim_spmv  # noqa: F401
