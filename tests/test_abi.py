"""The C-ABI library: it loads, exports every symbol include/pygim_hip.h declares, and fails
loudly (no CPU fallback) when there is no HIP device.  No compute calls here."""
import os
import re

import pytest
import torch

from pygim_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pygim_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pygim_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/pygim_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == syms, "pygim_amd/_lib.py EXPORTS out of sync with the header"


def test_header_cites_the_reference_interface():
    text = open(os.path.join(ROOT, "include", "pygim_hip.h")).read()
    for cite in ("spmm_default/pytorch_api.cpp:204-243", "spmm_default/pytorch_api.cpp:248-280",
                 "spmm_grande/pytorch_api.cpp:269-321", "spmv_sparseP/pytorch_api.cpp:231-266"):
        assert cite in text


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-device behaviour")
def test_no_device_fails_loudly():
    with pytest.raises(_lib.PygimError) as e:
        _lib.init_ranks(1)
    assert e.value.code == _lib.ERR_NO_DEVICE
    assert not _lib.is_initialized()
    import numpy as np
    rp = np.zeros(2, np.int32)
    with pytest.raises(_lib.PygimError) as e:
        _lib.group_create(_lib.CSR, _lib.INT32, [rp.ctypes.data], [rp.ctypes.data], None, [1], [1], [0], [1], [4], 4)
    assert e.value.code == _lib.ERR_NO_DEVICE
    from pygim_amd import pim_ops
    pim_ops.load("spmm")
    with pytest.raises(Exception):
        torch.ops.pim_ops.dpu_init_ranks(1)


def test_missing_extension_is_an_import_error(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpygim_hip.so")
    with pytest.raises(ImportError):
        _lib.lib()


def test_introspection_entry_points_reject_bad_arguments():
    """the plan introspection added in round 4 (pygim_group_lds_geometry / _note): an unknown handle is an error with a message, never
    a crash; a zero-capacity buffer is refused (no device needed: the handle table is empty either way)"""
    import ctypes

    L = _lib.lib()
    out = (ctypes.c_int64 * 8)()
    assert L.pygim_group_lds_geometry(ctypes.c_int64(123456789), out) != 0
    assert b"handle" in L.pygim_last_error()
    buf = ctypes.create_string_buffer(16)
    assert L.pygim_group_lds_note(ctypes.c_int64(123456789), buf, ctypes.c_int64(16)) != 0
    assert L.pygim_group_lds_note(ctypes.c_int64(123456789), buf, ctypes.c_int64(0)) != 0
    assert _lib.set_tunable("lds_code_boundary", 0) >= 0 and _lib.set_tunable("lds_fail", 0) == 0 and _lib.set_tunable("no_such_knob", 1) == -1
