"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
that include/gffx_hip.h declares; without a GPU the compute entry points fail loudly."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gffx_amd import _ffi, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "gffx_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gffx_hip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = C.CDLL(_ffi.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), "missing export: " + name
    assert sorted(_ffi.SIGNATURES) == declared  # the ctypes table covers the whole header


def test_abi_version_and_device_count_do_not_need_a_gpu():
    L = _ffi.lib()
    assert L.gffx_hip_abi_version() == 1
    assert L.gffx_hip_device_count() >= 0


def test_bad_arguments_are_reported_not_crashed():
    L = _ffi.lib()
    h = C.c_void_p()
    co = np.array([1, 2], dtype=np.uint32)  # chr_offsets[0] != 0
    z = np.zeros(2, dtype=np.uint32)
    p = lambda a: a.ctypes.data_as(_ffi.u32p)  # noqa: E731
    rc = L.gffx_hip_index_create(1, p(co), p(z), p(z), p(z), 0, C.byref(h))
    assert rc == -1 and b"chr_offsets" in L.gffx_hip_last_error()
    assert L.gffx_hip_batch_create(None, 10, C.byref(h)) == -1
    assert L.gffx_hip_batch_run(None, 2, 0, 0, 0) == -1


@pytest.mark.skipif(engine.device_count() > 0, reason="only meaningful without a GPU")
def test_no_cpu_fallback_without_a_gpu():
    with pytest.raises(_ffi.GffxHipError) as ei:
        engine.TreeIndexData.from_roots([0, 1], [5], [9], [0])
    assert ei.value.code == -2  # GFFX_E_NO_DEVICE
