"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/pdecnn.h declares; the ctypes table in _lib.py covers exactly that set.  No compute
calls here (argument validation only, which runs on the host)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "pdecnn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pde_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_something():
    names = declared_functions()
    assert "pde_adi_forward" in names and "pde_adi_backward" in names and len(names) >= 15


def test_library_exports_every_declared_symbol():
    from cnn_with_pde_amd import _lib
    assert os.path.isfile(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in pdecnn.h but not exported"


def test_ctypes_table_matches_header():
    from cnn_with_pde_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    _lib.load()                                   # sets restype/argtypes for every one


def test_struct_layout_matches_header():
    from cnn_with_pde_amd import _lib
    assert C.sizeof(_lib.PdeSweep) == 16
    assert C.sizeof(_lib.PdeAdiDesc) == 9 * 4 + 16 * _lib.PDE_MAX_SWEEPS


def test_argument_validation_without_gpu():
    """Entry points reject bad descriptors on the host, before touching the device."""
    from cnn_with_pde_amd import _lib
    lib = _lib.load()
    d = _lib.PdeAdiDesc()
    d.B, d.C, d.N, d.num_sweeps = 1, 1, 130, 3         # longer than any path holds (PDE_MAX_N_GENERIC = 128)
    assert lib.pde_adi_forward_workspace_bytes(C.byref(d)) == 0
    assert lib.pde_adi_forward(C.byref(d), None, None, None, None, None, None, None, None, None, None, 0, None) == -2
    d.N = 30                                           # no fused kernels: served by the any-size path ...
    assert lib.pde_adi_line_length_path(30) == 2 and lib.pde_adi_line_length_path(32) == 1 and lib.pde_adi_line_length_path(1) == 0
    assert lib.pde_adi_forward_workspace_bytes(C.byref(d)) > 0
    assert lib.pde_adi_forward(C.byref(d), None, None, None, None, None, None, None, None, None, None, 0, None) == -1
    assert lib.pde_adi_steps_workspace_bytes(C.byref(d), 3) == 0      # ... which has no per-step / one-launch entry points
    assert lib.pde_adi_small_supported(C.byref(d), 3) == 0
    d.N, d.num_sweeps = 32, _lib.PDE_MAX_SWEEPS + 1
    assert lib.pde_adi_forward(C.byref(d), None, None, None, None, None, None, None, None, None, None, 0, None) == -3
    d.num_sweeps = 3
    assert lib.pde_adi_forward_workspace_bytes(C.byref(d)) > 0
    assert lib.pde_adi_forward(C.byref(d), None, None, None, None, None, None, None, None, None, None, 0, None) == -1   # null pointers
    assert lib.pde_channel_mix_forward(0, 3, 16, 0, None, None, None, None) == -1
    assert lib.pde_explicit5_forward(1, 1, 8, 6, 0, None, None, None, 0.01, 1e-6, 0.15, 0.1, 1, None, None, None) == -1
    assert lib.pde_version().startswith(b"pdecnn-hip")


def test_native_host_extension_loads_and_is_linked_against_the_library():
    """csrc/host_ext.cpp: host glue over the same C ABI; importing it runs no kernel."""
    from cnn_with_pde_amd import _lib as L
    H = L.host_ext()
    assert H is not None and callable(H.adi)
    assert H.abi_version() == L.load().pde_version().decode()
