"""CPU tests of the drop-in boundary: the C-ABI libraries load and export every symbol include/*.h declares;
without a GPU the product path fails loudly instead of falling back."""
import ctypes as C
import subprocess

import numpy as np
import pytest


def test_libraries_load_and_export_every_declared_symbol(built):
    from petsc_amd import _lib
    hx, ks = _lib.load()
    d = _lib.declared_functions()
    assert len(d["hipx"]) >= 60 and len(d["ksp"]) >= 15
    for name, _, _ in d["hipx"]:
        assert hasattr(hx, name), name
    for name, _, _ in d["ksp"]:
        assert hasattr(ks, name), name
    # nm view: exported as unmangled C symbols
    syms = subprocess.check_output(["nm", "-D", "--defined-only", _lib.lib_paths()["hipx"]], text=True)
    for name, _, _ in d["hipx"]:
        assert (" T " + name + "\n") in syms, name


def test_product_libraries_do_not_link_the_oracle(built):
    from petsc_amd import _lib
    for p in _lib.lib_paths().values():
        out = subprocess.check_output(["ldd", p], text=True) + subprocess.check_output(["nm", "-D", p], text=True)
        assert "liboracle" not in out and "orc_" not in out


def test_no_gpu_means_loud_failure_not_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from petsc_amd import _lib
    hx, ks = _lib.load()
    assert hx.hipxInit(0) != 0
    assert b"no CPU fallback" in hx.hipxGetErrorString()
    x = np.zeros(4)
    r = C.c_double()
    assert hx.hipxVecDot(x.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), 4, C.byref(r)) == 58  # PETSC_ERR_ORDER: not initialised
    with pytest.raises(_lib.HipxError):
        _lib.init(0)


def test_hardware_probes_are_built_with_the_libraries(built):
    """scripts/diag/*.hip (request-path cost, wave placement: what profiles/r02_request_path_probe.txt and
    r02_wave_placement.txt come from) are compiled for gfx950 by build(), next to their sources."""
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    srcs = glob.glob(os.path.join(root, "scripts", "diag", "*.hip"))
    assert len(srcs) >= 2
    for s in srcs:
        exe = s[:-4]
        assert os.path.exists(exe) and os.access(exe, os.X_OK), exe
        assert b"gfx950" in open(exe, "rb").read()  # the embedded code object's target
