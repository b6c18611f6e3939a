#!/usr/bin/env python
"""Stand-alone timing of the vector kernels of the fused CG iteration (HIP events over repeated launches), N = 256^3 by default:
GB moved / time for each -- the by-kernel roofline numbers outside the solver loop, and the harness for kernel-shape experiments
(HIPX_CG_FUSED_U2 / _SWEEP / _BLOCKS, HIPX_RED_BLOCKS)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from petsc_amd import _lib  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 ** 3
    reps = 200
    hx = _lib.init(0)
    rng = np.random.default_rng(0)
    V = [_lib.DVec(n, rng.standard_normal(n)) for _ in range(5)]
    X, R, Z, P, W = V
    scal = _lib.DVec(8, np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0]))
    e0, e1 = C.c_void_p(), C.c_void_p()
    _lib.chk(hx.hipxEventCreate(C.byref(e0)))
    _lib.chk(hx.hipxEventCreate(C.byref(e1)))
    sp = scal.ptr.value

    def timeit(name, f, nbytes):
        for _ in range(10):
            f()
        _lib.chk(hx.hipxEventRecord(e0))
        for _ in range(reps):
            f()
        _lib.chk(hx.hipxEventRecord(e1))
        ms = C.c_float()
        _lib.chk(hx.hipxEventElapsedMs(e0, e1, C.byref(ms)))
        us = 1e3 * ms.value / reps
        print("%-44s %8.1f us  %7.2f TB/s  (%d MB)" % (name, us, nbytes / us / 1e6, nbytes // 10 ** 6), flush=True)

    def fused():  # C(i): r -= a w ; sums -- constant diagonal, device scalars
        _lib.chk(hx.hipxCGFusedUpdateBegin(None, R.ptr, None, P.ptr, W.ptr, None, 0.5, C.c_void_p(sp), C.c_void_p(sp + 8), n, 3, C.c_void_p(sp + 16)))

    def aypx():  # A(i): p = r dconst + b p ; x += a p
        _lib.chk(hx.hipxCGAypxAxpyDev(P.ptr, None, R.ptr, 0.5, X.ptr, C.c_void_p(sp), C.c_void_p(sp + 8), C.c_void_p(sp + 24), n))

    d = C.c_double()
    timeit("cg_fused_kernel (r, w -> r; 2 sums)", fused, 24 * n)
    timeit("cg_aypx_axpy_kernel (p, r, x -> p, x)", aypx, 40 * n)
    timeit("hipxVecDot (blocking)", lambda: _lib.chk(hx.hipxVecDot(X.ptr, W.ptr, n, C.byref(d))), 16 * n)
    timeit("hipxVecAXPY", lambda: _lib.chk(hx.hipxVecAXPY(Z.ptr, 0.5, W.ptr, n)), 24 * n)
    timeit("hipxVecCopy", lambda: _lib.chk(hx.hipxVecCopy(W.ptr, Z.ptr, n)), 16 * n)
    _lib.chk(hx.hipxDeviceSynchronize())


if __name__ == "__main__":
    main()
