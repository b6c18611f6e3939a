#!/usr/bin/env python
"""VecMDot / VecMAXPY on the GMRES(30) basis of a 16.8 M-row system: time and HBM rate as a function of the stride between the basis vectors of one slab
(VecDuplicateVecs_Seq_GEMV, bvec2.c:670: the vectors of a Krylov basis are one allocation) -- does a stride that is a power of two (n = 2^24 doubles = 128 MiB)
make the 18-31 streams of one launch meet in the same HBM channels?  HIPX_MDOT_GROUP=0|4|8 selects the mdot_wide_kernel form (read once per process)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from petsc_amd import _lib  # noqa: E402

hx = _lib.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 ** 3
e0, e1 = C.c_void_p(), C.c_void_p()
_lib.chk(hx.hipxEventCreate(C.byref(e0)))
_lib.chk(hx.hipxEventCreate(C.byref(e1)))
print("HIPX_MDOT_GROUP =", os.environ.get("HIPX_MDOT_GROUP", "(default)"), " n =", n)
for pad in (0, 64, 520, 4096 + 72, 65536 + 328, 262144 + 1096):
    ld = n + pad
    slab = _lib.DVec(ld * 33)
    _lib.chk(hx.hipxVecSet(slab.ptr, ld * 33, 0.5))
    X = slab.offset(0)
    for nv in (17, 30):
        ptrs = (C.c_void_p * nv)(*[slab.ptr.value + 8 * ld * (k + 1) for k in range(nv)])
        res = (C.c_double * nv)()
        al = (C.c_double * nv)(*[1e-3 * (k + 1) for k in range(nv)])
        out = {}
        for name in ("mdot", "maxpy"):
            for rep in range(2):
                if rep:
                    _lib.chk(hx.hipxEventRecord(e0))
                for _ in range(10 if rep else 2):
                    if name == "mdot":
                        _lib.chk(hx.hipxVecMDot(X, nv, ptrs, n, res))
                    else:
                        _lib.chk(hx.hipxVecMAXPY(C.c_void_p(slab.ptr.value + 8 * ld * 32), nv, al, ptrs, n))
                if rep:
                    _lib.chk(hx.hipxEventRecord(e1))
            ms = C.c_float()
            _lib.chk(hx.hipxEventElapsedMs(e0, e1, C.byref(ms)))
            t = ms.value / 10
            byts = 8 * n * ((nv + 1) if name == "mdot" else (nv + 2))
            out[name] = (t, byts / t / 1e9)
        print("pad %7d doubles  nv %2d   mdot %.3f ms %5.2f TB/s   maxpy %.3f ms %5.2f TB/s" % (pad, nv, out["mdot"][0], out["mdot"][1] / 1e3 * 1e3 / 1e3, out["maxpy"][0], out["maxpy"][1] / 1e3))
    slab.free()
