"""PCSOR's plane march between other kernels: one symmetric sweep on 27-pt 256^3 timed by events around each call, (a) back to back, (b) with memory-bound
vector work (k x VecAXPY over 134 MB vectors) between the calls, as inside GMRES.   python scripts/sor_box_context.py [naxpy]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from petsc_amd import _lib  # noqa: E402

hx = _lib.init(0)
_, ks = _lib.load()
naxpy = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 256
N = n ** 3
ai, aj, aa = bench.assemble(ks, 27, (n, n, n), 0, N)
A = _lib.mat_create_csr(N, N, ai, aj, aa)
del ai, aj, aa
b = 1.0 + (np.arange(N) % 17) / 17.0
B, X = _lib.DVec(N, b), _lib.DVec(N)
W = [_lib.DVec(N, b) for _ in range(8 if naxpy else 0)]  # (naxpy = 0: no other vectors at all -- the timings depend on what else is allocated)
os.environ["HIPX_SOR_MODE"] = "box"
e0, e1 = C.c_void_p(), C.c_void_p()
_lib.chk(hx.hipxEventCreate(C.byref(e0)))
_lib.chk(hx.hipxEventCreate(C.byref(e1)))
for _ in range(2):
    _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 12 | 16, 0.0, 1, 1, X.ptr))
for label, between in (("back to back", 0), ("%d VecAXPY between the calls" % naxpy, naxpy)):
    tot = 0.0
    reps = 10
    for _ in range(reps):
        for q in range(between):
            _lib.chk(hx.hipxVecAXPY(W[(q + 1) % 8].ptr, C.c_double(1e-9), W[q % 8].ptr, N))
        _lib.chk(hx.hipxEventRecord(e0))
        _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 12 | 16, 0.0, 1, 1, X.ptr))
        _lib.chk(hx.hipxEventRecord(e1))
        ms = C.c_float()
        _lib.chk(hx.hipxEventElapsedMs(e0, e1, C.byref(ms)))
        tot += ms.value
    print("PCApply_SOR 27-pt 256^3, %-32s %.3f ms per application" % (label + ":", tot / reps), flush=True)
for reps in (1, 2, 5):
    _lib.chk(hx.hipxEventRecord(e0))
    for _ in range(reps):
        _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 12 | 16, 0.0, 1, 1, X.ptr))
    _lib.chk(hx.hipxEventRecord(e1))
    ms = C.c_float()
    _lib.chk(hx.hipxEventElapsedMs(e0, e1, C.byref(ms)))
    print("one pair of events around %d applications: %.3f ms per application" % (reps, ms.value / reps), flush=True)
