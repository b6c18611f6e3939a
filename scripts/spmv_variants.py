"""GPU micro-benchmark: time every SpMV kernel geometry / load policy on the BASELINE matrices (HIP events, 50 launches).
  python scripts/spmv_variants.py [n] [stencil]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from petsc_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
st = int(sys.argv[2]) if len(sys.argv) > 2 else 7
hx = _lib.init(0)
_, ks = _lib.load()
N = n ** 3
ai, aj, aa = bench.assemble(ks, st, n, 0, N)
A = _lib.mat_create_csr(N, N, ai, aj, aa)
X = _lib.DVec(N, 1.0 + (np.arange(N) % 17) / 17.0)
Y = _lib.DVec(N)
bytes_ = 12 * len(aj) + 4 * (N + 1) + 16 * N
ref = None
for v in range(1, 13):
    _lib.chk(hx.hipxMatSetSpMVVariant(A, v))
    for _ in range(5):
        _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    _lib.chk(hx.hipxProfileSpMV(1))
    for _ in range(50):
        _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    cnt, ms = C.c_int(), C.c_double()
    _lib.chk(hx.hipxProfileSpMVGet(C.byref(cnt), C.byref(ms)))
    _lib.chk(hx.hipxProfileSpMV(0))
    y = Y.get()
    if ref is None:
        ref = y
    t = ms.value / cnt.value
    print("variant %2d  cfg %d nt %d : %.4f ms  %.1f GB/s  (%.1f%% of 8 TB/s)  identical=%s" % (v, (v - 1) // 2, (v - 1) % 2, t, bytes_ / t / 1e6, bytes_ / t / 1e6 / 80, np.array_equal(y, ref)))
