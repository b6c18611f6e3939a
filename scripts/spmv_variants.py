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
ai, aj, aa = bench.assemble(ks, st, (n, n, n), 0, N)
A = _lib.mat_create_csr(N, N, ai, aj, aa)
X = _lib.DVec(N, 1.0 + (np.arange(N) % 17) / 17.0)
Y = _lib.DVec(N)
bytes_ = 12 * len(aj) + 4 * (N + 1) + 16 * N
ref = None
for v in ([int(a) for a in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1, 22, 23, 24, 25, 0]):
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
    kn = C.create_string_buffer(256)
    _lib.chk(hx.hipxMatGetSpMVKernel(A, kn, 256))
    print(kn.value.decode().split(" ")[0], end="  ")
    print("variant %2d  cfg %d probe %d : %.4f ms  %.1f GB/s  (%.1f%% of 8 TB/s)  identical=%s" % (v, ((v % 100) - 1) // 2, v // 1000, t, bytes_ / t / 1e6, bytes_ / t / 1e6 / 80, np.array_equal(y, ref)))

# read-stream ceiling on this box: dot of two 1 GiB vectors (pure coalesced reads), AXPY (2 reads + 1 write)
nbig = 1 << 27
P, Q = _lib.DVec(nbig), _lib.DVec(nbig)
_lib.chk(hx.hipxVecSet(P.ptr, nbig, 1.0))
_lib.chk(hx.hipxVecSet(Q.ptr, nbig, 2.0))
e0, e1 = C.c_void_p(), C.c_void_p()
_lib.chk(hx.hipxEventCreate(C.byref(e0)))
_lib.chk(hx.hipxEventCreate(C.byref(e1)))
r = C.c_double()
for name, fn, byts in (("dot 2x1GiB", lambda: hx.hipxVecDot(P.ptr, Q.ptr, nbig, C.byref(r)), 16 * nbig),
                       ("axpy 1GiB", lambda: hx.hipxVecAXPY(Q.ptr, 0.5, P.ptr, nbig), 24 * nbig),
                       ("dot 2x128MiB", lambda: hx.hipxVecDot(P.ptr, Q.ptr, N, C.byref(r)), 16 * N),
                       ("norm 128MiB", lambda: hx.hipxVecDot(P.ptr, P.ptr, N, C.byref(r)), 8 * N)):
    for _ in range(3):
        _lib.chk(fn())
    _lib.chk(hx.hipxEventRecord(e0))
    for _ in range(20):
        _lib.chk(fn())
    _lib.chk(hx.hipxEventRecord(e1))
    ms = C.c_float()
    _lib.chk(hx.hipxEventElapsedMs(e0, e1, C.byref(ms)))
    print("%-14s %.4f ms  %.1f GB/s" % (name, ms.value / 20, byts / (ms.value / 20) / 1e6))
