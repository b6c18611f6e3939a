"""GPU timing of the config-3 style path on one GPU: 27-pt / 7-pt Poisson, KSPGMRES(30) + PCSOR, and the SOR apply alone."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from petsc_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
st = int(sys.argv[2]) if len(sys.argv) > 2 else 27
hx = _lib.init(0)
_, ks = _lib.load()
N = n ** 3
ai, aj, aa = bench.assemble(ks, st, (n, n, n), 0, N)
A = _lib.mat_create_csr(N, N, ai, aj, aa)
M = _lib.HipxMat(m=N, A=A, B=None, halo=None, lvec=None, nranks=1)
ones = _lib.DVec(N, np.ones(N))
B, X = _lib.DVec(N), _lib.DVec(N)
_lib.chk(ks.HipxMatMult(C.byref(M), ones.ptr, B.ptr))
t0 = time.perf_counter()
_lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 12 | 16, 0.0, 1, 1, X.ptr))
_lib.chk(hx.hipxDeviceSynchronize())
print("SOR first apply (level schedule build + sweep): %.3f s" % (time.perf_counter() - t0))
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(10):
        _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 12 | 16, 0.0, 1, 1, X.ptr))
    _lib.chk(hx.hipxDeviceSynchronize())
    dt = (time.perf_counter() - t0) / 10
    nnz = len(aj)
    print("SOR symmetric sweep: %.3f ms  (algorithmic %.1f GB/s)" % (dt * 1e3, (2 * 12 * nnz + 40 * N) / dt / 1e9))
if os.environ.get("SOR_ONLY"):
    sys.exit(0)
for pcname, pct in (("sor", 2), ("jacobi", 1)):
    pc = _lib.HipxPC()
    ks.HipxPCSetDefaults(C.byref(pc))
    pc.type = pct
    _lib.chk(ks.HipxPCSetUp(C.byref(pc), C.byref(M)))
    ksp = _lib.HipxKSP()
    ks.HipxKSPSetDefaults(C.byref(ksp))
    ksp.rtol, ksp.max_it = 1e-50, 60
    t0 = time.perf_counter()
    _lib.chk(ks.HipxKSPSolve_GMRES(C.byref(ksp), C.byref(M), C.byref(pc), B.ptr, X.ptr))
    _lib.chk(hx.hipxDeviceSynchronize())
    dt = time.perf_counter() - t0
    print("GMRES(30)+%s: %d iterations in %.3f s = %.1f it/s (rnorm %.3e)" % (pcname, ksp.its, dt, ksp.its / dt, ksp.rnorm))
    _lib.chk(ks.HipxKSPDestroyWork(C.byref(ksp)))  # the GMRES slab belongs to the HipxKSP until this call (include/hipx_ksp.h)
    _lib.chk(ks.HipxPCDestroy(C.byref(pc)))
