"""HipxKSPSolve_PIPECG on 7-pt 256^3 (+ PCJACOBI, constant diagonal): ms per iteration of 200-iteration solves for the developer switches of
pipecg_update_kernel (HIPX_PIPECG_NT = non-temporal loads / stores, HIPX_PIPECG_U = pairs in flight per thread).  python scripts/pipecg_kernel_timing.py [n]"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(n):
    from petsc_amd import _lib
    hx = _lib.init(0)
    _, ks = _lib.load()
    N = n ** 3
    ai = np.zeros(N + 1, np.int32)
    nnz = ks.HipxAssemble_poisson7(n, 0, N, None, None, None)
    aj, aa = np.zeros(nnz, np.int32), np.zeros(nnz)
    ks.HipxAssemble_poisson7(n, 0, N, ai.ctypes.data_as(C.c_void_p), aj.ctypes.data_as(C.c_void_p), aa.ctypes.data_as(C.c_void_p))
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    one_, B, X = _lib.DVec(N, np.ones(N)), _lib.DVec(N), _lib.DVec(N)
    _lib.chk(hx.hipxMatMult(A, one_.ptr, B.ptr))
    M = _lib.HipxMat(m=N, A=A, B=None, halo=None, lvec=None, nranks=1)
    p = _lib.HipxPC()
    ks.HipxPCSetDefaults(C.byref(p))
    _lib.chk(ks.HipxPCSetUp(C.byref(p), C.byref(M)))
    k = _lib.HipxKSP()
    ks.HipxKSPSetDefaults(C.byref(k))
    k.rtol, k.abstol, k.divtol, k.max_it = 1e-50, 1e-300, 1e300, 199
    best = 1e9
    for rep in range(4):
        _lib.chk(hx.hipxStreamSynchronize())
        t0 = time.perf_counter()
        _lib.chk(ks.HipxKSPSolve_PIPECG(C.byref(k), C.byref(M), C.byref(p), B.ptr, X.ptr))
        _lib.chk(hx.hipxStreamSynchronize())
        dt = time.perf_counter() - t0
        if rep:
            best = min(best, dt)
    print("NT=%s U=%s  %.4f ms / iteration  (%d passes, rnorm %.17g)" % (os.environ.get("HIPX_PIPECG_NT", "0"), os.environ.get("HIPX_PIPECG_U", "1"), 1e3 * best / k.its, k.its, k.rnorm), flush=True)


if __name__ == "__main__":
    if os.environ.get("HIPX_PIPECG_CHILD"):
        one(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
    else:
        for nt in ("0", "1", "2", "3"):
            for u in ("1", "2"):
                env = dict(os.environ, HIPX_PIPECG_CHILD="1", HIPX_PIPECG_NT=nt, HIPX_PIPECG_U=u)
                subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env)
