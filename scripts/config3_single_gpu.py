"""GPU: the matrix of BASELINE config 3 (27-pt Poisson 512^3: 134,217,728 rows, 3,581,577,000 nonzeros > 2^31, 64-bit row
offsets) on ONE MI355X -- full-size checks of the int64 path and the 1-GPU reference point for the 1 -> 8 GPU scaling target.
  python scripts/config3_single_gpu.py [n]        (default n = 512; needs ~50 GB of host RAM and ~65 GB of HBM)
Checks: sampled rows of y = A x bit-identical to the row sums recomputed on the host from the CSR arrays; CG + PCJACOBI steps."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from petsc_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = n ** 3
nnz = (3 * n - 2) ** 3
need_gb = (12 * nnz + 8 * N) / 1e9 * 1.15
avail_gb = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 1e6
print("n=%d N=%d nnz=%d host need %.0f GB, available %.0f GB" % (n, N, nnz, need_gb, avail_gb), flush=True)
if avail_gb < need_gb + 16:
    print("SKIP: not enough host memory")
    sys.exit(0)
hx = _lib.init(0)
_, ks = _lib.load()
t0 = time.time()
ai = np.zeros(N + 1, np.int64)
aj = np.zeros(nnz, np.int32)
aa = np.zeros(nnz, np.float64)
got = ks.HipxAssemble_bench27_64(n, 0, N, ai.ctypes.data_as(C.c_void_p), aj.ctypes.data_as(C.c_void_p), aa.ctypes.data_as(C.c_void_p))
assert got == nnz == ai[-1]
print("assembled in %.1f s" % (time.time() - t0), flush=True)
t0 = time.time()
A = _lib.mat_create_csr(N, N, ai, aj, aa)
print("uploaded in %.1f s" % (time.time() - t0), flush=True)
xh = 1.0 + (np.arange(N) % 17) / 17.0
X, Y = _lib.DVec(N, xh), _lib.DVec(N)
t0 = time.time()
_lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
_lib.chk(hx.hipxDeviceSynchronize())
kn = C.create_string_buffer(256)
_lib.chk(hx.hipxMatGetSpMVKernel(A, kn, 256))
print("first MatMult incl. packed-format set-up %.3f s; kernel: %s" % (time.time() - t0, kn.value.decode()), flush=True)
_lib.chk(hx.hipxProfileSpMV(1))
for _ in range(10):
    _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
cnt, ms = C.c_int(), C.c_double()
_lib.chk(hx.hipxProfileSpMVGet(C.byref(cnt), C.byref(ms)))
_lib.chk(hx.hipxProfileSpMV(0))
t = ms.value / cnt.value
byts = 12 * nnz + 8 * (N + 1) + 16 * N
print("SpMV %.3f ms  algorithmic %.2f GB -> %.0f GB/s (%.1f%% of 8 TB/s)" % (t, byts / 1e9, byts / t / 1e6, byts / t / 1e6 / 80), flush=True)
y = Y.get()
rng = np.random.default_rng(1)
rows = np.unique(np.concatenate([rng.integers(0, N, 4000), np.arange(0, 2000), np.arange(N - 2000, N), np.arange(N // 2, N // 2 + 1000)]))
bad = 0
for r in rows:
    s = 0.0
    for k in range(ai[r], ai[r + 1]):
        s += aa[k] * xh[aj[k]]  # left to right, separate multiply and add: aij.c:1486-1494
    bad += (s != y[r])
print("sampled rows bit-identical: %d of %d" % (len(rows) - bad, len(rows)), flush=True)
assert bad == 0
# CG + PCJACOBI, b = A*1
M = _lib.HipxMat(m=N, A=A, B=None, halo=None, lvec=None, nranks=1)
ones = _lib.DVec(N, np.ones(N))
B, XS = _lib.DVec(N), _lib.DVec(N)
_lib.chk(ks.HipxMatMult(C.byref(M), ones.ptr, B.ptr))
pc = _lib.HipxPC()
ks.HipxPCSetDefaults(C.byref(pc))
_lib.chk(ks.HipxPCSetUp(C.byref(pc), C.byref(M)))
k = _lib.HipxKSP()
ks.HipxKSPSetDefaults(C.byref(k))
k.rtol, k.abstol, k.divtol, k.max_it, k.fused = 1e-50, 1e-300, 1e300, 1000, 1
_lib.chk(ks.HipxKSPCGBegin(C.byref(k), C.byref(M), C.byref(pc), B.ptr, XS.ptr))
_lib.chk(ks.HipxKSPCGStep(C.byref(k), C.byref(M), C.byref(pc), B.ptr, XS.ptr, 5))
_lib.chk(hx.hipxDeviceSynchronize())
t0 = time.perf_counter()
_lib.chk(ks.HipxKSPCGStep(C.byref(k), C.byref(M), C.byref(pc), B.ptr, XS.ptr, 40))
_lib.chk(hx.hipxDeviceSynchronize())
el = time.perf_counter() - t0
print("CG+Jacobi: 40 iterations in %.3f s = %.1f it/s (%.2f ms/it), rnorm %.6e reason %d" % (el, 40 / el, 1e3 * el / 40, k.rnorm, k.reason), flush=True)
