"""GPU: surrogate for BASELINE config 4 (SuiteSparse Janna/Flan_1565: 1,564,794 rows, ~117 M nonzeros, ~75 per row, irregular;
not in the container and no network -- SURVEY.md 8(d) asks for a documented substitute).  Synthetic matrix with the same
shape statistics: a hexahedral elasticity mesh like Flan_1565's -- 80^3 nodes x 3 degrees of freedom = 1,536,000 rows, every
node coupled to its 27 neighbours by a dense 3x3 block (up to 81 entries per row, ~79 on average, 121 M nonzeros), node
numbering shuffled inside windows of 512, and ALL values distinct (standard normal), so no value dictionary applies.
Checks y = A x bit-identical on sampled rows, then times the SpMV kernel forms and CG + PCJACOBI.
  python scripts/config4_surrogate.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from petsc_amd import _lib  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
from surrogates import flan_surrogate  # noqa: E402

t0 = time.time()
rng = np.random.default_rng(7)
ai, cols, aa = flan_surrogate()
N, nnz = len(ai) - 1, int(ai[-1])
lens = np.diff(ai)
print("surrogate: N=%d nnz=%d (%.1f per row, max %d) built in %.1f s" % (N, nnz, nnz / N, lens.max(), time.time() - t0), flush=True)
hx = _lib.init(0)
_, ks = _lib.load()
A = _lib.mat_create_csr(N, N, ai, cols, aa)
xh = 1.0 + (np.arange(N) % 17) / 17.0
X, Y = _lib.DVec(N, xh), _lib.DVec(N)
byts = 12 * nnz + 4 * (N + 1) + 16 * N
ref = None
for v in ([int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else (0, 22, 23, 1)):
    _lib.chk(hx.hipxMatSetSpMVVariant(A, v))
    for _ in range(3):
        _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    _lib.chk(hx.hipxProfileSpMV(1))
    for _ in range(30):
        _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    cnt, ms = C.c_int(), C.c_double()
    _lib.chk(hx.hipxProfileSpMVGet(C.byref(cnt), C.byref(ms)))
    _lib.chk(hx.hipxProfileSpMV(0))
    kn = C.create_string_buffer(256)
    _lib.chk(hx.hipxMatGetSpMVKernel(A, kn, 256))
    y = Y.get()
    if ref is None:
        ref = y
    t = ms.value / cnt.value
    print("variant %2d %-22s %.4f ms  %.0f GB/s (%.1f%% of 8 TB/s) identical=%s" % (v, kn.value.decode().split(" ")[0], t, byts / t / 1e6, byts / t / 1e6 / 80, np.array_equal(y, ref)), flush=True)
bad = 0
samp = np.unique(np.concatenate([rng.integers(0, N, 3000), np.arange(0, 500), np.arange(N - 500, N)]))
for r in samp:
    s = 0.0
    for k in range(ai[r], ai[r + 1]):
        s += aa[k] * xh[cols[k]]
    bad += (s != ref[r])
print("sampled rows bit-identical: %d of %d" % (len(samp) - bad, len(samp)), flush=True)
assert bad == 0
