"""Symmetric SOR sweep on matrices with ARBITRARY values on a stencil pattern (variable-coefficient operators): the strand
schedule from the pattern templates with streamed coefficients against the level-ordered dependency-driven sweep, on one GPU.
  python scripts/sor_var_timing.py [stencil=7|27] [n=256] [nz=n]      (HIPX_SOR_VAR_RING=4|8|16: depth of the coefficient ring)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from petsc_amd import _lib  # noqa: E402

stencil = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nz = int(sys.argv[3]) if len(sys.argv) > 3 else n
hx = _lib.init(0)
_, ks = _lib.load()
m = n * n * nz
if stencil == 27 and nz != n:  # a slab of the cube's operator = the diagonal block one of n / nz ranks owns (config 3's per-rank matrix)
    import types
    from petsc_amd import dist as pdist
    nranks = n // nz
    ranges = pdist.split_ownership(n ** 3, nranks)
    rank = nranks // 2 - 1
    ai, aj, aa = bench.assemble(ks, 27, (n, n, n), int(ranges[rank]), int(ranges[rank + 1]))
    fake = types.SimpleNamespace(all_gather_object=lambda out, obj, group=None: out.__setitem__(slice(None), [obj] * len(out)))
    plan = pdist.build_plan(ai, aj, aa, ranges, rank, dist=fake)
    ai, aj, aa = plan["Ai"], plan["Aj"], plan["Aa"].copy()
    assert plan["m"] == m
else:
    ai, aj, aa = bench.assemble(ks, stencil, (n, n, nz), 0, m)
rng = np.random.default_rng(1)
aa = np.ascontiguousarray(aa * (1.0 + 0.3 * rng.random(aa.size)))
A = _lib.mat_create_csr(m, m, ai, aj, aa)
X, Y = _lib.DVec(m, 1.0 + (np.arange(m) % 17) / 17.0), _lib.DVec(m)
e0, e1 = C.c_void_p(), C.c_void_p()
_lib.chk(hx.hipxEventCreate(C.byref(e0)))
_lib.chk(hx.hipxEventCreate(C.byref(e1)))
nza = int(ai[-1])
out = {}
for mode in ("strand", "dep"):
    os.environ["HIPX_SOR_MODE"] = mode
    fn = lambda: hx.hipxMatSOR(A, X.ptr, 1.0, 12 | 16, 0.0, 1, 1, Y.ptr)  # noqa: E731
    for _ in range(2):
        _lib.chk(fn())
    reps = 5 if mode == "strand" else 2
    _lib.chk(hx.hipxEventRecord(e0))
    for _ in range(reps):
        _lib.chk(fn())
    _lib.chk(hx.hipxEventRecord(e1))
    ms = C.c_float()
    _lib.chk(hx.hipxEventElapsedMs(e0, e1, C.byref(ms)))
    used = C.c_int()
    _lib.chk(hx.hipxMatGetSORMode(A, C.byref(used)))
    t = ms.value / reps
    print("%d-pt %dx%dx%d variable coefficients: symmetric SOR sweep [%s, mode used %d] %8.3f ms  (%.0f GB/s on 2 x (12 nnz) + 40 m bytes)"
          % (stencil, n, n, nz, mode, used.value, t, (24 * nza + 40 * m) / t / 1e6))
    out[mode] = Y.get()
print("strand == dep bit for bit:", np.array_equal(out["strand"], out["dep"]))
# the same pattern with the operator's own (few distinct) values: row templates, coefficients from the per-template tables
_lib.mat_destroy(A)
ai, aj, aa0 = (plan["Ai"], plan["Aj"], plan["Aa"]) if (stencil == 27 and nz != n) else bench.assemble(ks, stencil, (n, n, nz), 0, m)
A = _lib.mat_create_csr(m, m, ai, aj, aa0)
os.environ["HIPX_SOR_MODE"] = "strand"
for _ in range(2):
    _lib.chk(hx.hipxMatSOR(A, X.ptr, 1.0, 12 | 16, 0.0, 1, 1, Y.ptr))
_lib.chk(hx.hipxEventRecord(e0))
for _ in range(5):
    _lib.chk(hx.hipxMatSOR(A, X.ptr, 1.0, 12 | 16, 0.0, 1, 1, Y.ptr))
_lib.chk(hx.hipxEventRecord(e1))
ms = C.c_float()
_lib.chk(hx.hipxEventElapsedMs(e0, e1, C.byref(ms)))
print("%d-pt %dx%dx%d constant coefficients (row templates): symmetric SOR sweep [strand] %8.3f ms" % (stencil, n, n, nz, ms.value / 5))
del os.environ["HIPX_SOR_MODE"]
