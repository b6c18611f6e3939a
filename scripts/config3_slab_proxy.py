"""Per-rank proxy of BASELINE config 3 on ONE GPU: the rows rank r of 8 would own of the 27-pt 512^3 operator (512x512x64 =
16.8 M rows, ~451 M nonzeros), split exactly as MatSetUpMultiply_MPIAIJ does; times the diagonal-block SpMV, the off-diagonal
MatMultAdd, a symmetric SOR sweep and MDot/MAXPY(30) -- the per-iteration pieces of KSPGMRES(30)+PCSOR.
  python scripts/config3_slab_proxy.py [n=512] [nranks=8] [rank=3]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from petsc_amd import _lib  # noqa: E402
from petsc_amd import dist as pdist  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 3
hx = _lib.init(0)
_, ks = _lib.load()
N = n ** 3
ranges = pdist.split_ownership(N, nranks)
rs, re = int(ranges[rank]), int(ranges[rank + 1])
t0 = time.perf_counter()
ai, aj, aa = bench.assemble(ks, 27, (n, n, n), rs, re)
plan = pdist.build_plan(ai, aj, aa, ranges, rank, dist=None) if nranks == 1 else None
if plan is None:  # build_plan needs the exchange only for the send lists; the split itself is local
    import types
    fake = types.SimpleNamespace(all_gather_object=lambda out, obj, group=None: out.__setitem__(slice(None), [obj] * len(out)))
    plan = pdist.build_plan(ai, aj, aa, ranges, rank, dist=fake)
m = plan["m"]
print("slab rows %d, diag nnz %d, offdiag nnz %d, ghosts %d, assembly+split %.1f s" % (m, plan["Ai"][-1], plan["Bi"][-1] if plan["nrows_c"] else 0, plan["nghost"], time.perf_counter() - t0))
A = _lib.mat_create_csr(m, m, plan["Ai"], plan["Aj"], plan["Aa"])
B = _lib.mat_create_cprow(m, max(plan["nghost"], 1), plan["nrows_c"], plan["Bi"], plan["ridx"], plan["Bj"], plan["Ba"])
X, Y, LV = _lib.DVec(m, 1.0 + (np.arange(m) % 17) / 17.0), _lib.DVec(m), _lib.DVec(max(plan["nghost"], 1), np.ones(max(plan["nghost"], 1)))
e0, e1 = C.c_void_p(), C.c_void_p()
_lib.chk(hx.hipxEventCreate(C.byref(e0)))
_lib.chk(hx.hipxEventCreate(C.byref(e1)))


def timed(name, fn, reps, byts):
    for _ in range(2):
        _lib.chk(fn())
    _lib.chk(hx.hipxEventRecord(e0))
    for _ in range(reps):
        _lib.chk(fn())
    _lib.chk(hx.hipxEventRecord(e1))
    ms = C.c_float()
    _lib.chk(hx.hipxEventElapsedMs(e0, e1, C.byref(ms)))
    t = ms.value / reps
    print("%-28s %8.3f ms   %7.1f GB/s algorithmic" % (name, t, byts / t / 1e6))


nza, nzb = int(plan["Ai"][-1]), int(plan["Bi"][-1]) if plan["nrows_c"] else 0
timed("diag SpMV (MatMult A)", lambda: hx.hipxMatMult(A, X.ptr, Y.ptr), 20, 12 * nza + 4 * (m + 1) + 16 * m)
timed("offdiag MatMultAdd (B)", lambda: hx.hipxMatMultAdd(B, LV.ptr, Y.ptr, Y.ptr), 20, 12 * nzb + 24 * plan["nrows_c"])
ysor = {}
for mode in ("strand", "dep"):
    os.environ["HIPX_SOR_MODE"] = mode
    timed("SOR local symmetric sweep [%s]" % mode, lambda: hx.hipxMatSOR(A, X.ptr, 1.0, 12 | 16, 0.0, 1, 1, Y.ptr), 5 if mode == "strand" else 2, 2 * 12 * nza + 40 * m)
    ysor[mode] = Y.get()
del os.environ["HIPX_SOR_MODE"]
print("SOR strand == dep bit for bit:", np.array_equal(ysor["strand"], ysor["dep"]))
vs = [_lib.DVec(m, np.full(m, 1.0 / (k + 1))) for k in range(30)]
ptrs = (C.c_void_p * 30)(*[v.ptr.value for v in vs])
res = (C.c_double * 30)()
al = (C.c_double * 30)(*[0.01] * 30)
timed("VecMDot (30 vectors)", lambda: hx.hipxVecMDot(X.ptr, 30, ptrs, m, res), 5, 8 * m * 31)
timed("VecMAXPY (30 vectors)", lambda: hx.hipxVecMAXPY(Y.ptr, 30, al, ptrs, m), 5, 8 * m * 32)
