"""TEST INFRASTRUCTURE ONLY.  KSPSolve_CG (cg.c:119-352, preconditioned norm, PCJACOBI or PCNONE) with twice-working-precision
reductions for systems TOO LARGE for the reference build or the C oracle (32-bit nonzero counts: 27-pt 512^3 has 3.6e9; the
config-5 boxes have up to 1e9 rows): the operator is never stored -- every product assembles row slabs on the fly with the
oracle's own assembly routine and multiplies them with the oracle's MatMult_SeqAIJ restatement (aij.c:1486-1494: left-to-right
row sums), the dots are oracle/exactblas.c's Dot2, the vector updates are the reference's loops statement by statement
(two roundings per element: product, then sum).

Pinned by tests/test_oracle_exact.py: at sizes the C oracle holds, the history is BIT-IDENTICAL to the oracle's exact mode and to
the reference's own executable run with the exact-BLAS shim.  Used only by tests/golden/make_exact_golden.py.
"""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
_shim = None


def shim():
    global _shim
    if _shim is None:
        _shim = C.CDLL(os.path.join(HERE, "libexactblas.so"))
        _shim.exactblas_dot2.restype = C.c_double
        _shim.exactblas_dot2.argtypes = [C.c_long, C.c_void_p, C.c_void_p]
    return _shim


def dot2(x, y):
    return shim().exactblas_dot2(len(x), x.ctypes.data, y.ctypes.data)


class StreamOperator:
    """y = A x for a stencil operator given by its row-slab assembly routine; nothing but one slab per thread is ever held."""

    def __init__(self, kind, n, N, slab_rows=1 << 21, threads=None):
        self.kind, self.n, self.N = kind, n, int(N)
        self.slabs = [(rs, min(rs + slab_rows, self.N)) for rs in range(0, self.N, slab_rows)]
        self.threads = threads or min(8, os.cpu_count() or 1)
        self.diag = None

    def _slab(self, args):
        rs, re, x, y, want_diag = args
        ai, aj, aa = orc.stencil(self.kind, self.n, rs, re)
        orc.lib().orc_MatMult_SeqAIJ(re - rs, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(x), C.c_void_p(y.ctypes.data + 8 * rs))
        if want_diag is not None:  # MatGetDiagonal_SeqAIJ aij.c:1347-1380: the entry with column == row (0 if absent)
            rows = np.repeat(np.arange(rs, re, dtype=np.int64), np.diff(ai))
            hit = aj == rows
            d = np.zeros(re - rs)
            d[rows[hit] - rs] = aa[hit]
            want_diag[rs:re] = d

    def mult(self, x, y, diag=None):
        with ThreadPoolExecutor(self.threads) as ex:
            list(ex.map(self._slab, [(rs, re, x, y, diag) for rs, re in self.slabs]))
        return y


def cg_exact(op, pc, its):
    """b = A*1, x0 = 0, rtol = 0: `its` iterations of cg.c's loop.  Returns the preconditioned residual norms (its + 1 entries)
    and the 2-norm of (x - 1)."""
    N = op.N
    ones = np.ones(N)
    b = np.empty(N)
    diag = np.empty(N) if pc == "jacobi" else None
    op.mult(ones, b, diag)  # b = A * 1 (ex2.c:139 style), the diagonal read off in the same pass
    del ones
    if pc == "jacobi":  # PCSetUp_Jacobi jacobi.c:205-266: 1/diag, zeros -> 1
        dinv = 1.0 / diag
        dinv[diag == 0.0] = 1.0
        del diag
    x = np.zeros(N)
    r = b.copy()  # cg.c:162
    z = r * dinv if pc == "jacobi" else r.copy()  # cg.c:170 PCApply
    dp = np.sqrt(dot2(z, z))  # cg.c:171 VecNorm = sqrt(ddot(z, z)), bvec2.c:204
    hist = [dp]
    beta = dot2(z, r)  # cg.c:216 VecXDot(Z, R)
    betaold, p = 1.0, None
    for i in range(its):
        if i == 0:
            p = z.copy()  # cg.c:236
        else:
            bb = beta / betaold
            np.multiply(p, bb, out=p)  # VecAYPX_Seq dvec2.c:779: yy[i] = xx[i] + beta * yy[i] (product rounded, then the sum)
            np.add(z, p, out=p)
        w = z  # cg.c:145: W aliases Z
        op.mult(p, w)  # cg.c:257
        dpi = dot2(p, w)  # cg.c:258
        betaold = beta
        a = beta / dpi  # cg.c:288
        x += a * p  # cg.c:305 VecAXPY: y[i] += alpha * x[i] (published daxpy: two roundings)
        r += (-a) * w  # cg.c:306
        z = w
        if pc == "jacobi":
            np.multiply(r, dinv, out=z)  # cg.c:308 PCApply_Jacobi = VecPointwiseMult(z, r, diag) jacobi.c:360
        else:
            np.copyto(z, r)
        dp = np.sqrt(dot2(z, z))  # cg.c:309
        hist.append(dp)
        beta = dot2(z, r)  # cg.c:344
    return np.array(hist), float(np.sqrt(dot2(x - 1.0, x - 1.0)))
