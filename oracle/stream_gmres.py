"""TEST INFRASTRUCTURE ONLY.  KSPGMRES(30) + PCSOR on a row partition over `nranks` ranks (BASELINE config 3: the 27-point operator
of bench_kspsolve.c on 512^3 over 8 ranks) for systems TOO LARGE to hold as one CSR in this container (3.6e9 nonzeros = 43 GB): the
Krylov loop is the C oracle's own orc_KSPSolve_GMRES (gmres.c:88-238, borthog2.c) with exact reductions; only the two operator
applications are handed in as callbacks:

  * y = A x       (MatMult_MPIAIJ mpiaij.c:1047-1061 = the row sums of the whole rows, diagonal-block entries first ... NO: the reference adds
                   the off-diagonal block's sum to the diagonal block's, mpiaij.c:1056-1059 -- see `mult` below: done exactly so, per rank)
  * z = M^{-1} r  (MatSOR_MPIAIJ mpiaij.c:1408-1412: with a zero initial guess and its = 1 every rank runs ONE local symmetric sweep of
                   MatSOR_SeqAIJ, aij.c:1842-2007, on its diagonal block)

Row slabs are assembled on the fly with the oracle's own assembly routine (orc_poisson3d_27pt = bench_kspsolve.c:115-303), split into
diagonal / off-diagonal block with the oracle's orc_MatSetUpMultiply_MPIAIJ (mmaij.c:27-65) and multiplied / relaxed with the oracle's
orc_MatMult_SeqAIJ / orc_MatMultAdd_SeqAIJ / orc_MatSOR_SeqAIJ.  Nothing but `resident` diagonal blocks and one slab per thread is held.

Pinned by tests/test_oracle_exact.py: at 128^3 on 8 ranks the history is BIT-IDENTICAL to the C oracle's exact mode on the stored matrix
(tests/golden/exact_histories.json[gmres_sor_27pt_128_np8]), and on one rank to the reference's own executable + exact-BLAS shim.
Used only by tests/golden/make_exact_golden.py.
"""
import ctypes as C
import hashlib
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import oracle as orc

MULT_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p)


def _split(rs, re, ai, aj, aa):
    """Diagonal block (local columns) and off-diagonal block (compacted columns + garray) of the rows [rs, re): mmaij.c:27-65."""
    ml, nz = re - rs, int(ai[-1])
    Ai, Aj, Aa = np.zeros(ml + 1, np.int32), np.zeros(nz + 1, np.int32), np.zeros(nz + 1)
    Bi, Bj, Ba, ga = np.zeros(ml + 1, np.int32), np.zeros(nz + 1, np.int32), np.zeros(nz + 1), np.zeros(nz + 1, np.int32)
    L = orc.lib()
    L.orc_MatSetUpMultiply_MPIAIJ.restype = C.c_int
    ng = L.orc_MatSetUpMultiply_MPIAIJ(ml, rs, re, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(Ai), orc.P(Aj), orc.P(Aa), orc.P(Bi), orc.P(Bj), orc.P(Ba), orc.P(ga))
    na, nb = int(Ai[-1]), int(Bi[-1])
    return (Ai, Aj[:na].copy(), Aa[:na].copy()), (Bi, Bj[:nb].copy(), Ba[:nb].copy(), ga[:ng].copy())


class StreamPartitionedOperator:
    """The operator of `kind` on n^3 points split over `nranks` ranks like PetscSplitOwnership; products and local SOR sweeps per rank."""

    def __init__(self, kind, n, nranks, threads=None, sub_rows=1 << 21, log=None):
        self.kind, self.n, self.N, self.nranks = kind, n, n ** 3, nranks
        self.ranges = np.zeros(nranks + 1, np.int32)
        orc.lib().orc_PetscSplitOwnership(self.N, nranks, orc.P(self.ranges))
        self.threads = threads or min(8, os.cpu_count() or 1)
        self.sub_rows = sub_rows
        self.log = log or (lambda *a: None)
        # the diagonal blocks: assembled once per rank; ranks whose block is the SAME matrix (constant-coefficient stencil, equal slabs: every
        # rank's block is the operator on an n x n x n/nranks box) share one resident copy -- checked by hashing every rank's block, not assumed
        self.blocks, self.block_of = {}, []
        for r in range(nranks):
            rs, re = int(self.ranges[r]), int(self.ranges[r + 1])
            (Ai, Aj, Aa), _ = _split(rs, re, *orc.stencil(kind, n, rs, re))
            h = hashlib.sha256()
            for a in (Ai, Aj, Aa):
                h.update(memoryview(a).cast("B"))
            key = h.hexdigest()
            if key not in self.blocks:
                self.blocks[key] = (Ai, Aj, Aa)
            self.block_of.append(key)
            self.log("rank %d rows [%d, %d): diagonal block %s (%d distinct resident)" % (r, rs, re, key[:12], len(self.blocks)))

    # ---- y = A x: per rank, y_local = A_d x_local, then y_local += A_o lvec (mpiaij.c:1056-1059: mult on the diagonal block, multadd on the
    # off-diagonal one).  The row is therefore summed as (diagonal-block entries left to right) + (off-diagonal entries left to right ADDED
    # ONE BY ONE onto that sum, aij.c:1606-1658) -- not as one left-to-right pass over the sorted row.
    def _mult_slab(self, args):
        r, s0, s1, x, y = args
        rs, re = int(self.ranges[r]), int(self.ranges[r + 1])
        ai, aj, aa = orc.stencil(self.kind, self.n, s0, s1)
        # split this sub-slab against the RANK's column range: (columns in [rs, re) -> diagonal part, the rest -> off-diagonal part)
        ml, nz = s1 - s0, int(ai[-1])
        Ai, Aj, Aa = np.zeros(ml + 1, np.int32), np.zeros(nz + 1, np.int32), np.zeros(nz + 1)
        Bi, Bj, Ba, ga = np.zeros(ml + 1, np.int32), np.zeros(nz + 1, np.int32), np.zeros(nz + 1), np.zeros(nz + 1, np.int32)
        L = orc.lib()
        ng = L.orc_MatSetUpMultiply_MPIAIJ(ml, rs, re, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(Ai), orc.P(Aj), orc.P(Aa), orc.P(Bi), orc.P(Bj), orc.P(Ba), orc.P(ga))
        xl = x[rs:re]
        yl = y[s0:s1]
        L.orc_MatMult_SeqAIJ(ml, orc.P(Ai), orc.P(Aj), orc.P(Aa), C.c_void_p(xl.ctypes.data), C.c_void_p(yl.ctypes.data))
        if ng > 0:
            lvec = np.ascontiguousarray(x[ga[:ng]])  # VecScatter: lvec[k] = x[garray[k]] (mmaij.c:108-117)
            L.orc_MatMultAdd_SeqAIJ(ml, orc.P(Bi), orc.P(Bj), orc.P(Ba), orc.P(lvec), C.c_void_p(yl.ctypes.data), C.c_void_p(yl.ctypes.data))

    def mult(self, x, y):
        jobs = []
        for r in range(self.nranks):
            rs, re = int(self.ranges[r]), int(self.ranges[r + 1])
            for s0 in range(rs, re, self.sub_rows):
                jobs.append((r, s0, min(s0 + self.sub_rows, re), x, y))
        with ThreadPoolExecutor(self.threads) as ex:
            list(ex.map(self._mult_slab, jobs))
        return y

    # ---- z = local symmetric sweep per rank
    def _sor_rank(self, args):
        r, rvec, z = args
        rs, re = int(self.ranges[r]), int(self.ranges[r + 1])
        Ai, Aj, Aa = self.blocks[self.block_of[r]]
        L = orc.lib()
        flag = 12 | 16  # SOR_LOCAL_SYMMETRIC_SWEEP | SOR_ZERO_INITIAL_GUESS (sor.c:442-446, petscmat.h:1664-1671)
        rc = L.orc_MatSOR_SeqAIJ_dispatch(re - rs, orc.P(Ai), orc.P(Aj), orc.P(Aa), C.c_void_p(rvec.ctypes.data + 8 * rs), C.c_double(1.0), flag, C.c_double(0.0), 1, 1,
                                          C.c_void_p(z.ctypes.data + 8 * rs), 0)
        assert rc == 0, rc

    def sor(self, rvec, z):
        with ThreadPoolExecutor(min(self.threads, self.nranks)) as ex:
            list(ex.map(self._sor_rank, [(r, rvec, z) for r in range(self.nranks)]))
        return z


def gmres_sor_exact(op, its, restart=30, log=None):
    """b = A*1, x0 = 0, rtol = 0: `its` iterations of KSPGMRES(restart) + PCSOR with exact reductions.  Returns the history."""
    N = op.N
    L = orc.lib()
    log = log or (lambda *a: None)

    def view(p):
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(N,))
    count = [0, 0]

    def mult_cb(_u, xp, yp):
        op.mult(view(xp), view(yp))
        count[0] += 1
        log("product %d" % count[0])

    def pc_cb(_u, rp, zp):
        op.sor(view(rp), view(zp))
        count[1] += 1
    mcb, pcb = MULT_CB(mult_cb), MULT_CB(pc_cb)
    b = np.empty(N)
    op.mult(np.ones(N), b)  # b = A * 1 with the partitioned product, as the reference's driver forms it
    k = orc.OrcKSP()
    L.orc_KSPSetDefaults(C.byref(k))
    k.m = N
    k.pc_type = 2
    k.sor_flag = 12
    k.rtol, k.abstol, k.max_it, k.normtype = 1e-50, 1e-300, its, 1
    k.gmres_restart = restart
    k.nranks = op.nranks
    k.ranges = op.ranges.ctypes.data
    k.mult_cb, k.pc_cb = C.cast(mcb, C.c_void_p).value, C.cast(pcb, C.c_void_p).value
    hist = np.zeros(its + 8 + its // restart)
    k.history, k.hist_len = hist.ctypes.data, len(hist)
    x = np.zeros(N)
    L.orc_set_exact_reductions(1)
    try:
        L.orc_KSPSolve_GMRES(C.byref(k), orc.P(b), orc.P(x))
    finally:
        L.orc_set_exact_reductions(0)
    return hist[:min(k.hist_n, len(hist))].copy()
