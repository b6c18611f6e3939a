"""ctypes bindings of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY: import from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never from petsc_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class OrcKSP(C.Structure):
    _fields_ = [("m", C.c_int), ("ai", C.c_void_p), ("aj", C.c_void_p), ("aa", C.c_void_p), ("nranks", C.c_int), ("ranges", C.c_void_p),
                ("pc_type", C.c_int), ("sor_flag", C.c_int), ("sor_omega", C.c_double), ("sor_shift", C.c_double), ("sor_its", C.c_int),
                ("sor_lits", C.c_int), ("normtype", C.c_int), ("rtol", C.c_double), ("abstol", C.c_double), ("divtol", C.c_double),
                ("max_it", C.c_int), ("min_it", C.c_int), ("gmres_restart", C.c_int), ("gmres_haptol", C.c_double), ("gmres_cgs_refine", C.c_int),
                ("guess_nonzero", C.c_int), ("its", C.c_int), ("reason", C.c_int), ("rnorm", C.c_double), ("history", C.c_void_p),
                ("hist_len", C.c_int), ("hist_n", C.c_int), ("no_inode", C.c_int),
                ("mult_cb", C.c_void_p), ("pc_cb", C.c_void_p), ("user", C.c_void_p)]  # round 5: streamed operators (oracle/stream_gmres.py)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", HERE, "-s"])
        _lib = C.CDLL(path)
        for f in ("orc_laplace2d_5pt", "orc_poisson3d_7pt", "orc_poisson3d_27pt", "orc_poisson3d_7pt_box"):
            getattr(_lib, f).restype = C.c_int64
        _lib.orc_VecDot_Seq.restype = C.c_double
        _lib.orc_VecNorm_Seq.restype = C.c_double
    return _lib


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def stencil(kind, n, rstart=None, rend=None, m=None):
    """CSR (ai, aj, aa) of rows [rstart, rend) with global columns.  kind: '5pt' (m x n grid), '7pt', '27pt'."""
    L = lib()
    if kind == "5pt":
        N = (m if m is not None else n) * n
        f = lambda *a: L.orc_laplace2d_5pt(m if m is not None else n, n, *a)  # noqa: E731
    elif kind == "7pt":
        N = n ** 3
        f = lambda *a: L.orc_poisson3d_7pt(n, *a)  # noqa: E731
    elif kind == "27pt":
        N = n ** 3
        f = lambda *a: L.orc_poisson3d_27pt(n, *a)  # noqa: E731
    elif kind == "7pt_box":  # n = (nx, ny, nz)
        nx, ny, nz_ = n
        N = nx * ny * nz_
        f = lambda *a: L.orc_poisson3d_7pt_box(nx, ny, nz_, *a)  # noqa: E731
    else:
        raise ValueError(kind)
    rs = 0 if rstart is None else rstart
    re = N if rend is None else rend
    nz = f(rs, re, None, None, None)
    ai = np.zeros(re - rs + 1, np.int32)
    aj = np.zeros(max(nz, 1), np.int32)
    aa = np.zeros(max(nz, 1), np.float64)
    f(rs, re, P(ai), P(aj), P(aa))
    return ai, aj[:nz], aa[:nz]


def matmult(ai, aj, aa, x, no_inode=False):
    """MatMult as the reference's MATSEQAIJ dispatches it (aij.c:1459): see matmult_ref (the same function; scalar stencils have no inodes)."""
    return matmult_ref(ai, aj, aa, x, no_inode=no_inode)


def matmult_ref(ai, aj, aa, x, yadd=None, no_inode=False):
    """y = A x (or yadd + A x) as the reference's MATSEQAIJ dispatches it (aij.c:1459, 1617): MatMult_SeqAIJ_Inode's pairwise row sums when
    the matrix has inodes (runs of rows with one column list, inode.c:3920; never on a scalar stencil), MatMult_SeqAIJ's left-to-right
    sums otherwise or with no_inode (-mat_no_inode)."""
    m = len(ai) - 1
    z = np.zeros(m)
    xx = np.ascontiguousarray(x, dtype=np.float64)
    yy = None if yadd is None else np.ascontiguousarray(yadd, dtype=np.float64)
    lib().orc_MatMult_SeqAIJ_dispatch(m, P(ai), P(aj), P(aa), P(xx), None if yy is None else P(yy), P(z), 1 if no_inode else 0)
    return z


def matmult_mpi(ai, aj, aa, x, nranks):
    """y = A x as MatMult_MPIAIJ forms it on `nranks` ranks (mpiaij.c:1047-1061): per rank the diagonal block's row sum, then the off-diagonal
    block's terms added one by one (differs from the one-rank row sum by rounding when the values are not exactly summable)."""
    m = len(ai) - 1
    y = np.zeros(m)
    xx = np.ascontiguousarray(x, dtype=np.float64)
    lib().orc_MatMult_MPIAIJ(m, P(ai), P(aj), P(aa), int(nranks), P(xx), P(y))
    return y


def ksp_solve(kind, ai, aj, aa, b, pc="jacobi", rtol=1e-5, max_it=10000, normtype=1, restart=30, refine=0, sor_flag=12, omega=1.0,
              nranks=1, x0=None, abstol=1e-50, sor_its=1, sor_lits=1, exact=False):
    """exact=True: every dot product / norm of the solve is evaluated as if in twice the working precision (the correctly rounded
    reduction): the yardstick both the reference's BLAS and the GPU's reduction tree are roundings of."""
    L = lib()
    L.orc_set_exact_reductions(1 if exact else 0)
    try:
        return _ksp_solve(L, kind, ai, aj, aa, b, pc, rtol, max_it, normtype, restart, refine, sor_flag, omega, nranks, x0, abstol, sor_its, sor_lits)
    finally:
        L.orc_set_exact_reductions(0)


def _ksp_solve(L, kind, ai, aj, aa, b, pc, rtol, max_it, normtype, restart, refine, sor_flag, omega, nranks, x0, abstol, sor_its, sor_lits):
    k = OrcKSP()
    L.orc_KSPSetDefaults(C.byref(k))
    m = len(ai) - 1
    k.m = m
    k.ai, k.aj, k.aa = ai.ctypes.data, aj.ctypes.data, aa.ctypes.data
    k.pc_type = {"none": 0, "jacobi": 1, "sor": 2}[pc]
    k.sor_flag = sor_flag
    k.sor_omega = omega
    k.sor_its, k.sor_lits = sor_its, sor_lits
    k.rtol, k.abstol, k.max_it, k.normtype = rtol, abstol, max_it, normtype
    k.gmres_restart, k.gmres_cgs_refine = restart, refine
    ranges = None
    if nranks > 1:
        ranges = np.zeros(nranks + 1, np.int32)
        L.orc_PetscSplitOwnership(m, nranks, P(ranges))
        k.nranks = nranks
        k.ranges = ranges.ctypes.data
    hist = np.zeros(max_it + 2 + max_it // max(restart, 1) + 2)
    k.history, k.hist_len = hist.ctypes.data, len(hist)
    x = np.zeros(m) if x0 is None else np.array(x0, dtype=np.float64)
    k.guess_nonzero = 0 if x0 is None else 1
    bb = np.ascontiguousarray(b, dtype=np.float64)
    {"cg": L.orc_KSPSolve_CG, "gmres": L.orc_KSPSolve_GMRES, "pipecg": L.orc_KSPSolve_PIPECG, "groppcg": L.orc_KSPSolve_GROPPCG}[kind](C.byref(k), P(bb), P(x))
    return x, int(k.its), int(k.reason), hist[:min(k.hist_n, len(hist))].copy()
