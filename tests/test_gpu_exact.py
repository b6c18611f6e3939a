"""GPU: the opt-in EXACT reduction mode (hipxSetReductionMode(HIPX_RED_EXACT) / HIPX_REDUCTIONS=exact / -hipx_reductions exact).

Every sum-reduction kernel then carries its sums as unevaluated (hi, lo) pairs (Dot2 / Sum2, Ogita-Rump-Oishi) through the thread
loop, the wave / workgroup folds, the fold of the workgroups' partials and the fold over ranks, and rounds once.  The yardstick is
what the REFERENCE computes when its BLAS reductions are exactly rounded: oracle/libexactblas.so (ddot / dasum / dgemv "T" under
bvec1.c:27, bvec2.c:202-223, dvec2.c:557), itself pinned against exact rational arithmetic by tests/test_oracle_exact.py.  Bar:
BIT-IDENTICAL results -- for single reductions on adversarial vectors, and for whole CG histories (the other kernels of the
iteration are bit-exact already), whatever the kernel fusion / launch-ahead form; GMRES(30)+PCSOR within 1e-12."""
import ctypes as C
import math
import os
from fractions import Fraction

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def exact(hx):
    from petsc_amd import _lib
    _lib.chk(hx.hipxSetReductionMode(1))
    yield hx
    _lib.chk(hx.hipxSetReductionMode(0))


def shim():
    L = C.CDLL(os.path.join(ROOT, "oracle", "libexactblas.so"))
    L.exactblas_dot2.restype = C.c_double
    L.exactblas_dot2.argtypes = [C.c_long, C.c_void_p, C.c_void_p]
    L.dasum_.restype = C.c_double
    return L


def dot2(L, x, y):
    x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
    return L.exactblas_dot2(len(x), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)) if len(x) else 0.0


def cancelling(n, seed):
    """x.y cancels: sum |x_i y_i| / |x.y| ~ 1e6 sqrt(n) for the sizes that are also checked against exact rational arithmetic (a plain fp64
    tree is wrong from the 8th digit on), ~1e2 sqrt(n) beyond (Dot2's own error n eps^2 cond must stay far below one ulp for two
    differently associated Dot2 evaluations to round to the same double); magnitudes spread over 6 decades."""
    rng = np.random.default_rng(seed)
    h = n // 2
    pert = 1e-6 if n <= 5000 else 1e-2
    a = rng.standard_normal(h) * 10.0 ** rng.integers(-3, 3, h)
    b = rng.standard_normal(h)
    x = np.concatenate([a, a, rng.standard_normal(n - 2 * h)])
    y = np.concatenate([b, -b * (1.0 + pert * rng.standard_normal(h)), 1e-3 * rng.standard_normal(n - 2 * h)])
    p = rng.permutation(n)
    return x[p], y[p]


SIZES = [0, 1, 2, 3, 63, 64, 1023, 2049, 100003, (1 << 20) + 1, 1 << 22]


@pytest.mark.parametrize("n", SIZES)
def test_dot_norm_sum_bit_identical_to_exact_blas(exact, n):
    from petsc_amd import _lib
    hx, L = exact, shim()
    mode = C.c_int(-1)
    _lib.chk(hx.hipxGetReductionMode(C.byref(mode)))
    assert mode.value == 1
    x, y = cancelling(n, 11 + n % 97)
    X, Y = _lib.DVec(max(n, 1), np.concatenate([x, np.zeros(1 if n == 0 else 0)])), _lib.DVec(max(n, 1), np.concatenate([y, np.zeros(1 if n == 0 else 0)]))
    r = C.c_double()
    _lib.chk(hx.hipxVecDot(X.ptr, Y.ptr, n, C.byref(r)))
    want = dot2(L, x, y)
    assert r.value == want, (n, r.value, want)
    if 0 < n <= 2049:  # and against exact rational arithmetic
        ex = float(sum(Fraction(float(a)) * Fraction(float(b)) for a, b in zip(x, y)))
        assert r.value == ex
    # NORM_2 = sqrt(ddot(x, x)) (bvec2.c:204), NORM_1 = dasum (bvec2.c:223), NORM_1_AND_2
    res = (C.c_double * 2)()
    _lib.chk(hx.hipxVecNorm(X.ptr, n, 1, res))
    assert res[0] == math.sqrt(dot2(L, x, x))
    nn, one = C.c_int(n), C.c_int(1)
    xs = np.ascontiguousarray(x)
    want1 = L.dasum_(C.byref(nn), xs.ctypes.data_as(C.c_void_p), C.byref(one)) if n else 0.0
    _lib.chk(hx.hipxVecNorm(X.ptr, n, 0, res))
    assert res[0] == want1
    _lib.chk(hx.hipxVecNorm(X.ptr, n, 4, res))
    assert res[0] == want1 and res[1] == math.sqrt(dot2(L, x, x))
    # VecSum: Sum2 == the correctly rounded sum
    _lib.chk(hx.hipxVecSum(X.ptr, n, C.byref(r)))
    assert r.value == math.fsum(x)
    # VecDotNorm2
    d, m = C.c_double(), C.c_double()
    _lib.chk(hx.hipxVecDotNorm2(X.ptr, Y.ptr, n, C.byref(d), C.byref(m)))
    assert d.value == want and m.value == dot2(L, y, y)
    X.free()
    Y.free()


def test_unaligned_and_mode_switch(hx):
    """8-byte-aligned views take the scalar loop: same exact value.  Fast mode is untouched by a round trip through exact mode."""
    from petsc_amd import _lib
    L = shim()
    n = 70001
    x, y = cancelling(n + 1, 5)
    X, Y = _lib.DVec(n + 1, x), _lib.DVec(n + 1, y)
    r0, r1, r2 = C.c_double(), C.c_double(), C.c_double()
    _lib.chk(hx.hipxVecDot(X.ptr, Y.ptr, n + 1, C.byref(r0)))
    _lib.chk(hx.hipxSetReductionMode(1))
    try:
        _lib.chk(hx.hipxVecDot(X.offset(1), Y.offset(1), n, C.byref(r1)))
        assert r1.value == dot2(L, x[1:], y[1:])
    finally:
        _lib.chk(hx.hipxSetReductionMode(0))
    _lib.chk(hx.hipxVecDot(X.ptr, Y.ptr, n + 1, C.byref(r2)))
    assert r0.value == r2.value  # fast mode: deterministic, unchanged
    # the plain tree is measurably off on this input; the exact mode is not
    ex = dot2(L, x, y)
    _lib.chk(hx.hipxSetReductionMode(1))
    try:
        _lib.chk(hx.hipxVecDot(X.ptr, Y.ptr, n + 1, C.byref(r1)))
    finally:
        _lib.chk(hx.hipxSetReductionMode(0))
    assert r1.value == ex
    X.free()
    Y.free()


@pytest.mark.parametrize("nv", [1, 2, 5, 8, 9, 16, 17, 30, 40])
def test_mdot_bit_identical_to_exact_gemv(exact, nv):
    """VecMDot_Seq_GEMV (dvec2.c:557: dgemv "T") with an exact BLAS = one Dot2 per vector; batches of <= 16 vectors per launch."""
    from petsc_amd import _lib
    hx, L = exact, shim()
    n = 200003
    x, _ = cancelling(n, 3)
    ys = [cancelling(n, 100 + j)[1] * (1.0 + j) for j in range(nv)]
    X = _lib.DVec(n, x)
    Ys = [_lib.DVec(n, y) for y in ys]
    ptrs = (C.c_void_p * nv)(*[v.ptr.value for v in Ys])
    res = (C.c_double * nv)()
    _lib.chk(hx.hipxVecMDot(X.ptr, nv, ptrs, n, res))
    for j in range(nv):
        assert res[j] == dot2(L, x, ys[j]), j
    for v in [X] + Ys:
        v.free()


def test_fused_cg_kernels_and_spmv_dot(exact):
    """hipxCGFusedUpdate's two sums and hipxMatMultDot's dot in exact mode = Dot2 of the vectors the same kernels leave in memory."""
    from petsc_amd import _lib
    hx, L = exact, shim()
    n = 48
    N = n ** 3
    ai, aj, aa = orc.stencil("7pt", n)
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    rng = np.random.default_rng(8)
    p, r0, d = rng.standard_normal(N), rng.standard_normal(N), 1.0 / (5.0 + rng.random(N))
    P, R, Z, W, D = _lib.DVec(N, p), _lib.DVec(N, r0), _lib.DVec(N), _lib.DVec(N), _lib.DVec(N, d)
    dot = C.c_double()
    _lib.chk(hx.hipxMatMultDot(A, P.ptr, W.ptr, C.byref(dot)))
    w = W.get()
    assert np.array_equal(w, orc.matmult(ai, aj, aa, p)) and dot.value == dot2(L, p, w)
    sums = (C.c_double * 2)()
    _lib.chk(hx.hipxCGFusedUpdate(None, R.ptr, Z.ptr, P.ptr, W.ptr, D.ptr, 0.37, N, sums))
    r, z = R.get(), Z.get()
    assert np.array_equal(r, r0 + (-0.37) * w) and np.array_equal(z, r * d)
    assert sums[0] == dot2(L, z, z) and sums[1] == dot2(L, z, r)
    for v in (P, R, Z, W, D):
        v.free()
    _lib.mat_destroy(A)


def host_solve(hx, ks, ksp_name, pcname, ai, aj, aa, b, fused, pipeline, rtol, max_it):
    from petsc_amd import _lib
    N = len(ai) - 1
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    M = _lib.HipxMat(m=N, A=A, B=None, halo=None, lvec=None, nranks=1)
    B, X = _lib.DVec(N, b), _lib.DVec(N, np.zeros(N))
    pc = _lib.HipxPC()
    ks.HipxPCSetDefaults(C.byref(pc))
    pc.type = {"none": 0, "jacobi": 1, "sor": 2}[pcname]
    _lib.chk(ks.HipxPCSetUp(C.byref(pc), C.byref(M)))
    k = _lib.HipxKSP()
    ks.HipxKSPSetDefaults(C.byref(k))
    k.rtol, k.max_it, k.fused, k.pipeline = rtol, max_it, fused, pipeline
    hist = np.zeros(max_it + 40)
    k.history, k.hist_len = hist.ctypes.data, len(hist)
    f = ks.HipxKSPSolve_CG if ksp_name == "cg" else ks.HipxKSPSolve_GMRES
    _lib.chk(f(C.byref(k), C.byref(M), C.byref(pc), B.ptr, X.ptr))
    out = hist[:k.hist_n].copy(), int(k.its), int(k.reason), X.get()
    ks.HipxKSPDestroyWork(C.byref(k))
    ks.HipxPCDestroy(C.byref(pc))
    B.free()
    X.free()
    _lib.mat_destroy(A)
    return out


@pytest.mark.parametrize("stencil,n,pc", [("7pt", 40, "jacobi"), ("7pt", 40, "none"), ("27pt", 24, "jacobi"), ("5pt", 100, "jacobi")])
def test_cg_history_bit_identical_to_exact_reference_arithmetic(exact, stencil, n, pc):
    """KSPCG to convergence: every residual norm equal, bit for bit, to the reference's arithmetic with exact BLAS reductions (the
    oracle's exact mode == the reference's own KSPSolve_CG under oracle/libexactblas.so, bit for bit: tests/test_oracle_exact.py),
    for every form of the host layer -- one kernel per call, fused kernels, fused + launch-ahead.  PCNONE takes the fused loop too."""
    from petsc_amd import _lib
    _, ks = _lib.load()
    ai, aj, aa = orc.stencil(stencil, n)
    N = len(ai) - 1
    b = orc.matmult(ai, aj, aa, np.ones(N))
    xe, ie, re_, he = orc.ksp_solve("cg", ai, aj, aa, b, pc=pc, rtol=1e-10, exact=True)
    assert re_ > 0 and ie > 10
    for fused, pipe in ((0, 0), (1, 0), (1, 1)):
        h, its, reason, x = host_solve(exact, ks, "cg", pc, ai, aj, aa, b, fused, pipe, 1e-10, 10000)
        assert (its, reason) == (ie, re_), (fused, pipe, its, ie)
        assert np.array_equal(h, he), (fused, pipe, float(np.abs(h - he).max()))
        assert np.array_equal(x, xe), (fused, pipe)  # the solution itself: every elementwise kernel and every scalar agree


def test_gmres_sor_history_exact_mode(exact):
    """KSPGMRES(30)+PCSOR, 27-pt 24^3, two restarts: the exact mode follows the exact-reduction yardstick to 1e-12 per entry to convergence."""
    from petsc_amd import _lib
    _, ks = _lib.load()
    ai, aj, aa = orc.stencil("27pt", 24)
    N = len(ai) - 1
    b = orc.matmult(ai, aj, aa, np.ones(N))
    xe, ie, re_, he = orc.ksp_solve("gmres", ai, aj, aa, b, pc="sor", rtol=1e-10, exact=True)
    h, its, reason, x = host_solve(exact, ks, "gmres", "sor", ai, aj, aa, b, 1, 1, 1e-10, 10000)
    assert (its, reason) == (ie, re_)
    rel = np.abs(h - he) / np.abs(he)
    print("GMRES(30)+SOR exact mode: max per-entry relative difference %.3e over %d entries" % (rel.max(), len(he)))
    assert rel.max() <= 1e-12
