"""GPU parity of the Vec BLAS-1 kernels (through the C ABI) against the CPU oracle.
Elementwise kernels: bit-exact (same IEEE operations in the same order, no FMA).  Reductions: the reference calls a
third-party BLAS, so the bar is rounding-level agreement (<= 1e-13 relative here) plus run-to-run determinism."""
import ctypes as C

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu

SIZES = [0, 1, 2, 3, 63, 64, 1023, 1024, 2049, 100003, 1 << 20]


def dv(a):
    from petsc_amd import _lib
    return _lib.DVec(len(a), a)


def rnd(n, seed):
    return np.random.default_rng(seed).standard_normal(n)


@pytest.mark.parametrize("n", SIZES)
def test_elementwise_bit_exact(hx, n):
    from petsc_amd._lib import chk
    L = orc.lib()
    x, y, w = rnd(n, 1), rnd(n, 2), rnd(n, 3)
    if n > 5:
        y[3] = 0.0
        x[3] = 0.0
        y[5] = 0.0
    cases = []
    for a in (0.0, 1.0, -1.0, 0.37, -2.5):
        cases.append(("axpy", lambda Y, X, W, a=a: hx.hipxVecAXPY(Y.ptr, a, X.ptr, n), lambda yy, xx, ww, a=a: L.orc_VecAXPY_Seq(n, orc.P(yy), C.c_double(a), orc.P(xx))))
        cases.append(("aypx", lambda Y, X, W, a=a: hx.hipxVecAYPX(Y.ptr, a, X.ptr, n), lambda yy, xx, ww, a=a: L.orc_VecAYPX_Seq(n, orc.P(yy), C.c_double(a), orc.P(xx))))
        cases.append(("scale", lambda Y, X, W, a=a: hx.hipxVecScale(Y.ptr, n, a), lambda yy, xx, ww, a=a: L.orc_VecScale_Seq(n, orc.P(yy), C.c_double(a))))
        cases.append(("waxpy", lambda Y, X, W, a=a: hx.hipxVecWAXPY(Y.ptr, a, X.ptr, W.ptr, n),
                      lambda yy, xx, ww, a=a: L.orc_VecWAXPY_Seq(n, orc.P(yy), C.c_double(a), orc.P(xx), orc.P(ww))))
        for b in (0.0, 1.0, 0.6):
            cases.append(("axpby", lambda Y, X, W, a=a, b=b: hx.hipxVecAXPBY(Y.ptr, a, b, X.ptr, n),
                          lambda yy, xx, ww, a=a, b=b: L.orc_VecAXPBY_Seq(n, orc.P(yy), C.c_double(a), C.c_double(b), orc.P(xx))))
            for c in (0.0, 1.0, -0.4):
                cases.append(("axpbypcz", lambda Y, X, W, a=a, b=b, c=c: hx.hipxVecAXPBYPCZ(Y.ptr, a, b, c, X.ptr, W.ptr, n),
                              lambda yy, xx, ww, a=a, b=b, c=c: L.orc_VecAXPBYPCZ_Seq(n, orc.P(yy), C.c_double(a), C.c_double(b), C.c_double(c), orc.P(xx), orc.P(ww))))
    cases.append(("pmult", lambda Y, X, W: hx.hipxVecPointwiseMult(Y.ptr, X.ptr, W.ptr, n), lambda yy, xx, ww: L.orc_VecPointwiseMult_Seq(n, orc.P(yy), orc.P(xx), orc.P(ww))))
    cases.append(("pdiv", lambda Y, X, W: hx.hipxVecPointwiseDivide(Y.ptr, W.ptr, X.ptr, n), lambda yy, xx, ww: L.orc_VecPointwiseDivide_Seq(n, orc.P(yy), orc.P(ww), orc.P(xx))))
    cases.append(("recip", lambda Y, X, W: hx.hipxVecReciprocal(Y.ptr, n), lambda yy, xx, ww: L.orc_VecReciprocal(n, orc.P(yy))))
    cases.append(("set", lambda Y, X, W: hx.hipxVecSet(Y.ptr, n, 3.25), lambda yy, xx, ww: L.orc_VecSet_Seq(n, orc.P(yy), C.c_double(3.25))))
    cases.append(("copy", lambda Y, X, W: hx.hipxVecCopy(X.ptr, Y.ptr, n), lambda yy, xx, ww: L.orc_VecCopy_Seq(n, orc.P(xx), orc.P(yy))))
    X, W = dv(x), dv(w)
    Y = dv(y)
    for name, gpu, cpu in cases:
        Y.set(y)
        chk(gpu(Y, X, W))
        yy = y.copy()
        cpu(yy, x.copy(), w.copy())
        got = Y.get()
        assert np.array_equal(got, yy, equal_nan=True), (name, n, np.abs(got - yy).max() if n else 0)
    for v in (X, W, Y):
        v.free()


@pytest.mark.parametrize("n", [5, 1001, 65536 + 3])
def test_unaligned_views_and_aliasing(hx, n):
    """Sub-vector views start at 8-byte (not 16-byte) boundaries: the scalar path must give the same bits."""
    from petsc_amd import _lib
    x, y = rnd(n + 1, 4), rnd(n + 1, 5)
    X, Y = dv(x), dv(y)
    _lib.chk(hx.hipxVecAXPY(Y.offset(1), 0.7, X.offset(1), n))
    ref = y.copy()
    orc.lib().orc_VecAXPY_Seq(n, orc.P(ref[1:]), C.c_double(0.7), orc.P(np.ascontiguousarray(x[1:])))
    exp = y.copy()
    exp[1:] = y[1:] + 0.7 * x[1:]
    assert np.array_equal(Y.get(), exp)
    # w aliases x in PointwiseMult (bvec2.c:81-84)
    X.set(x)
    _lib.chk(hx.hipxVecPointwiseMult(X.ptr, X.ptr, Y.ptr, n + 1))
    assert np.array_equal(X.get(), x * Y.get())
    X.free()
    Y.free()


@pytest.mark.parametrize("n", SIZES)
def test_reductions(hx, n):
    from petsc_amd._lib import chk
    x, y = rnd(n, 6), rnd(n, 7)
    X, Y = dv(x), dv(y)
    r = C.c_double()
    chk(hx.hipxVecDot(X.ptr, Y.ptr, n, C.byref(r)))
    ref = orc.lib().orc_VecDot_Seq(n, orc.P(x), orc.P(y))
    scale = np.abs(x * y).sum() + 1e-300
    assert abs(r.value - ref) <= 1e-14 * scale + 1e-300
    r2 = C.c_double()
    chk(hx.hipxVecDot(X.ptr, Y.ptr, n, C.byref(r2)))
    assert r2.value == r.value  # deterministic
    for t, npn in ((0, lambda v: np.abs(v).sum()), (1, lambda v: np.sqrt((v * v).sum())), (3, lambda v: np.abs(v).max() if len(v) else 0.0)):
        out = (C.c_double * 2)()
        chk(hx.hipxVecNorm(X.ptr, n, t, out))
        assert abs(out[0] - npn(x)) <= 1e-13 * (npn(x) + 1e-300)
    out = (C.c_double * 2)()
    chk(hx.hipxVecNorm(X.ptr, n, 4, out))
    assert abs(out[0] - np.abs(x).sum()) <= 1e-13 * (np.abs(x).sum() + 1e-300) and abs(out[1] - np.linalg.norm(x)) <= 1e-13 * (np.linalg.norm(x) + 1e-300)
    dp, nm = C.c_double(), C.c_double()
    chk(hx.hipxVecDotNorm2(X.ptr, Y.ptr, n, C.byref(dp), C.byref(nm)))
    assert abs(dp.value - ref) <= 1e-14 * scale + 1e-300 and abs(nm.value - (y * y).sum()) <= 1e-13 * ((y * y).sum() + 1e-300)
    s = C.c_double()
    chk(hx.hipxVecSum(X.ptr, n, C.byref(s)))
    assert abs(s.value - x.sum()) <= 1e-13 * (np.abs(x).sum() + 1e-300)
    if n:
        idx, val = C.c_int32(), C.c_double()
        chk(hx.hipxVecMax(X.ptr, n, C.byref(idx), C.byref(val)))
        assert val.value == x.max() and idx.value == int(np.argmax(x))
        chk(hx.hipxVecMin(X.ptr, n, C.byref(idx), C.byref(val)))
        assert val.value == x.min() and idx.value == int(np.argmin(x))
    X.free()
    Y.free()


def test_norm_inf_propagates_nan_like_reference(hx):
    from petsc_amd._lib import chk
    x = rnd(5000, 8)
    x[1234] = np.nan
    X = dv(x)
    out = (C.c_double * 2)()
    chk(hx.hipxVecNorm(X.ptr, len(x), 3, out))
    assert np.isnan(out[0]) and np.isnan(orc.lib().orc_VecNorm_Seq(len(x), orc.P(x), 3, None))
    X.free()


@pytest.mark.parametrize("n", [40009, 40010, 7])
@pytest.mark.parametrize("nv", [1, 2, 3, 4, 5, 7, 8, 9, 13, 16, 17, 30, 31, 32, 33, 35, 36, 40])
def test_mdot_maxpy(hx, nv, n):
    """VecMDot / VecMAXPY / VecMAXPBY for 1..40 vectors: up to 32 dot products and 36 AXPYs in ONE pass (GMRES(30)'s
    orthogonalisation), more in batches.  The dot products do not depend on the batching (each equals hipxVecDot of the pair bit
    for bit); MAXPY / MAXPBY keep the grouping of dvec2.c:658-693 -> bit-identical to the reference loops; odd and even
    lengths, beta = 0 / 1 / other."""
    from petsc_amd import _lib
    x = rnd(n, 9)
    ys = [rnd(n, 100 + j) for j in range(nv)]
    X = dv(x)
    Ys = [dv(v) for v in ys]
    ptrs = (C.c_void_p * nv)(*[v.ptr.value for v in Ys])
    res = (C.c_double * nv)()
    _lib.chk(hx.hipxVecMDot(X.ptr, nv, ptrs, n, res))
    one = C.c_double()
    for j in range(nv):
        assert abs(res[j] - float(x @ ys[j])) <= 1e-13 * np.abs(x * ys[j]).sum()
        _lib.chk(hx.hipxVecDot(X.ptr, Ys[j].ptr, n, C.byref(one)))
        assert res[j] == one.value
    alpha = rnd(nv, 10)
    al = (C.c_double * nv)(*alpha)
    _lib.chk(hx.hipxVecMAXPY(X.ptr, nv, al, ptrs, n))
    ref = x.copy()
    yp = (C.c_void_p * nv)(*[v.ctypes.data for v in ys])
    orc.lib().orc_VecMAXPY_Seq(n, orc.P(ref), nv, orc.P(alpha), yp)
    assert np.array_equal(X.get(), ref)  # same grouping as dvec2.c:658-693 -> bit-exact
    _lib.chk(hx.hipxVecMAXPBY(X.ptr, nv, al, 0.0, ptrs, n))
    ref2 = x.copy()
    orc.lib().orc_VecMAXPBY(n, orc.P(ref2), nv, orc.P(alpha), C.c_double(0.0), yp)
    assert np.array_equal(X.get(), ref2)
    for beta in (1.0, -0.75):
        X.set(x)
        _lib.chk(hx.hipxVecMAXPBY(X.ptr, nv, al, beta, ptrs, n))
        ref3 = x.copy()
        orc.lib().orc_VecMAXPBY(n, orc.P(ref3), nv, orc.P(alpha), C.c_double(beta), yp)
        assert np.array_equal(X.get(), ref3)
    X.free()
    for v in Ys:
        v.free()


def test_full_size_vector_properties(hx):
    """BASELINE size (N = 256^3): linearity / known sums instead of an element-wise oracle pass."""
    from petsc_amd import _lib
    n = 256 ** 3
    X = _lib.DVec(n)
    Y = _lib.DVec(n)
    _lib.chk(hx.hipxVecSet(X.ptr, n, 0.5))
    _lib.chk(hx.hipxVecSet(Y.ptr, n, 2.0))
    _lib.chk(hx.hipxVecAXPY(Y.ptr, 4.0, X.ptr, n))  # y = 4
    r = C.c_double()
    _lib.chk(hx.hipxVecDot(X.ptr, Y.ptr, n, C.byref(r)))
    assert r.value == 2.0 * n  # exact: all partial sums are integers < 2^53
    out = (C.c_double * 2)()
    _lib.chk(hx.hipxVecNorm(Y.ptr, n, 1, out))
    assert out[0] == 4.0 * 4096.0  # sqrt(16 * 2^24)
    X.free()
    Y.free()
