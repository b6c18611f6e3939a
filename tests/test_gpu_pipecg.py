"""GPU parity of the pipelined CG variants (round 6).

* HipxKSPSolve_PIPECG (host/hipx_ksp.c: one fused update kernel + one product per iteration, scalars on the device, launch-ahead) against the oracle's
  statement-by-statement restatement of KSPSolve_PIPECG (pipecg.c:20-160), which tests/test_oracle_exact.py pins bit for bit to the REFERENCE run with exact
  BLAS reductions.  Bar: identical iteration counts / reasons; histories equal BIT FOR BIT in the exact reduction mode (north_star's 1e-12 is then met with
  0.0), within rounding of the reductions in the fast mode; launch-ahead == host-synchronised bit for bit; committed reference+shim goldens at 64^3 .. 256^3.
* hipxVecBatchAXPYDotsBegin (csrc/hipx_pipe.hip: a recorded batch of VecAXPY / VecAYPX as one pass) against the separate kernels: every vector bit-identical,
  the sums equal to hipxVecDot's in the exact mode.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "exact_histories.json")))


def solve_pipecg(ai, aj, aa, b, pc="jacobi", rtol=1e-5, max_it=10000, normtype=1, pipeline=1, x0=None, exact=False, kind="pipecg"):
    from petsc_amd import _lib
    hx, ks = _lib.load()
    N = len(ai) - 1
    if exact:
        _lib.chk(hx.hipxSetReductionMode(1))
    try:
        A = _lib.mat_create_csr(N, N, ai, aj, aa)
        M = _lib.HipxMat(m=N, A=A, B=None, halo=None, lvec=None, nranks=1)
        p = _lib.HipxPC()
        ks.HipxPCSetDefaults(C.byref(p))
        p.type = {"none": 0, "jacobi": 1, "sor": 2}[pc]
        k = _lib.HipxKSP()
        ks.HipxKSPSetDefaults(C.byref(k))
        k.rtol, k.max_it, k.normtype, k.pipeline = rtol, max_it, normtype, pipeline
        hist = np.zeros(max_it + 100)
        k.history, k.hist_len = hist.ctypes.data, len(hist)
        B = _lib.DVec(N, b)
        X = _lib.DVec(N, x0 if x0 is not None else np.zeros(N))
        k.guess_nonzero = 0 if x0 is None else 1
        _lib.chk(ks.HipxPCSetUp(C.byref(p), C.byref(M)))
        _lib.chk({"pipecg": ks.HipxKSPSolve_PIPECG, "groppcg": ks.HipxKSPSolve_GROPPCG}[kind](C.byref(k), C.byref(M), C.byref(p), B.ptr, X.ptr))
        x = X.get()
        out = (x, int(k.its), int(k.reason), hist[:k.hist_n].copy())
        ks.HipxKSPDestroyWork(C.byref(k))
        ks.HipxPCDestroy(C.byref(p))
        B.free()
        X.free()
        _lib.mat_destroy(A)
        return out
    finally:
        if exact:
            _lib.chk(hx.hipxSetReductionMode(0))


def system(kind, n, scale=None):
    ai, aj, aa = orc.stencil(kind, n)
    if scale is not None:  # a variable diagonal (the streamed-diagonal form of the kernel): D A D, still symmetric positive definite
        N = len(ai) - 1
        d = 1.0 + 0.5 * ((np.arange(N) * 7919) % 13) / 13.0
        rows = np.repeat(np.arange(N), np.diff(ai))
        aa = aa * d[rows] * d[aj]
    b = orc.matmult(ai, aj, aa, np.ones(len(ai) - 1))
    return ai, aj, aa, b


def golden_system(kind, n):
    """the operator and b = A * 1 of a committed golden history; 256^3 is assembled by the host layer's own driver loops (no 1.9 GB of Python lists)"""
    from petsc_amd import _lib
    hx, ks = _lib.load()
    N = n ** 3
    if n < 200:
        ai, aj, aa = orc.stencil(kind, n)
        return ai, aj, aa, orc.matmult(ai, aj, aa, np.ones(N))
    assert kind == "7pt"
    ai = np.zeros(N + 1, np.int32)
    nnz = ks.HipxAssemble_poisson7(n, 0, N, None, None, None)
    aj, aa = np.zeros(nnz, np.int32), np.zeros(nnz)
    ks.HipxAssemble_poisson7(n, 0, N, ai.ctypes.data_as(C.c_void_p), aj.ctypes.data_as(C.c_void_p), aa.ctypes.data_as(C.c_void_p))
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    one, bb = _lib.DVec(N, np.ones(N)), _lib.DVec(N)
    _lib.chk(hx.hipxMatMult(A, one.ptr, bb.ptr))
    b = bb.get()
    one.free()
    bb.free()
    _lib.mat_destroy(A)
    return ai, aj, aa, b


CASES = [("7pt", 20, "jacobi", 1, None), ("7pt", 20, "none", 2, None), ("7pt", 20, "jacobi", 3, None), ("27pt", 16, "jacobi", 1, None), ("7pt", 24, "jacobi", 1, True),
         ("27pt", 12, "jacobi", 2, True), ("5pt", 40, "jacobi", 1, None), ("7pt", 17, "none", 1, None), ("7pt", 20, "jacobi", 0, None)]


@pytest.mark.parametrize("kind,n,pc,normtype,scale", CASES)
def test_pipecg_exact_mode_equals_the_oracle_bit_for_bit(hx, kind, n, pc, normtype, scale):
    ai, aj, aa, b = system(kind, n, scale)
    mi = 25 if normtype == 0 else 10000
    xo, its_o, reason_o, ho = orc.ksp_solve("pipecg", ai, aj, aa, b, pc=pc, rtol=1e-8, max_it=mi, normtype=normtype, exact=True)
    for pipeline in (1, 0):
        xg, its, reason, hg = solve_pipecg(ai, aj, aa, b, pc=pc, rtol=1e-8, max_it=mi, normtype=normtype, pipeline=pipeline, exact=True)
        assert (its, reason) == (its_o, reason_o), (pipeline, its, its_o, reason, reason_o)
        assert np.array_equal(hg, ho), (pipeline, np.abs(hg - ho).max())
        assert np.array_equal(xg, xo), (pipeline, np.abs(xg - xo).max())  # elementwise operations in the reference's order: the solution too


@pytest.mark.parametrize("kind,n,pc,normtype,scale", CASES[:6])
def test_pipecg_fast_mode_within_rounding_and_launch_ahead_equals_synchronised(hx, kind, n, pc, normtype, scale):
    ai, aj, aa, b = system(kind, n, scale)
    xo, its_o, reason_o, ho = orc.ksp_solve("pipecg", ai, aj, aa, b, pc=pc, rtol=1e-8, max_it=10000, normtype=normtype, exact=True)
    x1, its1, r1, h1 = solve_pipecg(ai, aj, aa, b, pc=pc, rtol=1e-8, normtype=normtype, pipeline=1)
    x0, its0, r0, h0 = solve_pipecg(ai, aj, aa, b, pc=pc, rtol=1e-8, normtype=normtype, pipeline=0)
    assert (its1, r1) == (its0, r0) and np.array_equal(h1, h0) and np.array_equal(x1, x0)
    assert abs(its1 - its_o) <= 1 and r1 == reason_o
    m = min(len(h1), len(ho), 12)
    assert (np.abs(h1[:m] - ho[:m]) / ho[:m]).max() <= 1e-12, (np.abs(h1[:m] - ho[:m]) / ho[:m]).max()  # leading entries: north_star's tolerance
    m = min(len(h1), len(ho))
    assert (np.abs(h1[:m] - ho[:m]) / ho[:m]).max() <= 1e-6  # (the pipelined recurrences carry every rounding forward: pipecg.c's own MKL run is 4e-8 away)


def test_pipecg_loop_bound_and_nonzero_guess(hx):
    ai, aj, aa, b = system("7pt", 16)
    N = len(ai) - 1
    xo, its_o, reason_o, ho = orc.ksp_solve("pipecg", ai, aj, aa, b, pc="jacobi", rtol=1e-30, max_it=7, exact=True)
    for pipeline in (1, 0):
        xg, its, reason, hg = solve_pipecg(ai, aj, aa, b, pc="jacobi", rtol=1e-30, max_it=7, pipeline=pipeline, exact=True)
        assert (its, reason) == (its_o, reason_o) == (8, -3)  # pipecg.c:158-160: `i <= max_it`
        assert np.array_equal(hg, ho) and np.array_equal(xg, xo)
    x0 = 0.5 + (np.arange(N) % 5) / 10.0
    xo, its_o, reason_o, ho = orc.ksp_solve("pipecg", ai, aj, aa, b, pc="jacobi", rtol=1e-8, x0=x0, exact=True)
    xg, its, reason, hg = solve_pipecg(ai, aj, aa, b, pc="jacobi", rtol=1e-8, x0=x0, exact=True)
    assert (its, reason) == (its_o, reason_o) and np.array_equal(hg, ho) and np.array_equal(xg, xo)


@pytest.mark.parametrize("key,kind,n,its", [("pipecg_jacobi_7pt_64", "7pt", 64, 40), ("pipecg_jacobi_7pt_128", "7pt", 128, 40), ("pipecg_jacobi_27pt_96", "27pt", 96, 30),
                                            ("pipecg_jacobi_7pt_256", "7pt", 256, 40)])
def test_pipecg_against_the_reference_with_exact_blas(hx, key, kind, n, its):
    """The committed histories of the REFERENCE's KSPSolve_PIPECG + exact BLAS (tests/golden/make_exact_golden.py), at up to BASELINE config 2's size."""
    if key not in GOLD:
        pytest.skip("golden %s not generated" % key)
    ai, aj, aa, b = golden_system(kind, n)
    href = np.array([float.fromhex(v) for v in GOLD[key]["history_hex"]])
    xg, it, reason, hg = solve_pipecg(ai, aj, aa, b, pc="jacobi", rtol=1e-50, max_it=its, exact=True)
    assert len(hg) == len(href) == its + 1 and (it, reason) == (its + 1, -3)
    assert np.array_equal(hg, href), (np.abs(hg - href) / href).max()
    xf, it, reason, hf = solve_pipecg(ai, aj, aa, b, pc="jacobi", rtol=1e-50, max_it=its)
    m = 12
    assert (np.abs(hf[:m] - href[:m]) / href[:m]).max() <= 1e-12
    assert (np.abs(hf - href) / href).max() <= 1e-7


# ---------------------------------------------------------------- Gropp's CG through the host layer (HipxKSPSolve_GROPPCG)
@pytest.mark.parametrize("kind,n,pc,normtype,scale", CASES)
def test_groppcg_exact_mode_equals_the_oracle_bit_for_bit(hx, kind, n, pc, normtype, scale):
    """HipxKSPSolve_GROPPCG (two fused passes + one product per iteration, alpha / beta on the device, launch-ahead) against the oracle's restatement of
    groppcg.c:23-140 (pinned bit for bit to the reference + exact BLAS, tests/test_oracle_exact.py): history, iteration count, reason and SOLUTION equal."""
    ai, aj, aa, b = system(kind, n, scale)
    mi = 25 if normtype == 0 else 10000
    xo, its_o, reason_o, ho = orc.ksp_solve("groppcg", ai, aj, aa, b, pc=pc, rtol=1e-8, max_it=mi, normtype=normtype, exact=True)
    for pipeline in (1, 0):
        xg, its, reason, hg = solve_pipecg(ai, aj, aa, b, pc=pc, rtol=1e-8, max_it=mi, normtype=normtype, pipeline=pipeline, exact=True, kind="groppcg")
        assert (its, reason) == (its_o, reason_o), (pipeline, its, its_o, reason, reason_o)
        assert np.array_equal(hg, ho), (pipeline, np.abs(hg - ho).max())
        assert np.array_equal(xg, xo), (pipeline, np.abs(xg - xo).max())


def test_groppcg_fast_mode_loop_bound_and_nonzero_guess(hx):
    ai, aj, aa, b = system("7pt", 20)
    N = len(ai) - 1
    xo, its_o, reason_o, ho = orc.ksp_solve("groppcg", ai, aj, aa, b, pc="jacobi", rtol=1e-8, exact=True)
    x1, its1, r1, h1 = solve_pipecg(ai, aj, aa, b, rtol=1e-8, pipeline=1, kind="groppcg")
    x0, its0, r0, h0 = solve_pipecg(ai, aj, aa, b, rtol=1e-8, pipeline=0, kind="groppcg")
    assert (its1, r1) == (its0, r0) and np.array_equal(h1, h0) and np.array_equal(x1, x0)  # launch-ahead == host-synchronised
    assert abs(its1 - its_o) <= 1 and r1 == reason_o
    m = min(len(h1), len(ho), 12)
    assert (np.abs(h1[:m] - ho[:m]) / ho[:m]).max() <= 1e-12
    assert (np.abs(h1[:min(len(h1), len(ho))] - ho[:min(len(h1), len(ho))]) / ho[:min(len(h1), len(ho))]).max() <= 1e-6
    xo, its_o, reason_o, ho = orc.ksp_solve("groppcg", ai, aj, aa, b, pc="jacobi", rtol=1e-30, max_it=7, exact=True)
    for pipeline in (1, 0):
        xg, its, reason, hg = solve_pipecg(ai, aj, aa, b, rtol=1e-30, max_it=7, pipeline=pipeline, exact=True, kind="groppcg")
        assert (its, reason) == (its_o, reason_o) == (7, -3) and np.array_equal(hg, ho) and np.array_equal(xg, xo)  # groppcg.c:137-139: `i < max_it`
    g = 0.5 + (np.arange(N) % 5) / 10.0
    xo, its_o, reason_o, ho = orc.ksp_solve("groppcg", ai, aj, aa, b, pc="jacobi", rtol=1e-8, x0=g, exact=True)
    xg, its, reason, hg = solve_pipecg(ai, aj, aa, b, rtol=1e-8, x0=g, exact=True, kind="groppcg")
    assert (its, reason) == (its_o, reason_o) and np.array_equal(hg, ho) and np.array_equal(xg, xo)


@pytest.mark.parametrize("key,kind,n,its", [("groppcg_jacobi_7pt_64", "7pt", 64, 40), ("groppcg_jacobi_7pt_128", "7pt", 128, 40), ("groppcg_jacobi_27pt_96", "27pt", 96, 30),
                                            ("groppcg_jacobi_7pt_256", "7pt", 256, 40)])
def test_groppcg_against_the_reference_with_exact_blas(hx, key, kind, n, its):
    """The committed histories of the REFERENCE's KSPSolve_GROPPCG + exact BLAS (tests/golden/make_exact_golden.py)."""
    if key not in GOLD:
        pytest.skip("golden %s not generated" % key)
    ai, aj, aa, b = golden_system(kind, n)
    href = np.array([float.fromhex(v) for v in GOLD[key]["history_hex"]])
    xg, it, reason, hg = solve_pipecg(ai, aj, aa, b, pc="jacobi", rtol=1e-50, max_it=its, exact=True, kind="groppcg")
    assert len(hg) == len(href) == its + 1 and (it, reason) == (its, -3)
    assert np.array_equal(hg, href), (np.abs(hg - href) / href).max()


# ---------------------------------------------------------------- the recorded batches
PROGS = {  # (kind, y, x) per operation, slots numbered by first appearance; the sums the kernel leaves
    "pipecg": ([2, 2, 2, 2, 1, 1, 1, 1], [0, 2, 4, 6, 8, 5, 7, 9], [1, 3, 5, 7, 4, 2, 0, 6], 10, [(5, 5), (9, 9), (9, 5), (7, 5)]),
    "pipecg_first": ([1, 1, 1, 1], [0, 2, 4, 6], [1, 3, 5, 7], 8, [(2, 2), (6, 6), (6, 2), (4, 2)]),
    "groppcg_a": ([1, 1, 1], [0, 2, 4], [1, 3, 5], 6, [(4, 4), (2, 2), (2, 4)]),
    "groppcg_b": ([2, 2], [0, 2], [1, 3], 4, [(0, 2)]),
    "pipecr": ([2, 2, 2, 1, 1, 1], [0, 2, 4, 6, 5, 7], [1, 3, 5, 4, 2, 0], 8, [(5, 5), (7, 5)]),
}


@pytest.mark.parametrize("name", sorted(PROGS))
@pytest.mark.parametrize("n", [1, 2, 1023, 4096 + 3, 300001])
@pytest.mark.parametrize("exact", [False, True])
def test_batch_kernel_equals_the_separate_calls(hx, name, n, exact):
    from petsc_amd import _lib
    hx, _ = _lib.load()
    kind, ys, xs, nvec, dots = PROGS[name]
    rng = np.random.default_rng(n * 31 + len(kind))
    host = [rng.standard_normal(n) for _ in range(nvec)]
    s = [(-1.0) ** k * (0.3 + 0.17 * k) for k in range(len(kind))]
    if name == "groppcg_b":
        s[1] = -1.0  # VecAYPX's beta = -1 branch (dvec2.c:767: x - y) is the general expression's bits
    if name == "pipecg_first":
        s[0] = 1.0
    if exact:
        _lib.chk(hx.hipxSetReductionMode(1))
    try:
        a = [_lib.DVec(n, h) for h in host]      # separate kernels
        bvec = [_lib.DVec(n, h) for h in host]   # one batch
        for k in range(len(kind)):
            if kind[k] == 1:
                _lib.chk(hx.hipxVecAXPY(a[ys[k]].ptr, s[k], a[xs[k]].ptr, n))
            else:
                _lib.chk(hx.hipxVecAYPX(a[ys[k]].ptr, s[k], a[xs[k]].ptr, n))
        ptrs = (C.c_void_p * nvec)(*[v.ptr.value for v in bvec])
        nd = C.c_int(0)
        da, db = (C.c_int * 4)(), (C.c_int * 4)()
        ka, ya, xa = (C.c_int * len(kind))(*kind), (C.c_int * len(kind))(*ys), (C.c_int * len(kind))(*xs)
        sa = (C.c_double * len(kind))(*s)
        _lib.chk(hx.hipxVecBatchAXPYDotsBegin(len(kind), ka, ya, xa, sa, nvec, ptrs, n, 10, C.byref(nd), da, db))
        assert nd.value == len(dots) and [(da[k], db[k]) for k in range(nd.value)] == dots
        sums = (C.c_double * 4)()
        _lib.chk(hx.hipxRedEnd(10, nd.value, sums))
        for v in range(nvec):
            assert np.array_equal(a[v].get(), bvec[v].get()), (name, v)
        for k, (p, q) in enumerate(dots):
            r = C.c_double(0)
            _lib.chk(hx.hipxVecDot(a[p].ptr, a[q].ptr, n, C.byref(r)))
            if exact:
                assert sums[k] == r.value, (name, k)
            else:
                ref = float(np.dot(a[p].get(), a[q].get()))
                assert abs(sums[k] - ref) <= 1e-13 * max(np.abs(a[p].get()) @ np.abs(a[q].get()), 1e-300)
        for v in a + bvec:
            v.free()
    finally:
        if exact:
            _lib.chk(hx.hipxSetReductionMode(0))


def test_batch_kernel_declines_unknown_programs_and_unaligned_vectors_work(hx):
    from petsc_amd import _lib
    hx, _ = _lib.load()
    n = 5000
    v = [_lib.DVec(n, np.full(n, 1.0 + k)) for k in range(3)]
    ptrs = (C.c_void_p * 3)(*[x.ptr.value for x in v])
    nd = C.c_int(7)
    da, db = (C.c_int * 4)(), (C.c_int * 4)()
    _lib.chk(hx.hipxVecBatchAXPYDotsBegin(2, (C.c_int * 2)(1, 2), (C.c_int * 2)(0, 2), (C.c_int * 2)(1, 0), (C.c_double * 2)(0.5, 0.25), 3, ptrs, n, 10, C.byref(nd), da, db))
    assert nd.value == -1  # not a compiled program: nothing ran
    for k in range(3):
        assert np.array_equal(v[k].get(), np.full(n, 1.0 + k))
        v[k].free()
    # vectors that start 8 bytes off a 16-byte boundary take the scalar path: same bits
    kind, ys, xs, nvec, dots = PROGS["groppcg_a"]
    big = [_lib.DVec(n + 1, np.arange(n + 1) * (0.1 + k % nvec)) for k in range(2 * nvec)]
    s = [0.5, -0.25, 0.125]
    for k in range(3):
        _lib.chk(hx.hipxVecAXPY(big[ys[k]].offset(1), s[k], big[xs[k]].offset(1), n))
    ptrs = (C.c_void_p * nvec)(*[big[nvec + k].offset(1).value for k in range(nvec)])
    _lib.chk(hx.hipxVecBatchAXPYDotsBegin(3, (C.c_int * 3)(*kind), (C.c_int * 3)(*ys), (C.c_int * 3)(*xs), (C.c_double * 3)(*s), nvec, ptrs, n, 10, C.byref(nd), da, db))
    sums = (C.c_double * 4)()
    _lib.chk(hx.hipxRedEnd(10, nd.value, sums))
    for k in range(nvec):
        assert np.array_equal(big[k].get(), big[nvec + k].get())
    for x in big:
        x.free()
