"""GPU parity of the CSR kernels against the oracle's MatMult_SeqAIJ / MatMultAdd_SeqAIJ / MatGetDiagonal restatements.
Bar: bit-exact y for every row that fits the LDS tile (left-to-right row sums, no FMA); rows longer than the tile use a
tree sum and must agree to 1e-14 relative."""
import ctypes as C

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


def xvec(n):
    return 1.0 + (np.arange(n) % 17) / 17.0


def spmv_gpu(hx, ai, aj, aa, x, ncols=None, variant=0, y0=None):
    from petsc_amd import _lib
    m = len(ai) - 1
    n = ncols if ncols is not None else m
    A = _lib.mat_create_csr(m, n, ai, aj, aa)
    _lib.chk(hx.hipxMatSetSpMVVariant(A, variant))
    X, Y = _lib.DVec(n, x), _lib.DVec(m)
    if y0 is None:
        _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    else:
        Y0 = _lib.DVec(m, y0)
        _lib.chk(hx.hipxMatMultAdd(A, X.ptr, Y0.ptr, Y.ptr))
        Y0.free()
    y = Y.get()
    X.free()
    Y.free()
    _lib.mat_destroy(A)
    return y


@pytest.mark.parametrize("kind,n,m", [("5pt", 100, 100), ("5pt", 7, 8), ("7pt", 1, None), ("7pt", 2, None), ("7pt", 33, None), ("7pt", 64, None),
                                       ("27pt", 3, None), ("27pt", 17, None), ("27pt", 40, None)])
@pytest.mark.parametrize("variant", [1, 2, 3, 22, 23, 24, 25, 26, 28, 29, 101])
def test_stencil_spmv_bit_exact(hx, kind, n, m, variant):
    ai, aj, aa = orc.stencil(kind, n, m=m)
    N = len(ai) - 1
    x = xvec(N)
    y = spmv_gpu(hx, ai, aj, aa, x, variant=variant)
    assert np.array_equal(y, orc.matmult(ai, aj, aa, x))


def random_csr(m, n, rng, maxlen, empty_frac=0.2):
    lens = rng.integers(0, maxlen + 1, size=m)
    lens[rng.random(m) < empty_frac] = 0
    ai = np.zeros(m + 1, np.int32)
    ai[1:] = np.cumsum(lens)
    aj = np.zeros(ai[-1], np.int32)
    for r in range(m):
        k = lens[r]
        if k:
            aj[ai[r]:ai[r + 1]] = np.sort(rng.choice(n, size=min(k, n), replace=False))[:k]
    aa = rng.standard_normal(ai[-1])
    return ai, aj, aa


@pytest.mark.parametrize("seed,m,n,maxlen", [(0, 1, 1, 1), (1, 17, 29, 5), (2, 1000, 777, 40), (3, 5000, 5000, 8), (4, 300, 4000, 300), (5, 257, 100, 0), (7, 2000, 9000, 160)])
@pytest.mark.parametrize("variant", [1, 22, 23, 24, 25, 26, 28, 29])
def test_ragged_rows_bit_exact_and_multadd(hx, seed, m, n, maxlen, variant):
    rng = np.random.default_rng(seed)
    ai, aj, aa = random_csr(m, n, rng, maxlen)
    x = rng.standard_normal(n)
    y = spmv_gpu(hx, ai, aj, aa, x, ncols=n, variant=variant)
    assert np.array_equal(y, orc.matmult(ai, aj, aa, x))
    y0 = rng.standard_normal(m)
    z = spmv_gpu(hx, ai, aj, aa, x, ncols=n, y0=y0, variant=variant)
    zr = np.zeros(m)
    orc.lib().orc_MatMult_SeqAIJ_dispatch(m, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(x), orc.P(y0), orc.P(zr), 0)
    assert np.array_equal(z, zr)


def kernel_name(hx, A):
    buf = C.create_string_buffer(256)
    from petsc_amd import _lib
    _lib.chk(hx.hipxMatGetSpMVKernel(A, buf, 256))
    return buf.value.decode()


@pytest.mark.parametrize("seed,m,n,maxlen", [(1, 17, 29, 5), (2, 1000, 777, 40), (3, 5000, 5000, 8), (4, 300, 4000, 300), (6, 1500, 200000, 12)])
@pytest.mark.parametrize("variant", [24, 25])
def test_value_dictionary_ragged_rows_bit_exact(hx, seed, m, n, maxlen, variant):
    """Few distinct values (incl. -0.0, a denormal, +-inf-free extremes) on irregular rows: the 8-bit value-code kernels must
    reproduce MatMult_SeqAIJ bit for bit, also after hipxMatUpdateValues (dictionary rebuilt) and when the new values no
    longer fit a dictionary (falls back to the 8-byte value stream)."""
    from petsc_amd import _lib
    rng = np.random.default_rng(seed)
    ai, aj, _ = random_csr(m, n, rng, maxlen)
    pool = np.array([-1.0, 6.0, -0.0, 0.0, 4.9e-324, 1.0e150, -3.0 / 13.0, 44.0 / 13.0, 1e-17])
    aa = pool[rng.integers(0, len(pool), size=ai[-1])]
    x = rng.standard_normal(n)
    A = _lib.mat_create_csr(m, n, ai, aj, aa)
    _lib.chk(hx.hipxMatSetSpMVVariant(A, variant))
    X, Y = _lib.DVec(n, x), _lib.DVec(m)
    if ai[-1]:
        assert "value dictionary" in kernel_name(hx, A)
    _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    yr = orc.matmult(ai, aj, aa, x)
    assert np.array_equal(Y.get(), yr) and np.array_equal(np.signbit(Y.get()), np.signbit(yr))
    aa2 = (pool * 0.5)[rng.integers(0, len(pool), size=ai[-1])]       # other dictionary
    _lib.chk(hx.hipxMatUpdateValues(A, orc.P(aa2)))
    _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    assert np.array_equal(Y.get(), orc.matmult(ai, aj, aa2, x))
    aa3 = rng.standard_normal(ai[-1])                                 # all distinct: no dictionary
    _lib.chk(hx.hipxMatUpdateValues(A, orc.P(aa3)))
    if ai[-1] > 300:
        assert "value dictionary" not in kernel_name(hx, A)
    _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    assert np.array_equal(Y.get(), orc.matmult(ai, aj, aa3, x))
    X.free()
    Y.free()
    _lib.mat_destroy(A)


TEMPLATE_KERNELS = ("spmv_tmpl_kernel", "spmv_pair_kernel", "spmv_march_kernel", "spmv_march2_kernel")  # row templates: the general walk, the pair form (every row a subset of the interior row), the march form (three-plane base rows)


def is_template_kernel(name):
    return name.startswith(tuple(k + " " for k in TEMPLATE_KERNELS))


def test_auto_variant_selects_packed_kernels(hx):
    """variant 0 on >= 2^20 nonzeros: short rows on <= 256 row patterns -> pattern templates (values streamed), other short rows ->
    row-parallel packed kernel, long rows -> staged packed kernel; constant
    coefficient stencils get the row templates (1 byte per row), matrices with distinct values do not.  (Guards the default path.)"""
    from petsc_amd import _lib
    rng = np.random.default_rng(3)
    for kind, n, want in [("7pt", 56, TEMPLATE_KERNELS), ("27pt", 36, TEMPLATE_KERNELS)]:
        ai, aj, aa = orc.stencil(kind, n)
        N = len(ai) - 1
        x = xvec(N)
        for vals, w in [(aa, want), (aa * (1.0 + 1e-3 * rng.standard_normal(aa.size)), "spmv_tp_kernel" if kind == "7pt" else "spmv_pk16_kernel")]:  # round 3: short rows on a stencil pattern with arbitrary values -> pattern templates
            A = _lib.mat_create_csr(N, N, ai, aj, vals)
            assert kernel_name(hx, A).startswith(tuple(k + " " for k in ((w,) if isinstance(w, str) else w))), kernel_name(hx, A)
            _lib.chk(hx.hipxMatSetSpMVVariant(A, 0))  # explicit "auto" must not change the selection
            assert kernel_name(hx, A).startswith(tuple(k + " " for k in ((w,) if isinstance(w, str) else w)))
            X, Y = _lib.DVec(N, x), _lib.DVec(N)
            _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
            assert np.array_equal(Y.get(), orc.matmult(ai, aj, vals, x))
            X.free()
            Y.free()
            _lib.mat_destroy(A)


@pytest.mark.parametrize("kind,n,m", [("5pt", 31, 17), ("7pt", 21, None), ("27pt", 14, None)])
def test_row_templates_kernel_selected_and_bit_exact(hx, kind, n, m):
    """Variant 26: stencil matrices are stored as one template id per row.  Same doubles, same left-to-right row sums ->
    bit-identical MatMult / MatMultAdd / fused dot; new values (hipxMatUpdateValues) rebuild the templates; values that
    make every row distinct fall back to the general packed kernels; 64-bit row offsets take the same path."""
    from petsc_amd import _lib
    rng = np.random.default_rng(5)
    ai, aj, aa = orc.stencil(kind, n, m=m)
    N = len(ai) - 1
    x = rng.standard_normal(N)
    for wide in (False, True):
        A = _lib.mat_create_csr(N, N, ai.astype(np.int64) if wide else ai, aj, aa)
        _lib.chk(hx.hipxMatSetSpMVVariant(A, 26))
        assert is_template_kernel(kernel_name(hx, A)), kernel_name(hx, A)
        X, Y, Y0 = _lib.DVec(N, x), _lib.DVec(N), _lib.DVec(N, x[::-1].copy())
        _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
        yr = orc.matmult(ai, aj, aa, x)
        assert np.array_equal(Y.get(), yr)
        _lib.chk(hx.hipxMatMultAdd(A, X.ptr, Y0.ptr, Y.ptr))
        zr = np.zeros(N)
        orc.lib().orc_MatMult_SeqAIJ_dispatch(N, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(x), orc.P(x[::-1].copy()), orc.P(zr), 0)
        assert np.array_equal(Y.get(), zr)
        dot = C.c_double()
        _lib.chk(hx.hipxMatMultDot(A, X.ptr, Y.ptr, C.byref(dot)))
        assert np.array_equal(Y.get(), yr) and abs(dot.value - float(x @ yr)) <= 1e-12 * np.abs(x * yr).sum()
        aa2 = aa * 0.75                                                   # other values, same templates' shape
        _lib.chk(hx.hipxMatUpdateValues(A, orc.P(aa2)))
        assert is_template_kernel(kernel_name(hx, A))
        _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
        assert np.array_equal(Y.get(), orc.matmult(ai, aj, aa2, x))
        if N > 300:
            aa3 = aa * (1.0 + 1e-3 * rng.standard_normal(aa.size))       # every row distinct: no templates
            _lib.chk(hx.hipxMatUpdateValues(A, orc.P(aa3)))
            assert not is_template_kernel(kernel_name(hx, A))
            _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
            assert np.array_equal(Y.get(), orc.matmult(ai, aj, aa3, x))
        for v in (X, Y, Y0):
            v.free()
        _lib.mat_destroy(A)


def test_long_rows_beyond_lds_tile(hx):
    """Rows with more nonzeros than the 2048-entry LDS tile take the block-wide path (tree sum): 1e-14 relative."""
    rng = np.random.default_rng(11)
    m, n = 40, 20000
    lens = np.array([3, 5000, 0, 2047, 2049, 7, 12000] + [4] * 33)
    ai = np.zeros(m + 1, np.int32)
    ai[1:] = np.cumsum(lens)
    aj = np.concatenate([np.sort(rng.choice(n, size=k, replace=False)) for k in lens]).astype(np.int32)
    aa = rng.standard_normal(ai[-1])
    x = rng.standard_normal(n)
    y = spmv_gpu(hx, ai, aj, aa, x, ncols=n)
    yr0 = orc.matmult(ai, aj, aa, x)
    mag0 = np.array([np.abs(aa[ai[r]:ai[r + 1]] * x[aj[ai[r]:ai[r + 1]]]).sum() for r in range(m)])
    for v in (22, 23):  # their tile may be larger than 2048 (4096 for long rows): rows that fit are exact, the rest tree-summed
        yv = spmv_gpu(hx, ai, aj, aa, x, ncols=n, variant=v)
        assert np.array_equal(yv[lens <= 2040], yr0[lens <= 2040])
        assert np.all(np.abs(yv - yr0) <= 1e-14 * (mag0 + 1e-300))
    aq = np.round(aa)  # few distinct values: dictionary kernels; rows beyond THEIR tile (2048 / 4096 / 8192) take the block-wide path
    yqr = orc.matmult(ai, aj, aq, x)
    magq = np.array([np.abs(aq[ai[r]:ai[r + 1]] * x[aj[ai[r]:ai[r + 1]]]).sum() for r in range(m)])
    for v in (24, 25):
        yq = spmv_gpu(hx, ai, aj, aq, x, ncols=n, variant=v)
        assert np.array_equal(yq[lens <= 2040], yqr[lens <= 2040])
        assert np.all(np.abs(yq - yqr) <= 1e-14 * (magq + 1e-300))
    yr = orc.matmult(ai, aj, aa, x)
    short = lens <= 2040
    assert np.array_equal(y[short], yr[short])
    mag = np.array([np.abs(aa[ai[r]:ai[r + 1]] * x[aj[ai[r]:ai[r + 1]]]).sum() for r in range(m)])
    assert np.all(np.abs(y - yr) <= 1e-14 * (mag + 1e-300))


def test_empty_matrix_and_empty_rows(hx):
    ai = np.zeros(1, np.int32)
    y = spmv_gpu(hx, ai, np.zeros(0, np.int32), np.zeros(0), np.zeros(0))
    assert y.size == 0
    ai = np.zeros(11, np.int32)
    y = spmv_gpu(hx, ai, np.zeros(0, np.int32), np.zeros(0), np.ones(10))
    assert np.array_equal(y, np.zeros(10))


def test_int64_row_offsets(hx):
    ai, aj, aa = orc.stencil("27pt", 12)
    x = xvec(len(ai) - 1)
    y = spmv_gpu(hx, ai.astype(np.int64), aj, aa, x)
    assert np.array_equal(y, orc.matmult(ai, aj, aa, x))


def test_get_diagonal_and_jacobi_setup(hx):
    from petsc_amd import _lib
    rng = np.random.default_rng(5)
    ai, aj, aa = orc.stencil("7pt", 10)
    aa = aa.copy()
    N = len(ai) - 1
    # knock out one diagonal value (zero pivot) -> PCJACOBI puts 1 there (jacobi.c:255-266)
    d0 = [k for k in range(ai[7], ai[8]) if aj[k] == 7][0]
    aa[d0] = 0.0
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    D = _lib.DVec(N)
    _lib.chk(hx.hipxMatGetDiagonal(A, D.ptr))
    dref = np.zeros(N)
    orc.lib().orc_MatGetDiagonal_SeqAIJ(N, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(dref))
    assert np.array_equal(D.get(), dref)
    _lib.chk(hx.hipxPCJacobiSetUp(A, D.ptr))
    jref = np.zeros(N)
    orc.lib().orc_PCSetUp_Jacobi(N, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(jref))
    assert np.array_equal(D.get(), jref) and jref[7] == 1.0
    # missing diagonal entry -> 0 (aij.c:1347-1380)
    ai2, aj2, aa2 = random_csr(50, 50, rng, 4)
    A2 = _lib.mat_create_csr(50, 50, ai2, aj2, aa2)
    D2 = _lib.DVec(50)
    _lib.chk(hx.hipxMatGetDiagonal(A2, D2.ptr))
    d2 = np.zeros(50)
    orc.lib().orc_MatGetDiagonal_SeqAIJ(50, orc.P(ai2), orc.P(aj2), orc.P(aa2), orc.P(d2))
    assert np.array_equal(D2.get(), d2)
    for v in (D, D2):
        v.free()
    _lib.mat_destroy(A)
    _lib.mat_destroy(A2)


def test_compressed_row_offdiag_block(hx):
    """MPIAIJ off-diagonal block: compressed rows, MatMult zeroes y, MatMultAdd in place (mpiaij.c:1059)."""
    from petsc_amd import _lib
    n, nr = 8, 3
    N = n ** 3
    ranges = np.zeros(nr + 1, np.int32)
    orc.lib().orc_PetscSplitOwnership(N, nr, orc.P(ranges))
    rs, re = int(ranges[1]), int(ranges[2])
    ai, aj, aa = orc.stencil("27pt", n, rs, re)
    m, nz = re - rs, len(aj)
    Ai, Aj, Bi, Bj, ga = (np.zeros(k, np.int32) for k in (m + 1, nz + 1, m + 1, nz + 1, nz + 1))
    Aa, Ba = np.zeros(nz + 1), np.zeros(nz + 1)
    ec = orc.lib().orc_MatSetUpMultiply_MPIAIJ(m, rs, re, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(Ai), orc.P(Aj), orc.P(Aa), orc.P(Bi), orc.P(Bj), orc.P(Ba), orc.P(ga))
    ci, ridx = np.zeros(m + 1, np.int32), np.zeros(m + 1, np.int32)
    nrc = orc.lib().orc_MatCheckCompressedRow(m, orc.P(Bi), orc.P(ci), orc.P(ridx))
    B = _lib.mat_create_cprow(m, ec, nrc, ci[:nrc + 1], ridx[:nrc], Bj[:Bi[m]], Ba[:Bi[m]])
    lv = xvec(ec)
    LV, Y = _lib.DVec(ec, lv), _lib.DVec(m, np.full(m, 7.0))
    _lib.chk(hx.hipxMatMult(B, LV.ptr, Y.ptr))
    yr = np.zeros(m)
    orc.lib().orc_MatMult_SeqAIJ_cprow(m, nrc, orc.P(ci), orc.P(ridx), orc.P(Bj), orc.P(Ba), orc.P(lv), orc.P(yr))
    assert np.array_equal(Y.get(), yr)
    y0 = np.random.default_rng(3).standard_normal(m)
    Y.set(y0)
    _lib.chk(hx.hipxMatMultAdd(B, LV.ptr, Y.ptr, Y.ptr))
    zr = np.zeros(m)
    orc.lib().orc_MatMultAdd_SeqAIJ(m, orc.P(Bi), orc.P(Bj), orc.P(Ba), orc.P(lv), orc.P(y0), orc.P(zr))
    assert np.array_equal(Y.get(), zr)
    LV.free()
    Y.free()
    _lib.mat_destroy(B)


def test_full_size_7pt_256_properties(hx):
    """BASELINE config 2 size (N = 16.7 M, nnz = 117 M): A*1 is known in closed form (6 - #neighbours), A is symmetric
    (x.Ay == y.Ax to rounding), and a strided sample of rows is compared with the oracle bit for bit."""
    from petsc_amd import _lib
    _, ks = _lib.load()
    n = 256
    N = n ** 3
    nz = ks.HipxAssemble_poisson7(n, 0, N, None, None, None)
    assert nz == 7 * N - 6 * n * n
    ai, aj, aa = np.zeros(N + 1, np.int32), np.zeros(nz, np.int32), np.zeros(nz)
    ks.HipxAssemble_poisson7(n, 0, N, ai.ctypes.data_as(C.c_void_p), aj.ctypes.data_as(C.c_void_p), aa.ctypes.data_as(C.c_void_p))
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    ones = np.ones(N)
    X, Y = _lib.DVec(N, ones), _lib.DVec(N)
    _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    y = Y.get()
    assert np.array_equal(y, 6.0 - (np.diff(ai) - 1))
    x = xvec(N)
    X.set(x)
    _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    y = Y.get()
    rows = np.arange(0, N, 4099)
    for r in rows[:2000]:
        s = 0.0
        for k in range(ai[r], ai[r + 1]):
            s += aa[k] * x[aj[k]]
        assert y[r] == s
    assert is_template_kernel(kernel_name(hx, A))
    for variant in (25, 23, 1):  # every kernel form gives the same 16.7 M doubles, bit for bit
        _lib.chk(hx.hipxMatSetSpMVVariant(A, variant))
        Yv = _lib.DVec(N)
        _lib.chk(hx.hipxMatMult(A, X.ptr, Yv.ptr))
        assert np.array_equal(Yv.get(), y), variant
        Yv.free()
    _lib.chk(hx.hipxMatSetSpMVVariant(A, 0))
    d1, d2 = C.c_double(), C.c_double()
    W = _lib.DVec(N, ones)
    _lib.chk(hx.hipxVecDot(W.ptr, Y.ptr, N, C.byref(d1)))  # 1 . (A x)
    _lib.chk(hx.hipxMatMult(A, W.ptr, Y.ptr))
    _lib.chk(hx.hipxVecDot(X.ptr, Y.ptr, N, C.byref(d2)))  # x . (A 1)
    assert abs(d1.value - d2.value) <= 1e-12 * abs(d1.value)
    dot = C.c_double()
    _lib.chk(hx.hipxMatMultDot(A, X.ptr, Y.ptr, C.byref(dot)))
    ref = C.c_double()
    _lib.chk(hx.hipxVecDot(X.ptr, Y.ptr, N, C.byref(ref)))
    assert abs(dot.value - ref.value) <= 1e-13 * abs(ref.value)
    for v in (X, Y, W):
        v.free()
    _lib.mat_destroy(A)


@pytest.mark.parametrize("nb,ragged", [(2000, False), (4099, True), (12000, True)])
def test_sell_triple_run_column_codes_for_three_unknowns_per_node(hx, nb, ragged):
    """SELL-64 with ONE column code per run of three consecutive columns (round 4: FEM matrices with three unknowns per node -- dense 3 x 3 blocks,
    MatMult_SeqBAIJ_3's operands): block rows of 14 or 20 ... 26 blocks (ragged: slices are padded with whole runs), all values distinct; y, y + A x and
    the fused dot bit-identical to the CSR loop; the same matrix with one row's run broken (a column shifted by one block) keeps the plain codes --
    same bits."""
    from petsc_amd import _lib
    rng = np.random.default_rng(41 + nb)
    nblk = rng.integers(20, 27, nb) if ragged else np.full(nb, 14)  # (mildly ragged: the SELL copy is only kept below 25 % padding)
    rows_i, cols, = [], []
    ai = [0]
    aj = []
    for b in range(nb):
        nbrs = np.unique(np.clip(b + rng.integers(-300, 300, nblk[b] * 2), 0, nb - 1))[:nblk[b]]
        cc = (3 * nbrs[:, None] + np.arange(3)[None, :]).ravel()
        for r in range(3):
            aj.append(cc)
            ai.append(ai[-1] + len(cc))
    ai = np.array(ai, np.int32)
    aj = np.concatenate(aj).astype(np.int32)
    N = 3 * nb
    aa = rng.standard_normal(len(aj))
    x, y0 = rng.standard_normal(N), rng.standard_normal(N)
    yr = orc.matmult(ai, aj, aa, x)
    zr = np.zeros(N)
    orc.lib().orc_MatMult_SeqAIJ_dispatch(N, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(x), orc.P(y0), orc.P(zr), 0)
    X, Y, Y0 = _lib.DVec(N, x), _lib.DVec(N), _lib.DVec(N, y0)
    for broken in (False, True):
        aj2 = aj.copy()
        if broken:  # row 7: its first run no longer consecutive (the columns stay sorted and distinct)
            k0 = ai[7]
            if aj2[k0] >= 3:
                aj2[k0] -= 3
            else:
                continue
            yr2 = orc.matmult(ai, aj2, aa, x)
        else:
            yr2 = yr
        A = _lib.mat_create_csr(N, N, ai, aj2, aa)
        _lib.chk(hx.hipxMatSetSpMVVariant(A, 28))
        assert kernel_name(hx, A).startswith("spmv_sell_kernel "), kernel_name(hx, A)
        Y.set(np.full(N, np.nan))
        _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
        assert np.array_equal(Y.get(), yr2), broken
        if not broken:
            _lib.chk(hx.hipxMatMultAdd(A, X.ptr, Y0.ptr, Y.ptr))
            assert np.array_equal(Y.get(), zr)
            d = C.c_double()
            _lib.chk(hx.hipxMatMultDot(A, X.ptr, Y.ptr, C.byref(d)))
            assert np.array_equal(Y.get(), yr) and abs(d.value - float(x @ yr)) <= 1e-12 * abs(float(np.abs(x) @ np.abs(yr)))
            aa2 = rng.standard_normal(len(aj))
            _lib.chk(hx.hipxMatUpdateValues(A, orc.P(aa2)))
            _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
            assert np.array_equal(Y.get(), orc.matmult(ai, aj, aa2, x))
        _lib.mat_destroy(A)
    for v in (X, Y, Y0):
        v.free()


def test_sell_copy_selection_padding_limit_fused_dot_and_value_update(hx):
    """variant 28 (hipx_sell.hip): the SELL-64 copy is used when its padding stays below the limit, with 16-bit
    window-coded columns when every slice fits 16 windows of 4096 columns (else 32-bit columns); very ragged matrices keep the CSR
    kernels; the fused x . (A x) and hipxMatUpdateValues go through the copy; -0.0 row sums keep their sign (padding is never
    multiplied)."""
    from petsc_amd import _lib
    rng = np.random.default_rng(11)
    # (1) FEM-like long rows, all values distinct: packed SELL
    n, per = 6000, 60
    ai = np.arange(0, (n + 1) * per, per, dtype=np.int32)
    aj = np.sort((np.arange(n)[:, None] + rng.integers(-400, 400, size=(n, per))) % n, axis=1).astype(np.int32)
    for r in range(n):  # distinct columns per row
        while len(np.unique(aj[r])) < per:
            aj[r] = np.sort(np.unique(np.concatenate([aj[r], rng.integers(0, n, per)]))[:per])
    aj = aj.ravel()
    aa = rng.standard_normal(ai[-1])
    x = rng.standard_normal(n)
    A = _lib.mat_create_csr(n, n, ai, aj, aa)
    _lib.chk(hx.hipxMatSetSpMVVariant(A, 28))
    assert kernel_name(hx, A).startswith("spmv_sell_kernel "), kernel_name(hx, A)
    X, Y = _lib.DVec(n, x), _lib.DVec(n)
    _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    yr = orc.matmult(ai, aj, aa, x)
    assert np.array_equal(Y.get(), yr)
    d = C.c_double()
    _lib.chk(hx.hipxMatMultDot(A, X.ptr, Y.ptr, C.byref(d)))  # fused dot: one partial per slice, folded in slice order
    assert np.array_equal(Y.get(), yr) and abs(d.value - float(x @ yr)) <= 1e-12 * abs(float(np.abs(x) @ np.abs(yr)))
    d2 = C.c_double()
    _lib.chk(hx.hipxMatMultDot(A, X.ptr, Y.ptr, C.byref(d2)))
    assert d2.value == d.value  # deterministic
    aa2 = rng.standard_normal(ai[-1])
    _lib.chk(hx.hipxMatUpdateValues(A, orc.P(aa2)))
    _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    assert np.array_equal(Y.get(), orc.matmult(ai, aj, aa2, x))
    _lib.mat_destroy(A)
    # (2) columns spread over the whole matrix: more than 16 windows per slice -> 32-bit columns, same bits
    n2 = 300000
    ai2, aj2, aa2 = random_csr(2000, n2, rng, 30, empty_frac=0.0)
    ai2[1:] = np.cumsum(np.full(2000, 30))  # random_csr may have produced shorter rows: rebuild exactly 30 per row
    aj2 = np.sort(rng.integers(0, n2, size=(2000, 30)), axis=1).astype(np.int32).ravel()
    aa2 = rng.standard_normal(2000 * 30)
    x2 = rng.standard_normal(n2)
    A2 = _lib.mat_create_csr(2000, n2, ai2, aj2, aa2)
    _lib.chk(hx.hipxMatSetSpMVVariant(A2, 28))
    assert kernel_name(hx, A2).startswith("spmv_sell_kernel ")
    X2, Y2 = _lib.DVec(n2, x2), _lib.DVec(2000)
    _lib.chk(hx.hipxMatMult(A2, X2.ptr, Y2.ptr))
    assert np.array_equal(Y2.get(), orc.matmult(ai2, aj2, aa2, x2))
    _lib.mat_destroy(A2)
    # (3) one long row among empty ones: padding far above the limit -> the CSR kernel stays
    ai3 = np.zeros(513, np.int32)
    ai3[7:] = 200
    aj3 = np.arange(200, dtype=np.int32)
    aa3 = rng.standard_normal(200)
    A3 = _lib.mat_create_csr(512, 512, ai3, aj3, aa3)
    _lib.chk(hx.hipxMatSetSpMVVariant(A3, 28))
    assert not kernel_name(hx, A3).startswith("spmv_sell_kernel")
    X3, Y3 = _lib.DVec(512, x[:512]), _lib.DVec(512)
    _lib.chk(hx.hipxMatMult(A3, X3.ptr, Y3.ptr))
    assert np.array_equal(Y3.get(), orc.matmult(ai3, aj3, aa3, x[:512]))
    _lib.mat_destroy(A3)
    # (4) signed zeros: a row of -0.0 products sums to -0.0 (no +0.0 padding term may enter the sum)
    ai4 = np.array([0, 2, 3] + [3 + 5 * k for k in range(1, 63)], np.int32)
    aj4 = np.concatenate([[0, 1], [2], np.tile(np.arange(5), 62)]).astype(np.int32)
    aa4 = np.concatenate([[-0.0, -0.0], [1.0], rng.standard_normal(5 * 62)])
    x4 = np.abs(rng.standard_normal(64)) + 1.0
    A4 = _lib.mat_create_csr(64, 64, ai4, aj4, aa4)
    _lib.chk(hx.hipxMatSetSpMVVariant(A4, 28))
    X4, Y4 = _lib.DVec(64, x4), _lib.DVec(64)
    _lib.chk(hx.hipxMatMult(A4, X4.ptr, Y4.ptr))
    y4, y4r = Y4.get(), orc.matmult(ai4, aj4, aa4, x4)
    assert np.array_equal(y4, y4r) and np.array_equal(np.signbit(y4), np.signbit(y4r))
    _lib.mat_destroy(A4)
    for v in (X, Y, X2, Y2, X3, Y3, X4, Y4):
        v.free()


@pytest.mark.parametrize("kind,n", [("7pt", 40), ("27pt", 24), ("5pt", 90)])
def test_pattern_templates_with_arbitrary_values(hx, kind, n):
    """variant 29: a variable-coefficient operator -- the stencil PATTERN with all-distinct values -- gets the
    pattern-template kernel (1-byte pattern id per row, values streamed): bit-identical y, MatMultAdd, the fused dot, and a value
    update on the same pattern; a matrix with more than 256 row patterns keeps the packed CSR kernel."""
    from petsc_amd import _lib
    rng = np.random.default_rng(5)
    ai, aj, aa = orc.stencil(kind, n)
    N = len(ai) - 1
    aa = aa * (1.0 + 0.3 * rng.standard_normal(aa.size))  # all distinct: no value dictionary, no (offset, value) templates
    x = rng.standard_normal(N)
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    _lib.chk(hx.hipxMatSetSpMVVariant(A, 29))
    assert kernel_name(hx, A).startswith("spmv_tp_kernel "), kernel_name(hx, A)
    X, Y, Y0 = _lib.DVec(N, x), _lib.DVec(N), _lib.DVec(N, x[::-1].copy())
    _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    yr = orc.matmult(ai, aj, aa, x)
    assert np.array_equal(Y.get(), yr)
    _lib.chk(hx.hipxMatMultAdd(A, X.ptr, Y0.ptr, Y.ptr))
    zr = np.zeros(N)
    orc.lib().orc_MatMult_SeqAIJ_dispatch(N, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(x), orc.P(x[::-1].copy()), orc.P(zr), 0)
    assert np.array_equal(Y.get(), zr)
    d = C.c_double()
    _lib.chk(hx.hipxMatMultDot(A, X.ptr, Y.ptr, C.byref(d)))
    assert np.array_equal(Y.get(), yr) and abs(d.value - float(x @ yr)) <= 1e-12 * float(np.abs(x) @ np.abs(yr))
    aa2 = aa * (1.0 + 0.1 * rng.standard_normal(aa.size))
    _lib.chk(hx.hipxMatUpdateValues(A, orc.P(aa2)))
    assert kernel_name(hx, A).startswith("spmv_tp_kernel ")
    _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    assert np.array_equal(Y.get(), orc.matmult(ai, aj, aa2, x))
    for v in (X, Y, Y0):
        v.free()
    _lib.mat_destroy(A)
    ai3, aj3, aa3 = random_csr(3000, 3000, rng, 12)
    A3 = _lib.mat_create_csr(3000, 3000, ai3, aj3, aa3)
    _lib.chk(hx.hipxMatSetSpMVVariant(A3, 29))
    assert not kernel_name(hx, A3).startswith("spmv_tp_kernel")
    X3, Y3 = _lib.DVec(3000, x[:3000] if N >= 3000 else rng.standard_normal(3000)), _lib.DVec(3000)
    _lib.chk(hx.hipxMatMult(A3, X3.ptr, Y3.ptr))
    assert np.array_equal(Y3.get(), orc.matmult(ai3, aj3, aa3, X3.get()))
    X3.free()
    Y3.free()
    _lib.mat_destroy(A3)


@pytest.mark.parametrize("kind,n,m", [("7pt", 16, None), ("7pt", 21, None), ("27pt", 14, None), ("27pt", 15, None), ("5pt", 64, 40), ("5pt", 33, 31), ("7pt", 40, None)])
def test_pair_form_of_the_template_kernel_bit_exact(hx, kind, n, m):
    """spmv_pair_kernel: stencil matrices whose rows are all subsets of the interior row -- two consecutive rows per thread, aligned
    16-byte loads of x at the even offsets, the +-1 entries from the neighbouring lanes.  Grids with even and odd line lengths (odd:
    the pairs straddle lines and the plane offsets are odd), row counts that are not multiples of the 512-row chunk (tail kernel),
    MatMult / MatMultAdd / fused dot, and vectors that are NOT 16-byte aligned (the general template kernel takes over): y
    bit-identical to MatMult_SeqAIJ every time.  A row-scaled copy (templates, but not subsets of one base row) keeps the general kernel."""
    from petsc_amd import _lib
    rng = np.random.default_rng(17)
    ai, aj, aa = orc.stencil(kind, n, m=m)
    N = len(ai) - 1
    x = rng.standard_normal(N)
    y0 = rng.standard_normal(N)
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    _lib.chk(hx.hipxMatSetSpMVVariant(A, 26))  # row templates whatever the size (auto keeps small matrices on the plain kernels)
    name = kernel_name(hx, A)
    if N >= 512 and n % 2 == 0:
        assert name.startswith("spmv_pair_kernel "), name  # (even line lengths: 5 / 5 / 9 pairs for the 5- / 7- / 27-point operators)
    else:
        assert is_template_kernel(name), name             # (odd line lengths: more pairs; beyond 16 the general template kernel)
    yr = orc.matmult(ai, aj, aa, x)
    zr = np.zeros(N)
    orc.lib().orc_MatMult_SeqAIJ_dispatch(N, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(x), orc.P(y0), orc.P(zr), 0)
    X, Y, Y0 = _lib.DVec(N + 2, np.concatenate([x, [0.0, 0.0]])), _lib.DVec(N + 2), _lib.DVec(N + 2, np.concatenate([y0, [0.0, 0.0]]))
    for _ in range(3):  # (the chunk queue's ticket counters run on from launch to launch)
        _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
        assert np.array_equal(Y.get()[:N], yr)
    _lib.chk(hx.hipxMatMultAdd(A, X.ptr, Y0.ptr, Y.ptr))
    assert np.array_equal(Y.get()[:N], zr)
    dot = C.c_double()
    _lib.chk(hx.hipxMatMultDot(A, X.ptr, Y.ptr, C.byref(dot)))
    assert np.array_equal(Y.get()[:N], yr) and abs(dot.value - float(x @ yr)) <= 1e-12 * np.abs(x * yr).sum()
    # vectors shifted by one double: not 16-byte aligned -> the general template kernel, same bits; then aligned again
    Xs, Ys = _lib.DVec(N + 2, np.concatenate([[0.0], x, [0.0]])), _lib.DVec(N + 2)
    _lib.chk(hx.hipxMatMult(A, Xs.offset(1), Ys.offset(1)))
    assert np.array_equal(Ys.get()[1:N + 1], yr)
    _lib.chk(hx.hipxMatMultDot(A, Xs.offset(1), Ys.offset(1), C.byref(dot)))
    assert np.array_equal(Ys.get()[1:N + 1], yr) and abs(dot.value - float(x @ yr)) <= 1e-12 * np.abs(x * yr).sum()
    _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    assert np.array_equal(Y.get()[:N], yr)
    _lib.mat_destroy(A)
    # rows scaled by one of three factors: still <= 256 row templates, but no common base row
    scale = np.repeat(np.array([1.0, 0.5, 2.0])[np.arange(N) % 3], np.diff(ai))
    A = _lib.mat_create_csr(N, N, ai, aj, aa * scale)
    _lib.chk(hx.hipxMatSetSpMVVariant(A, 26))
    assert kernel_name(hx, A).startswith("spmv_tmpl_kernel "), kernel_name(hx, A)
    _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    assert np.array_equal(Y.get()[:N], orc.matmult(ai, aj, aa * scale, x))
    _lib.mat_destroy(A)
    for v in (X, Y, Y0, Xs, Ys):
        v.free()


@pytest.mark.parametrize("kind,n,m", [("7pt", 24, None), ("7pt", 18, 11), ("27pt", 16, None), ("5pt", 50, None), ("7pt", 15, None)])
def test_pair_form_chebyshev_epilogue(hx, kind, n, m):
    """hipxMatMultChebyshev (SpMV + the Chebyshev recurrence in one kernel) against hipxMatMult + hipxVecChebyshevStep: bit for bit (same
    operations in the same order per element; the row sums are those of hipxMatMult); the four association orders of VecAXPBYPCZ_Seq,
    PCJACOBI and PCNONE; row counts that are not multiples of the 512-row chunk (tail kernel); a grid with odd lines (two kernels)."""
    from petsc_amd import _lib
    rng = np.random.default_rng(11)
    ai, aj, aa = orc.stencil(kind, n, m=m)
    N = len(ai) - 1
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    _lib.chk(hx.hipxMatSetSpMVVariant(A, 26))
    name = kernel_name(hx, A)
    assert is_template_kernel(name) and (n % 2 or name.startswith("spmv_pair_kernel ")), name  # (odd lines: pair form with more pairs, or two kernels)
    null = C.c_void_p()
    bvec, dinv, pk, po = rng.standard_normal(N), 1.0 / (1.0 + rng.random(N)), rng.standard_normal(N), rng.standard_normal(N)
    Bv, D, PK, PP, PNx, R, Y = _lib.DVec(N, bvec), _lib.DVec(N, dinv), _lib.DVec(N, pk), _lib.DVec(N, po), _lib.DVec(N), _lib.DVec(N), _lib.DVec(N)
    for al, be, ga in [(1.0, 0.0, 0.7), (0.8, 1.7, 1.0), (0.8, 1.7, 0.0), (0.9, 1.3, 0.6)]:
        for use_d in (1, 0):
            dp = D.ptr if use_d else null
            _lib.chk(hx.hipxMatMultChebyshev(A, PK.ptr, PNx.ptr, C.c_double(al), C.c_double(be), C.c_double(ga), PP.ptr, dp, Bv.ptr))
            got_p = PNx.get()
            _lib.chk(hx.hipxMatMult(A, PK.ptr, Y.ptr))
            _lib.chk(hx.hipxVecChebyshevStep(PNx.ptr, C.c_double(al), C.c_double(be), C.c_double(ga), PP.ptr, PK.ptr, dp, Bv.ptr, Y.ptr, R.ptr, N))
            assert np.array_equal(got_p, PNx.get())
            assert np.array_equal(R.get(), bvec - orc.matmult(ai, aj, aa, pk))
    for v in (Bv, D, PK, PP, PNx, R, Y):
        v.free()
    _lib.mat_destroy(A)


def _leading_block(ai, aj, aa, M):
    """Leading principal M x M block of a CSR matrix (rows near the cut lose their entries beyond it)."""
    import scipy.sparse as sp
    N = len(ai) - 1
    B = sp.csr_matrix((aa, aj, ai), shape=(N, N))[:M, :M].tocsr()
    B.sort_indices()
    return B.indptr.astype(ai.dtype), B.indices.astype(aj.dtype), B.data.copy()


@pytest.mark.parametrize("kind,n,m,cut", [("7pt", 32, None, 0), ("7pt", 40, None, 0), ("7pt_box", (34, 34, 9), None, 0), ("27pt", 34, None, 0), ("27pt", 36, None, 0), ("5pt", 1024, 40, 0),
                                          ("5pt", 1500, 24, 0), ("7pt", 40, None, 1600 * 17 + 334), ("27pt", 34, None, 1156 * 9 + 2), ("7pt", 40, None, 1600 * 17 + 333)])
def test_march_form_of_the_template_kernel_bit_exact(hx, kind, n, m, cut):
    """spmv_march_kernel (variant 30: whenever the base template has the three-plane shape): planes of 1024 ... 2.25 M rows, one and several tiles
    per plane (the last one partial), 7 / 27 / 5 entries, a row count that is not a multiple of the plane (leading block of a stencil
    matrix), MatMult / MatMultAdd / fused dot, then vectors that are not 16-byte aligned (another template kernel takes over) -- y
    bit-identical to MatMult_SeqAIJ every time, and the same bits as the pair form's."""
    from petsc_amd import _lib
    rng = np.random.default_rng(23)
    ai, aj, aa = orc.stencil(kind, n, m=m)
    if cut:
        ai, aj, aa = _leading_block(ai, aj, aa, cut)
    N = len(ai) - 1
    x, y0 = rng.standard_normal(N), rng.standard_normal(N)
    yr = orc.matmult(ai, aj, aa, x)
    zr = np.zeros(N)
    orc.lib().orc_MatMult_SeqAIJ_dispatch(N, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(x), orc.P(y0), orc.P(zr), 0)
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    _lib.chk(hx.hipxMatSetSpMVVariant(A, 30))
    name = kernel_name(hx, A)
    assert is_template_kernel(name) and name.startswith(("spmv_march_kernel ", "spmv_march2_kernel ")) == (N % 2 == 0), name  # (an odd row count keeps the other template kernels)
    X, Y, Y0 = _lib.DVec(N + 2, np.concatenate([x, [0.0, 0.0]])), _lib.DVec(N + 2), _lib.DVec(N + 2, np.concatenate([y0, [0.0, 0.0]]))
    for _ in range(2):
        Y.set(np.full(N + 2, np.nan))
        _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
        assert np.array_equal(Y.get()[:N], yr)
    _lib.chk(hx.hipxMatMultAdd(A, X.ptr, Y0.ptr, Y.ptr))
    assert np.array_equal(Y.get()[:N], zr)
    dot, dot2 = C.c_double(), C.c_double()
    Y.set(np.full(N + 2, np.nan))
    _lib.chk(hx.hipxMatMultDot(A, X.ptr, Y.ptr, C.byref(dot)))
    assert np.array_equal(Y.get()[:N], yr) and abs(dot.value - float(x @ yr)) <= 1e-12 * np.abs(x * yr).sum()
    _lib.chk(hx.hipxMatMultDot(A, X.ptr, Y.ptr, C.byref(dot2)))
    assert dot2.value == dot.value  # deterministic (static split)
    Xs, Ys = _lib.DVec(N + 2, np.concatenate([[0.0], x, [0.0]])), _lib.DVec(N + 2)   # x not 16-byte aligned: not the march form
    _lib.chk(hx.hipxMatMult(A, Xs.offset(1), Ys.offset(1)))
    assert np.array_equal(Ys.get()[1:N + 1], yr)
    _lib.chk(hx.hipxMatMultDot(A, Xs.offset(1), Ys.offset(1), C.byref(dot2)))
    assert np.array_equal(Ys.get()[1:N + 1], yr) and abs(dot2.value - dot.value) <= 1e-12 * np.abs(x * yr).sum()
    _lib.chk(hx.hipxMatMultDot(A, X.ptr, Y.ptr, C.byref(dot2)))
    assert dot2.value == dot.value
    _lib.chk(hx.hipxMatSetSpMVVariant(A, 26))  # the pair form (these sizes give too few workgroups for the march form): same bits
    assert not kernel_name(hx, A).startswith("spmv_march")
    _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    assert np.array_equal(Y.get()[:N], yr)
    for v in (X, Y, Y0, Xs, Ys):
        v.free()
    _lib.mat_destroy(A)


@pytest.mark.parametrize("kind,n,m", [("7pt", 32, None), ("7pt", 64, None), ("7pt", 96, None), ("7pt", 128, None), ("27pt", 64, None), ("27pt", 96, None), ("5pt", 1024, 48),
                                       ("5pt", 2048, 24), ("7pt", 192, None), ("7pt_box", (256, 64, 20), None),
                                       # lines of 768 and 1024 points (BASELINE config 5's planes): the 512-thread form, 4096-row tiles, one workgroup per CU
                                       ("7pt_box", (1024, 1024, 6), None), ("7pt_box", (768, 768, 4), None)])
def test_march2_kernel_bit_exact_and_same_bits_as_the_first_march_kernel(hx, kind, n, m):
    """spmv_march2_kernel (round 4: whole planes and tiles, plane-periodic template ids, run-addressed operands): planes of 1024 ... 36864 rows,
    1024- and 2048-row tiles, 5 / 7 / 27 entries, halos of 2 ... 256 elements: y bit-identical to MatMult_SeqAIJ, the fused dot deterministic
    and equal to the first march kernel's own (same partial layout), which the developer switch HIPX_MARCH1 still runs (subprocess)."""
    from petsc_amd import _lib
    rng = np.random.default_rng(29)
    ai, aj, aa = orc.stencil(kind, n, m=m)
    N = len(ai) - 1
    x = rng.standard_normal(N)
    yr = orc.matmult(ai, aj, aa, x)
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    _lib.chk(hx.hipxMatSetSpMVVariant(A, 30))
    name = kernel_name(hx, A)
    assert name.startswith("spmv_march2_kernel "), name
    X, Y = _lib.DVec(N, x), _lib.DVec(N)
    for _ in range(2):
        Y.set(np.full(N, np.nan))
        _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
        assert np.array_equal(Y.get(), yr)
    dot, dot2 = C.c_double(), C.c_double()
    Y.set(np.full(N, np.nan))
    _lib.chk(hx.hipxMatMultDot(A, X.ptr, Y.ptr, C.byref(dot)))
    assert np.array_equal(Y.get(), yr) and abs(dot.value - float(x @ yr)) <= 1e-12 * np.abs(x * yr).sum()
    _lib.chk(hx.hipxMatMultDot(A, X.ptr, Y.ptr, C.byref(dot2)))
    assert dot2.value == dot.value
    for v in (X, Y):
        v.free()
    _lib.mat_destroy(A)


@pytest.mark.parametrize("kind,n,m", [("7pt", 40, None), ("27pt", 36, None), ("7pt_box", (200, 200, 24), None), ("5pt", 1500, 24), ("27pt", (72, 60, 12), None), ("7pt_box", (50, 42, 30), None)])
def test_march2_planes_that_are_not_whole_tiles(hx, kind, n, m):
    """Round 5: planes whose row count is not a multiple of the tile (1600 = 1024 + 576, 1296 = 1024 + 272, 40000 = 19 x 2048 + 1088, lines of 1500
    points, 4320 = 4 x 1024 + 224, 2100 = 2 x 1024 + 52): spmv_march2_kernel takes the whole tiles, spmv_march2_rem_kernel the rest of every plane.
    MatMult bit-identical to MatMult_SeqAIJ; the fused dot deterministic; the CG prologue form (p_new, x, w and the dot) bit-identical to the separate
    kernels -- with the fold of the dot partials inside the product kernel (at these sizes one sum_kernel workgroup would do it, so the fold is the kernel's own)."""
    from petsc_amd import _lib
    from test_gpu_sor import box_csr
    rng = np.random.default_rng(37)
    if isinstance(n, tuple) and kind == "27pt":
        ai, aj, aa = box_csr(n[0], n[1], n[2], 27)
    else:
        ai, aj, aa = orc.stencil(kind, n, m=m)
    N = len(ai) - 1
    x = rng.standard_normal(N)
    yr = orc.matmult(ai, aj, aa, x)
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    _lib.chk(hx.hipxMatSetSpMVVariant(A, 30))
    name = kernel_name(hx, A)
    assert name.startswith("spmv_march2_kernel "), name
    X, Y = _lib.DVec(N, x), _lib.DVec(N)
    for _ in range(2):
        Y.set(np.full(N, np.nan))
        _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
        assert np.array_equal(Y.get(), yr)
    dot, dot2, dot3 = C.c_double(), C.c_double(), C.c_double()
    Y.set(np.full(N, np.nan))
    _lib.chk(hx.hipxMatMultDot(A, X.ptr, Y.ptr, C.byref(dot)))
    assert np.array_equal(Y.get(), yr) and abs(dot.value - float(x @ yr)) <= 1e-12 * np.abs(x * yr).sum()
    _lib.chk(hx.hipxMatMultDot(A, X.ptr, Y.ptr, C.byref(dot2)))
    assert dot2.value == dot.value
    Y.set(np.full(N, np.nan))
    _lib.chk(hx.hipxMatMultDotBegin(A, X.ptr, Y.ptr, 5, None))  # the fold inside the product kernel (when one sum_kernel workgroup would do it)
    _lib.chk(hx.hipxRedEnd(5, 1, C.byref(dot3)))
    assert np.array_equal(Y.get(), yr) and dot3.value == dot.value
    # the CG prologue form against the separate kernels
    if kind != "27pt":
        p0, r, x0 = rng.standard_normal(N), rng.standard_normal(N), rng.standard_normal(N)
        b, a, dconst = 0.731 / 1.913, 1.913 / 2.57, 0.37
        P, R, XS, W = _lib.DVec(N, p0), _lib.DVec(N, r), _lib.DVec(N, x0), _lib.DVec(N)
        _lib.chk(hx.hipxCGAypxAxpyR(P.ptr, b, R.ptr, dconst, XS.ptr, a, N))
        dref = C.c_double()
        _lib.chk(hx.hipxMatMultDot(A, P.ptr, W.ptr, C.byref(dref)))
        pr, xr, wr = P.get(), XS.get(), W.get()
        assert np.array_equal(wr, orc.matmult(ai, aj, aa, pr))
        P.set(p0)
        XS.set(x0)
        P2, W2 = _lib.DVec(N, np.full(N, np.nan)), _lib.DVec(N, np.full(N, np.nan))
        fused, d = C.c_int(0), C.c_double()
        _lib.chk(hx.hipxMatMultCGDirectionDotBegin(A, P.ptr, P2.ptr, R.ptr, dconst, XS.ptr, b, a, None, None, None, W2.ptr, 5, None, C.byref(fused)))
        assert fused.value == 1
        _lib.chk(hx.hipxRedEnd(5, 1, C.byref(d)))
        assert np.array_equal(P2.get(), pr) and np.array_equal(XS.get(), xr) and np.array_equal(W2.get(), wr)
        assert d.value == dref.value
        for v in (P, R, XS, W, P2, W2):
            v.free()
    for v in (X, Y):
        v.free()
    _lib.mat_destroy(A)


@pytest.mark.parametrize("n", [256, 200, 128])
def test_march2_in_kernel_fold_equals_the_separate_fold_under_load(hx, n):
    """the fold of the product's dot partials by spmv_march2_kernel's LAST workgroup hands the partials over with relaxed agent-scope atomic stores
    (sc1: written through) + s_waitcnt vmcnt(0) + a ticket, not with release/acquire fences (their L2 write-back costs 24 us per launch).  That is the
    documented gfx950 granule hand-off (MI355X_MICROARCH 'valid forms': sc1 stores AND sc1 loads both sides), not the HIP memory model -- so it is pinned here:
    60 back-to-back launches per size (the chip busy, L2 warm, every launch racing the one before), each p . A p compared BITWISE with the sum the separate
    fold kernel (hipxMatMultDot -> sum_kernel over the same partials) forms, on the headline size, on 200^3 (whole tiles + the remainder kernel's partials)
    and on 128^3 (fewer workgroups than CUs x 2)."""
    from petsc_amd import _lib
    _, ks = _lib.load()
    N = n ** 3
    nz = ks.HipxAssemble_poisson7(n, 0, N, None, None, None)
    ai, aj, aa = np.zeros(N + 1, np.int32), np.zeros(nz, np.int32), np.zeros(nz)
    ks.HipxAssemble_poisson7(n, 0, N, ai.ctypes.data_as(C.c_void_p), aj.ctypes.data_as(C.c_void_p), aa.ctypes.data_as(C.c_void_p))
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    del ai, aj, aa
    rng = np.random.default_rng(n)
    xs = [_lib.DVec(N, rng.standard_normal(N)) for _ in range(3)]
    Y = _lib.DVec(N)
    _lib.chk(hx.hipxMatMult(A, xs[0].ptr, Y.ptr))
    assert kernel_name(hx, A).startswith("spmv_march2_kernel "), kernel_name(hx, A)
    want = []
    for X in xs:  # the separate fold: partials -> sum_kernel
        d = C.c_double()
        _lib.chk(hx.hipxMatMultDot(A, X.ptr, Y.ptr, C.byref(d)))
        want.append(d.value)
    bad = 0
    for rep in range(20):
        for k, X in enumerate(xs):  # three different operands in turn: a stale partial of the launch before would show
            _lib.chk(hx.hipxMatMultDotBegin(A, X.ptr, Y.ptr, 5 + k, None))
        for k in range(3):
            d = C.c_double()
            _lib.chk(hx.hipxRedEnd(5 + k, 1, C.byref(d)))
            bad += d.value != want[k]
    assert bad == 0, "%d of 60 in-kernel folds differ from the separate fold" % bad
    for v in xs + [Y]:
        v.free()
    _lib.mat_destroy(A)


def test_march2_refuses_matrices_whose_template_ids_are_not_plane_periodic(hx):
    """One interior row of an interior plane loses an entry (its template differs from the same row of the other planes): the second-generation
    kernel's set-up check must see it and the first march kernel takes the matrix -- still bit-identical."""
    from petsc_amd import _lib
    ai, aj, aa = orc.stencil("7pt", 64)
    N = len(ai) - 1
    r = 20 * 4096 + 17 * 64 + 9
    k = ai[r] + 1  # drop the second entry of row r (keep CSR valid: shift the arrays)
    aj2, aa2 = np.delete(aj, k), np.delete(aa, k)
    ai2 = ai.copy()
    ai2[r + 1:] -= 1
    x = np.random.default_rng(3).standard_normal(N)
    yr = orc.matmult(ai2, aj2, aa2, x)
    A = _lib.mat_create_csr(N, N, ai2, aj2, aa2)
    _lib.chk(hx.hipxMatSetSpMVVariant(A, 30))
    name = kernel_name(hx, A)
    assert name.startswith("spmv_march_kernel "), name
    X, Y = _lib.DVec(N, x), _lib.DVec(N)
    _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    assert np.array_equal(Y.get(), yr)
    X.free()
    Y.free()
    _lib.mat_destroy(A)


@pytest.mark.parametrize("kind,n,m,dconst", [("7pt", 64, None, 1.0 / 6.0), ("7pt", 96, None, 1.0), ("7pt", 192, None, 0.37), ("5pt", 1024, 48, 0.25), ("7pt_box", (1024, 1024, 5), None, 1.0),
                                             ("27pt", 256, None, 1.0 / 26.0)])  # (the 27-entry kernels carry the prologue from 256-point lines on)
def test_cg_direction_update_as_the_products_prologue_bit_identical(hx, kind, n, m, dconst):
    """hipxMatMultCGDirectionDotBegin (p_new = r * dconst + b p, x += a p, w = A p_new, p_new . w in one kernel) against the separate kernels it
    replaces (hipxCGAypxAxpyR, hipxMatMultDot): p_new, x, w bit-identical, the dot the same double (same partials); host scalars and
    device-resident scalars (b = beta_new / beta_old, a = beta_old / dpi formed on the device)."""
    from petsc_amd import _lib
    rng = np.random.default_rng(31)
    ai, aj, aa = orc.stencil(kind, n, m=m)
    N = len(ai) - 1
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    _lib.chk(hx.hipxMatSetSpMVVariant(A, 30))
    assert kernel_name(hx, A).startswith("spmv_march2_kernel ")
    p0, r, x0 = rng.standard_normal(N), rng.standard_normal(N), rng.standard_normal(N)
    bn, bo, dpi = 0.731, 1.913, 2.57
    b, a = bn / bo, bo / dpi
    # the separate kernels
    P, R, X, W = _lib.DVec(N, p0), _lib.DVec(N, r), _lib.DVec(N, x0), _lib.DVec(N)
    _lib.chk(hx.hipxCGAypxAxpyR(P.ptr, b, R.ptr, dconst, X.ptr, a, N))
    dref = C.c_double()
    _lib.chk(hx.hipxMatMultDot(A, P.ptr, W.ptr, C.byref(dref)))
    pr, xr, wr = P.get(), X.get(), W.get()
    assert np.array_equal(pr, r * dconst + b * p0) and np.array_equal(xr, x0 + a * p0) and np.array_equal(wr, orc.matmult(ai, aj, aa, pr))
    scal = _lib.DVec(4, np.array([bn, bo, dpi, 0.0]))
    for dev in (False, True):
        P.set(p0)
        X.set(x0)
        P2, W2 = _lib.DVec(N, np.full(N, np.nan)), _lib.DVec(N, np.full(N, np.nan))
        fused, d = C.c_int(0), C.c_double()
        dn = C.c_void_p(scal.ptr.value) if dev else None
        do = C.c_void_p(scal.ptr.value + 8) if dev else None
        dd = C.c_void_p(scal.ptr.value + 16) if dev else None
        _lib.chk(hx.hipxMatMultCGDirectionDotBegin(A, P.ptr, P2.ptr, R.ptr, dconst, X.ptr, 0.0 if dev else b, 0.0 if dev else a, dn, do, dd, W2.ptr, 5, C.c_void_p(scal.ptr.value + 24), C.byref(fused)))
        assert fused.value == 1
        _lib.chk(hx.hipxRedEnd(5, 1, C.byref(d)))
        assert np.array_equal(P2.get(), pr) and np.array_equal(X.get(), xr) and np.array_equal(W2.get(), wr), dev
        assert np.array_equal(P.get(), p0)  # the old direction is left alone
        assert d.value == dref.value and scal.get()[3] == dref.value
        P2.free()
        W2.free()
    for v in (P, R, X, W, scal):
        v.free()
    _lib.mat_destroy(A)


@pytest.mark.parametrize("kind,n,rows", [("7pt", 40, 1600 * 20), ("27pt", 34, 1156 * 20)])
def test_rectangular_template_matrix_keeps_the_general_template_kernel(hx, kind, n, rows):
    """The first `rows` rows of a stencil matrix with ALL its columns (rows x n^3: what a row slab looks like before it is split into blocks):
    row templates apply, but the pair and march forms bound their loads of x by the row count -- they are for square matrices only.  The
    general template kernel takes it; y bit-identical to MatMult_SeqAIJ (the last rows reach columns beyond `rows`)."""
    from petsc_amd import _lib
    rng = np.random.default_rng(29)
    ai, aj, aa = orc.stencil(kind, n, rstart=0, rend=rows)
    N = n ** 3
    assert aj.max() >= rows
    x = rng.standard_normal(N)
    yr = np.zeros(rows)
    orc.lib().orc_MatMult_SeqAIJ(rows, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(x), orc.P(yr))
    A = _lib.mat_create_csr(rows, N, ai, aj, aa)
    for variant in (26, 30):
        _lib.chk(hx.hipxMatSetSpMVVariant(A, variant))
        assert kernel_name(hx, A).startswith("spmv_tmpl_kernel "), kernel_name(hx, A)
        X, Y = _lib.DVec(N, x), _lib.DVec(rows)
        _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
        assert np.array_equal(Y.get(), yr)
        X.free()
        Y.free()
    _lib.mat_destroy(A)
