"""GPU parity of hipxMatSOR on matrices with INODES against the oracle's MatSOR_SeqAIJ_Inode restatement (inode.c:2494-3810; pinned
bit for bit against the reference's own MatSOR in tests/test_oracle.py / tests/golden/inode_sor.json): bit-exact x for every sweep
kind, node sizes 1-5, diagonal blocks that interchange rows; the node partition as MatSeqAIJCheckInode finds it; hipxMatSetInodes;
MatMult / MatMultAdd in MatMult_SeqAIJ_Inode's pairwise order (inode.c:356-760); KSPCG / KSPGMRES + PCSOR histories on the blocked
elasticity stand-in."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle as orc
from surrogates import flan_surrogate_spd, inode_matrix
from test_gpu_ksp import solve_gpu

pytestmark = pytest.mark.gpu

ZERO, EISENSTAT = 16, 32
GI = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "inode_sor.json")))


def sor_cpu(ai, aj, aa, b, flag, its, lits, x0, no_inode=0, omega=1.0):
    x = np.array(x0, dtype=np.float64)
    rc = orc.lib().orc_MatSOR_SeqAIJ_dispatch(len(ai) - 1, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(b), C.c_double(omega), flag, C.c_double(0.0), its, lits, orc.P(x), no_inode)
    assert rc == 0
    return x


def sor_gpu(hx, ai, aj, aa, b, flag, its, lits, x0, omega=1.0, set_nodes=None, calls=1):
    from petsc_amd import _lib
    N = len(ai) - 1
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    if set_nodes is not None:
        ns = np.ascontiguousarray(set_nodes, dtype=np.int32)
        _lib.chk(hx.hipxMatSetInodes(A, len(ns) - 1 if len(ns) else 0, ns.ctypes.data_as(C.c_void_p) if len(ns) else None))
    B, X = _lib.DVec(N, b), _lib.DVec(N, x0)
    for _ in range(calls):
        X.set(x0)
        _lib.chk(hx.hipxMatSOR(A, B.ptr, omega, flag, 0.0, its, lits, X.ptr))
    used, nc = C.c_int(-2), C.c_int32(-5)
    _lib.chk(hx.hipxMatGetSORMode(A, C.byref(used)))
    _lib.chk(hx.hipxMatGetInodes(A, C.byref(nc)))
    x = X.get()
    B.free()
    X.free()
    _lib.mat_destroy(A)
    return x, used.value, nc.value


CASES = [(ZERO | 1, 1, 1), (ZERO | 2, 1, 1), (ZERO | 3, 1, 1), (ZERO | 12, 1, 1), (3, 1, 1), (1, 1, 1), (2, 1, 1), (12, 2, 1), (ZERO | 3, 3, 1), (ZERO | 1, 2, 1), (ZERO | 2, 2, 1),
         (ZERO | 12, 1, 2), (EISENSTAT, 1, 1), (2, 2, 3)]


@pytest.mark.parametrize("flag,its,lits", CASES)
@pytest.mark.parametrize("nnodes,sizes", [(60, (1, 2, 3, 4, 5)), (900, (1, 2, 3, 4, 5)), (700, (3,)), (500, (2, 1)), (400, (4, 2))])
def test_inode_sor_bit_exact(hx, flag, its, lits, nnodes, sizes):
    """Every sweep kind of MatSOR_SeqAIJ_Inode, every node size (kernels instantiated for largest nodes of 2, 3, 4 and 5 rows)."""
    ai, aj, aa = inode_matrix(nnodes=nnodes, sizes=sizes, seed=11 + nnodes, long_run=len(sizes) == 5)
    N = len(ai) - 1
    rng = np.random.default_rng(5)
    b, x0 = rng.standard_normal(N), rng.standard_normal(N)
    g, mode, nc = sor_gpu(hx, ai, aj, aa, b, flag, its, lits, x0)
    ns = np.zeros(N + 1, np.int32)
    want = orc.lib().orc_MatSeqAIJCheckInode(N, orc.P(ai), orc.P(aj), 5, orc.P(ns))
    assert want > 0 and nc == want and mode == 3, (want, nc, mode)
    o = sor_cpu(ai, aj, aa, b, flag, its, lits, x0)
    assert np.array_equal(g, o), np.abs(g - o).max()


@pytest.mark.parametrize("key", ["flag28_its1_lits1", "flag28_its1_lits1_noinode", "flag3_its1_lits1", "flag32_its1_lits1", "flag2_its2_lits3"])
def test_inode_sor_equals_the_reference_golden(hx, key):
    """The reference's own MatSOR output on the golden matrix (tests/golden/inode_sor.json, ref_driver -dump_sor), directly."""
    ai, aj, aa = inode_matrix()
    N = len(ai) - 1
    t = key.split("_")
    flag, its, lits = int(t[0][4:]), int(t[1][3:]), int(t[2][4:])
    b = orc.matmult(ai, aj, aa, np.ones(N))
    noin = key.endswith("noinode")
    g, mode, nc = sor_gpu(hx, ai, aj, aa, b, flag, its, lits, 0.5 + (np.arange(N) % 7) / 7.0, set_nodes=[] if noin else None)
    assert (mode == 3) == (not noin) and (nc == 0) == noin
    ref = np.array([float.fromhex(v) for v in GI["sor"][key]])
    assert np.array_equal(g, ref), np.abs(g - ref).max()


def test_inodes_are_not_used_where_the_reference_does_not(hx):
    """omega != 1 takes the point routine (aij.c:1852); a stencil has no inodes; hipxMatSetInodes(A, 0) = -mat_no_inode; a caller's own
    partition (coarser than the automatic one would be: nodes of 1 and 2 rows inside nodes of 3) is honoured."""
    ai, aj, aa = inode_matrix(nnodes=300, sizes=(3,), seed=3)
    N = len(ai) - 1
    rng = np.random.default_rng(9)
    b, x0 = rng.standard_normal(N), rng.standard_normal(N)
    g, mode, nc = sor_gpu(hx, ai, aj, aa, b, ZERO | 12, 1, 1, x0, omega=1.2)
    assert mode != 3
    o = x0.copy()
    orc.lib().orc_MatSOR_SeqAIJ(N, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(b), C.c_double(1.2), ZERO | 12, C.c_double(0.0), 1, 1, orc.P(o))
    assert np.array_equal(g, o)
    g, mode, nc = sor_gpu(hx, ai, aj, aa, b, ZERO | 12, 1, 1, x0, set_nodes=[])
    assert mode != 3 and nc == 0
    assert np.array_equal(g, sor_cpu(ai, aj, aa, b, ZERO | 12, 1, 1, x0, no_inode=1))
    # a caller's partition: split every node of 3 rows into 1 + 2
    ns = np.zeros(N + 1, np.int32)
    want = orc.lib().orc_MatSeqAIJCheckInode(N, orc.P(ai), orc.P(aj), 5, orc.P(ns))
    fine = np.unique(np.concatenate([ns[:want + 1], ns[:want] + 1])).astype(np.int32)
    g, mode, nc = sor_gpu(hx, ai, aj, aa, b, ZERO | 12, 1, 1, x0, set_nodes=fine)
    assert mode == 3 and nc == len(fine) - 1
    o = x0.copy()
    rc = orc.lib().orc_MatSOR_SeqAIJ_Inode(N, orc.P(ai), orc.P(aj), orc.P(aa), len(fine) - 1, orc.P(fine), orc.P(b), C.c_double(1.0), ZERO | 12, C.c_double(0.0), 1, 1, orc.P(o))
    assert rc == 0 and np.array_equal(g, o)
    sa, sj, sv = orc.stencil("7pt", 10)
    g, mode, nc = sor_gpu(hx, sa, sj, sv, np.ones(1000), ZERO | 12, 1, 1, np.zeros(1000))
    assert nc == 0 and mode != 3


def test_inode_sor_follows_new_values(hx):
    """hipxMatUpdateValues invalidates the node-level copy and the block inverses."""
    from petsc_amd import _lib
    ai, aj, aa = inode_matrix(nnodes=200, seed=21)
    N = len(ai) - 1
    rng = np.random.default_rng(2)
    b = rng.standard_normal(N)
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    B, X = _lib.DVec(N, b), _lib.DVec(N)
    for scale in (1.0, 1.7):
        a2 = aa * scale
        _lib.chk(hx.hipxMatUpdateValues(A, a2.ctypes.data_as(C.c_void_p)))
        _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, ZERO | 12, 0.0, 1, 1, X.ptr))
        assert np.array_equal(X.get(), sor_cpu(ai, aj, a2, b, ZERO | 12, 1, 1, np.zeros(N)))
    B.free()
    X.free()
    _lib.mat_destroy(A)


@pytest.mark.parametrize("n", [8, 16])
def test_inode_sor_on_the_blocked_elasticity_pattern(hx, n):
    """The Flan_1565 stand-in's shape (3 unknowns per node, 81 entries per row, shuffled numbering), 1536 and 12288 rows."""
    ai, aj, aa = flan_surrogate_spd(n=n)
    N = len(ai) - 1
    rng = np.random.default_rng(4)
    b, x0 = rng.standard_normal(N), rng.standard_normal(N)
    for flag, its in ((ZERO | 12, 1), (3, 2)):
        g, mode, nc = sor_gpu(hx, ai, aj, aa, b, flag, its, 1, x0, calls=2)
        assert mode == 3 and nc == N // 3
        assert np.array_equal(g, sor_cpu(ai, aj, aa, b, flag, its, 1, x0))


@pytest.mark.parametrize("key", sorted(GI["ksp"]))
def test_ksp_sor_on_the_blocked_operator(hx, key):
    """KSPCG / KSPGMRES + PCSOR on the n = 8 stand-in against the REFERENCE's history with exact BLAS reductions (golden), with the
    device reductions in exact mode: every kernel of the iteration bit-identical (the product in MatMult_SeqAIJ_Inode's pairwise order) and the
    reductions exactly rounded on both sides -> the CG histories are EQUAL; GMRES (its Gram-Schmidt sums are dgemv calls on the reference's
    side) within 1e-12."""
    from petsc_amd import _lib
    ai, aj, aa = flan_surrogate_spd(n=8)
    N = len(ai) - 1
    L = orc.lib()
    b = np.zeros(N)
    noin = key.endswith("noinode")
    (L.orc_MatMult_SeqAIJ if noin else L.orc_MatMult_SeqAIJ_Inode)(N, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(np.ones(N)), orc.P(b))
    old = C.c_int(0)
    _lib.chk(hx.hipxGetReductionMode(C.byref(old)))
    _lib.chk(hx.hipxSetReductionMode(1))
    if noin:
        os.environ["HIPX_MAT_NO_INODE"] = "1"
    try:
        x, its, reason, h = solve_gpu("gmres" if key.startswith("gmres") else "cg", ai, aj, aa, b, pc="sor", rtol=1e-50, max_it=12)
    finally:
        os.environ.pop("HIPX_MAT_NO_INODE", None)
        _lib.chk(hx.hipxSetReductionMode(old.value))
    ref = np.array([float.fromhex(v) for v in GI["ksp"][key]["history_hex"]])
    assert len(h) == len(ref)
    rel = np.abs(h - ref) / np.abs(ref)
    if key.startswith("cg"):
        assert np.array_equal(h, ref), rel
    assert rel.max() < 1e-12, rel


@pytest.mark.parametrize("which", ["inode60", "inode900", "flan8", "flan16", "empty_rows"])
def test_matmult_of_a_matrix_with_inodes_sums_in_pairs(hx, which):
    """MatMult_SeqAIJ_Inode / MatMultAdd_SeqAIJ_Inode (aij.c:1459, 1617): every row's terms enter in pairs.  y, y0 + A x and the fused
    x . (A x) against the oracle's restatement, on the SELL-64 copy (long rows) and on the plain one-lane-per-row kernel (where the copy
    does not apply); declared free of inodes the same matrix gives MatMult_SeqAIJ's bits again."""
    from petsc_amd import _lib
    rng = np.random.default_rng(17)
    if which.startswith("inode"):
        ai, aj, aa = inode_matrix(nnodes=int(which[5:]), seed=5)
    elif which.startswith("flan"):
        ai, aj, aa = flan_surrogate_spd(n=int(which[4:]))
    else:  # runs of EMPTY rows are nodes too (inode.c:3946-3953: equal lengths, nothing to compare): the rows between them get the pairwise sums
        m = 600
        lens = np.where(np.arange(m) % 4 == 0, 37, 0)
        ai = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        aj = np.concatenate([np.sort(rng.choice(m, 37, replace=False)) for _ in range(int((lens > 0).sum()))]).astype(np.int32)
        aa = rng.standard_normal(len(aj))
    N = len(ai) - 1
    aa = aa * (1.0 + 1e-3 * rng.standard_normal(len(aa)))  # (values no longer multiples of 2^-10: the two orders round differently)
    x, y0 = rng.standard_normal(N), rng.standard_normal(N)
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    X, Y, Y0 = _lib.DVec(N, x), _lib.DVec(N), _lib.DVec(N, y0)
    kn = C.create_string_buffer(256)
    _lib.chk(hx.hipxMatGetSpMVKernel(A, kn, 256))
    assert b"inodes" in kn.value, kn.value
    yr = orc.matmult_ref(ai, aj, aa, x)
    _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    assert np.array_equal(Y.get(), yr), kn.value
    _lib.chk(hx.hipxMatMultAdd(A, X.ptr, Y0.ptr, Y.ptr))
    assert np.array_equal(Y.get(), orc.matmult_ref(ai, aj, aa, x, yadd=y0))
    d = C.c_double()
    _lib.chk(hx.hipxMatMultDot(A, X.ptr, Y.ptr, C.byref(d)))
    assert np.array_equal(Y.get(), yr) and abs(d.value - float(x @ yr)) <= 1e-12 * float(np.abs(x) @ np.abs(yr))
    plain = orc.matmult_ref(ai, aj, aa, x, no_inode=True)
    assert not np.array_equal(plain, yr)
    _lib.chk(hx.hipxMatSetInodes(A, 0, None))
    _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    assert np.array_equal(Y.get(), plain)
    for v in (X, Y, Y0):
        v.free()
    _lib.mat_destroy(A)


@pytest.mark.parametrize("coop", ["0", "1"])
def test_both_forms_of_the_node_sweep(hx, coop):
    """HIPX_SOR_INODE_COOP: 1 = 16 lanes per node (what narrow levels get), 0 = one lane per node (wide levels): the same bits."""
    ai, aj, aa = inode_matrix(nnodes=700, seed=8)
    N = len(ai) - 1
    rng = np.random.default_rng(6)
    b, x0 = rng.standard_normal(N), rng.standard_normal(N)
    os.environ["HIPX_SOR_INODE_COOP"] = coop
    try:
        for flag, its in ((ZERO | 12, 1), (3, 2), (2, 2), (EISENSTAT, 1)):
            g, mode, nc = sor_gpu(hx, ai, aj, aa, b, flag, its, 1, x0)
            assert mode == 3 and np.array_equal(g, sor_cpu(ai, aj, aa, b, flag, its, 1, x0)), (flag, its)
    finally:
        os.environ.pop("HIPX_SOR_INODE_COOP", None)


def test_a_wrong_partition_is_refused(hx):
    """hipxMatSetInodes with rows that do NOT share a column list: the relaxation refuses it (error 73, as PETSC_ERR_ARG_WRONGSTATE) instead of reading past rows."""
    from petsc_amd import _lib
    ai, aj, aa = orc.stencil("7pt", 6)
    N = len(ai) - 1
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    ns = np.arange(0, N + 1, 2, dtype=np.int32)  # pairs of stencil rows: different columns
    _lib.chk(hx.hipxMatSetInodes(A, len(ns) - 1, ns.ctypes.data_as(C.c_void_p)))
    B, X = _lib.DVec(N, np.ones(N)), _lib.DVec(N)
    rc = hx.hipxMatSOR(A, B.ptr, 1.0, ZERO | 12, 0.0, 1, 1, X.ptr)
    assert rc == 73, rc
    _lib.chk(hx.hipxMatSetInodes(A, 0, None))
    _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, ZERO | 12, 0.0, 1, 1, X.ptr))
    B.free()
    X.free()
    _lib.mat_destroy(A)
