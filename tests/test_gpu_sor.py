"""GPU parity of MatSOR (level-scheduled sweeps) against the oracle's MatSOR_SeqAIJ restatement (aij.c:1842-2007):
bit-exact x for every sweep type the reference implements on this path, and GMRES/CG + PCSOR histories."""
import ctypes as C

import numpy as np
import pytest

import oracle as orc
from test_gpu_ksp import compare, solve_gpu
from test_gpu_mat import random_csr

pytestmark = pytest.mark.gpu

FWD, BWD, SYM, LFWD, LBWD, LSYM, ZERO, UPPER = 1, 2, 3, 4, 8, 12, 16, 64


def sor_gpu(hx, ai, aj, aa, b, omega, flag, shift, its, lits, x0):
    from petsc_amd import _lib
    N = len(ai) - 1
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    B, X = _lib.DVec(N, b), _lib.DVec(N, x0)
    _lib.chk(hx.hipxMatSOR(A, B.ptr, omega, flag, shift, its, lits, X.ptr))
    x = X.get()
    B.free()
    X.free()
    _lib.mat_destroy(A)
    return x


def sor_cpu(ai, aj, aa, b, omega, flag, shift, its, lits, x0):
    x = np.array(x0, dtype=np.float64)
    orc.lib().orc_MatSOR_SeqAIJ(len(ai) - 1, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(b), C.c_double(omega), flag, C.c_double(shift), its, lits, orc.P(x))
    return x


@pytest.mark.parametrize("kind,n,m", [("5pt", 9, 7), ("7pt", 12, None), ("27pt", 9, None)])
@pytest.mark.parametrize("flag", [SYM | ZERO, LSYM | ZERO, FWD | ZERO, BWD | ZERO, SYM, FWD, BWD, UPPER])
@pytest.mark.parametrize("omega,shift,its,lits", [(1.0, 0.0, 1, 1), (1.3, 0.0, 2, 1), (0.8, 0.25, 1, 3)])
def test_sor_sweeps_bit_exact(hx, kind, n, m, flag, omega, shift, its, lits):
    ai, aj, aa = orc.stencil(kind, n, m=m)
    N = len(ai) - 1
    rng = np.random.default_rng(7)
    b = rng.standard_normal(N)
    x0 = rng.standard_normal(N)
    g = sor_gpu(hx, ai, aj, aa, b, omega, flag, shift, its, lits, x0)
    o = sor_cpu(ai, aj, aa, b, omega, flag, shift, its, lits, x0)
    assert np.array_equal(g, o), np.abs(g - o).max()


def test_sor_structurally_unsymmetric_matrix(hx):
    """Levels come from the symmetrised pattern, so old/new value usage stays sequential even when a_ij != 0 = a_ji."""
    rng = np.random.default_rng(3)
    m = 400
    ai, aj, aa = random_csr(m, m, rng, 6, empty_frac=0.0)
    # force a full, dominant diagonal
    rows = []
    for r in range(m):
        cols = set(aj[ai[r]:ai[r + 1]].tolist()) | {r}
        rows.append(sorted(cols))
    ai = np.zeros(m + 1, np.int32)
    ai[1:] = np.cumsum([len(c) for c in rows])
    aj = np.concatenate(rows).astype(np.int32)
    aa = rng.standard_normal(len(aj))
    for r in range(m):
        k = ai[r] + rows[r].index(r)
        aa[k] = 10.0 + rng.random()
    b, x0 = rng.standard_normal(m), rng.standard_normal(m)
    for flag in (SYM | ZERO, SYM, FWD, BWD):
        g = sor_gpu(hx, ai, aj, aa, b, 1.1, flag, 0.0, 2, 1, x0)
        o = sor_cpu(ai, aj, aa, b, 1.1, flag, 0.0, 2, 1, x0)
        assert np.array_equal(g, o)


def test_sor_errors_like_reference(hx):
    from petsc_amd import _lib
    ai, aj, aa = orc.stencil("7pt", 4)
    N = len(ai) - 1
    aa = aa.copy()
    d0 = [k for k in range(ai[5], ai[6]) if aj[k] == 5][0]
    aa[d0] = 0.0
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    B, X = _lib.DVec(N, np.ones(N)), _lib.DVec(N)
    assert hx.hipxMatSOR(A, B.ptr, 1.0, SYM | ZERO, 0.0, 1, 1, X.ptr) == 71  # zero pivot (aij.c:1820)
    assert hx.hipxMatSOR(A, B.ptr, 1.0, 128, 0.0, 1, 1, X.ptr) == 56  # SOR_APPLY_LOWER unsupported (aij.c:1886)
    B.free()
    X.free()
    _lib.mat_destroy(A)


def test_golden_ex2_suffix3_on_gpu(hx):
    """The reference's own golden (output/ex2_3.out): GMRES + symmetric SOR on the 8x7 5-point Laplacian."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_outputs.json")))["ex2_3"]
    m, n = g["m"], g["n"]
    ai, aj, aa = orc.stencil("5pt", n, m=m)
    u = np.ones(m * n)
    b = orc.matmult(ai, aj, aa, u)
    from petsc_amd import _lib
    _, ks = _lib.load()
    x, its, reason, hist = solve_gpu("gmres", ai, aj, aa, b, pc="sor", rtol=1e-2 / ((m + 1) * (n + 1)), refine=2, sor_flag=3)
    assert its == g["iterations"]
    assert [float("%g" % h) for h in hist] == g["history"]
    assert float("%g" % np.linalg.norm(x - u)) == g["error"]


@pytest.mark.parametrize("kind,n", [("7pt", 16), ("27pt", 12)])
def test_gmres_sor_and_cg_ssor_histories(hx, kind, n):
    ai, aj, aa = orc.stencil(kind, n)
    b = orc.matmult(ai, aj, aa, np.ones(len(ai) - 1))
    g = solve_gpu("gmres", ai, aj, aa, b, pc="sor", rtol=1e-8)
    o = orc.ksp_solve("gmres", ai, aj, aa, b, pc="sor", rtol=1e-8)
    compare(g, o, 1e-8)
    g = solve_gpu("cg", ai, aj, aa, b, pc="sor", rtol=1e-8)
    o = orc.ksp_solve("cg", ai, aj, aa, b, pc="sor", rtol=1e-8)
    compare(g, o, 1e-8)
