"""GPU parity of MatSOR against the oracle's MatSOR_SeqAIJ restatement (aij.c:1842-2007): bit-exact x for every sweep type
the reference implements on this path, in every schedule libhipx has (strands for stencil matrices, the level-ordered
dependency-driven sweep, one launch per level), up to BASELINE sizes (>= 1 M rows), and GMRES/CG + PCSOR histories."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as orc
from test_gpu_ksp import compare, exact_solve, solve_gpu
from test_gpu_mat import random_csr

pytestmark = pytest.mark.gpu

FWD, BWD, SYM, LFWD, LBWD, LSYM, ZERO, EISENSTAT, UPPER = 1, 2, 3, 4, 8, 12, 16, 32, 64


MODES = {"levels": 0, "dep": 1, "strand": 2, "box": 4}


class sor_mode:
    """HIPX_SOR_MODE for the calls inside the block (read by hipxMatSOR at every call); None = the library's own choice."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.old = os.environ.pop("HIPX_SOR_MODE", None)
        if self.mode:
            os.environ["HIPX_SOR_MODE"] = self.mode

    def __exit__(self, *a):
        os.environ.pop("HIPX_SOR_MODE", None)
        if self.old is not None:
            os.environ["HIPX_SOR_MODE"] = self.old


def sor_gpu(hx, ai, aj, aa, b, omega, flag, shift, its, lits, x0, mode=None, want_mode=None):
    from petsc_amd import _lib
    N = len(ai) - 1
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    B, X = _lib.DVec(N, b), _lib.DVec(N, x0)
    with sor_mode(mode):
        _lib.chk(hx.hipxMatSOR(A, B.ptr, omega, flag, shift, its, lits, X.ptr))
    used = C.c_int(-2)
    _lib.chk(hx.hipxMatGetSORMode(A, C.byref(used)))
    if mode and flag != UPPER:
        assert used.value == MODES[mode], (used.value, mode)
    if want_mode is not None and flag != UPPER:
        assert used.value == MODES[want_mode], (used.value, want_mode)
    x = X.get()
    B.free()
    X.free()
    _lib.mat_destroy(A)
    return x


def sor_cpu(ai, aj, aa, b, omega, flag, shift, its, lits, x0):
    x = np.array(x0, dtype=np.float64)
    # (as the reference dispatches it, aij.c:1852: the point routine for every matrix of this file -- none has inodes; tests/test_gpu_inode.py)
    orc.lib().orc_MatSOR_SeqAIJ_dispatch(len(ai) - 1, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(b), C.c_double(omega), flag, C.c_double(shift), its, lits, orc.P(x), 0)
    return x


@pytest.mark.parametrize("kind,n,m", [("5pt", 9, 7), ("7pt", 12, None), ("27pt", 9, None), ("7pt", 16, None), ("27pt", 24, None), ("5pt", 64, 40)])
@pytest.mark.parametrize("flag", [SYM | ZERO, LSYM | ZERO, FWD | ZERO, BWD | ZERO, SYM, FWD, BWD, UPPER])
@pytest.mark.parametrize("omega,shift,its,lits", [(1.0, 0.0, 1, 1), (1.3, 0.0, 2, 1), (0.8, 0.25, 1, 3)])
@pytest.mark.parametrize("mode", ["strand", "dep", "levels"])
def test_sor_sweeps_bit_exact(hx, kind, n, m, flag, omega, shift, its, lits, mode):
    """Every sweep kind of aij.c:1930-2002 in every schedule; grids with L % 8 == 0 (16, 24, 64: the loader's aligned 64-byte
    runs) and without (9, 12), strands shorter than a panel and several panels per plane."""
    if mode != "strand" and (n >= 24 or (flag, omega) != (SYM | ZERO, 1.0)) and n > 12:
        pytest.skip("the level-ordered schedules are covered on the small grids")
    ai, aj, aa = orc.stencil(kind, n, m=m)
    N = len(ai) - 1
    rng = np.random.default_rng(7)
    b = rng.standard_normal(N)
    x0 = rng.standard_normal(N)
    g = sor_gpu(hx, ai, aj, aa, b, omega, flag, shift, its, lits, x0, mode=mode)
    o = sor_cpu(ai, aj, aa, b, omega, flag, shift, its, lits, x0)
    assert np.array_equal(g, o), np.abs(g - o).max()


def assemble_c(stencil, n):
    from petsc_amd import _lib
    _, ks = _lib.load()
    f = {7: ks.HipxAssemble_poisson7, 27: ks.HipxAssemble_bench27}[stencil]
    N = n ** 3
    nz = f(n, 0, N, None, None, None)
    ai, aj, aa = np.zeros(N + 1, np.int32), np.zeros(nz, np.int32), np.zeros(nz)
    f(n, 0, N, ai.ctypes.data_as(C.c_void_p), aj.ctypes.data_as(C.c_void_p), aa.ctypes.data_as(C.c_void_p))
    return ai, aj, aa


@pytest.mark.parametrize("stencil,n", [(7, 128), (27, 96), (27, 128), (7, 200)])
@pytest.mark.parametrize("mode", ["strand", "dep"])
def test_sor_bit_exact_at_scale(hx, stencil, n, mode):
    """>= 1 M rows (7-pt 128^3 = 2.1 M, 27-pt 96^3 = 0.88 M, 27-pt 128^3: the size class where the first dependency-driven
    sweep gave up during round 1; 7-pt 200^3 = 8 M rows = several rounds of panels per CU): PCSOR's default symmetric sweep and
    a general 2-iteration sweep, bit-identical to the sequential CPU sweep, in both dependency-driven schedules."""
    if mode == "dep" and n > 128:
        pytest.skip("level-ordered schedule: covered at 128^3")
    ai, aj, aa = assemble_c(stencil, n)
    N = len(ai) - 1
    rng = np.random.default_rng(11)
    b = rng.standard_normal(N)
    x0 = rng.standard_normal(N)
    for flag, omega, its in [(LSYM | ZERO, 1.0, 1), (SYM, 1.2, 2)]:
        g = sor_gpu(hx, ai, aj, aa, b, omega, flag, 0.0, its, 1, x0, mode=mode)
        o = sor_cpu(ai, aj, aa, b, omega, flag, 0.0, its, 1, x0)
        assert np.array_equal(g, o), (flag, np.abs(g - o).max())


def test_sor_config3_rank_slab_bit_exact(hx):
    """BASELINE config 3's per-rank operator: the diagonal block of rank 3 of 8 of the 27-pt 512^3 matrix (512 x 512 x 64 =
    16.8 M rows, 447 M nonzeros, split as MatSetUpMultiply_MPIAIJ does) -- PCSOR's local symmetric sweep, bit-identical to the
    sequential CPU sweep.  This is the sweep GMRES(30)+PCSOR runs on every rank (mpiaij.c:1408-1412)."""
    import sys
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from petsc_amd import _lib
    from petsc_amd import dist as pdist
    _, ks = _lib.load()
    n, nranks, rank = 512, 8, 3
    ranges = pdist.split_ownership(n ** 3, nranks)
    rs, re = int(ranges[rank]), int(ranges[rank + 1])
    ai, aj, aa = bench.assemble(ks, 27, (n, n, n), rs, re)
    fake = types.SimpleNamespace(all_gather_object=lambda out, obj, group=None: out.__setitem__(slice(None), [obj] * len(out)))
    plan = pdist.build_plan(ai, aj, aa, ranges, rank, dist=fake)
    del ai, aj, aa
    Ai, Aj, Aa = plan["Ai"], plan["Aj"], plan["Aa"]
    m = plan["m"]
    assert m == 512 * 512 * 64
    rng = np.random.default_rng(3)
    b = rng.standard_normal(m)
    g = sor_gpu(hx, Ai, Aj, Aa, b, 1.0, LSYM | ZERO, 0.0, 1, 1, np.zeros(m), want_mode="box")  # (round 5: the library's choice on this operator is the plane march)
    o = sor_cpu(Ai, Aj, Aa, b, 1.0, LSYM | ZERO, 0.0, 1, 1, np.zeros(m))
    assert np.array_equal(g, o), np.abs(g - o).max()
    g = sor_gpu(hx, Ai, Aj, Aa, b, 1.0, LSYM | ZERO, 0.0, 1, 1, np.zeros(m), mode="strand")
    assert np.array_equal(g, o), np.abs(g - o).max()


@pytest.mark.parametrize("kind,n,m", [("5pt", 9, 7), ("7pt", 12, None), ("27pt", 16, None), ("27pt", 9, None)])
@pytest.mark.parametrize("omega,shift", [(1.0, 0.0), (1.4, 0.0), (0.7, 0.3)])
@pytest.mark.parametrize("mode", ["strand", "dep", "levels"])
def test_sor_eisenstat_bit_exact(hx, kind, n, m, omega, shift, mode):
    """SOR_EISENSTAT (aij.c:1887-1929, what PCEISENSTAT applies): (L + E)^-1 A (U + E)^-1 b by Eisenstat's trick = a backward
    sweep, a diagonal update and a forward sweep; bit-identical to the reference loops in every schedule."""
    ai, aj, aa = orc.stencil(kind, n, m=m)
    N = len(ai) - 1
    rng = np.random.default_rng(21)
    b, x0 = rng.standard_normal(N), rng.standard_normal(N)
    g = sor_gpu(hx, ai, aj, aa, b, omega, EISENSTAT, shift, 1, 1, x0, mode=mode)
    o = sor_cpu(ai, aj, aa, b, omega, EISENSTAT, shift, 1, 1, x0)
    assert np.array_equal(g, o), np.abs(g - o).max()


def perturbed(aa, seed=7):
    """arbitrary values on the same pattern: every entry its own value, the diagonal kept dominant"""
    rng = np.random.default_rng(seed)
    return np.ascontiguousarray(aa * (1.0 + 0.3 * rng.random(aa.size)))


@pytest.mark.parametrize("kind,n,m", [("5pt", 9, 7), ("7pt", 12, None), ("27pt", 9, None), ("7pt", 16, None), ("27pt", 24, None), ("5pt", 64, 40), ("27pt", 16, None)])
@pytest.mark.parametrize("flag", [SYM | ZERO, LSYM | ZERO, FWD | ZERO, BWD | ZERO, EISENSTAT])
@pytest.mark.parametrize("omega,shift", [(1.0, 0.0), (1.3, 0.0), (0.8, 0.25)])
def test_sor_variable_coefficients_strand_bit_exact(hx, kind, n, m, flag, omega, shift):
    """Arbitrary values on a stencil pattern (no row templates: every row has its own coefficients): the strand schedule from the
    PATTERN templates with the coefficients streamed per row -- bit-identical to MatSOR_SeqAIJ for the sweeps PCSOR applies by
    default (zero initial guess, one iteration: aij.c:1930-1960) and for Eisenstat (aij.c:1887-1929)."""
    ai, aj, aa = orc.stencil(kind, n, m=m)
    aa = perturbed(aa)
    N = len(ai) - 1
    rng = np.random.default_rng(N + flag)
    b, x0 = rng.standard_normal(N), rng.standard_normal(N)
    g = sor_gpu(hx, ai, aj, aa, b, omega, flag, shift, 1, 1, x0, want_mode="strand")
    o = sor_cpu(ai, aj, aa, b, omega, flag, shift, 1, 1, x0)
    assert np.array_equal(g, o), np.abs(g - o).max()


@pytest.mark.parametrize("stencil,n", [(7, 128), (27, 96), (27, 128), (7, 100)])
def test_sor_variable_coefficients_at_scale(hx, stencil, n):
    """>= 1 M rows with arbitrary values (7-pt 100^3: strands of a length that is not a multiple of 8 -> the kernels for any
    alignment), the default symmetric sweep; then new VALUES through hipxMatUpdateValues: the coefficient streams follow."""
    from petsc_amd import _lib
    ai, aj, aa = assemble_c(stencil, n)
    aa = perturbed(aa)
    N = len(ai) - 1
    rng = np.random.default_rng(13)
    b = rng.standard_normal(N)
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    B, X = _lib.DVec(N, b), _lib.DVec(N)
    used = C.c_int()
    for vals in (aa, perturbed(aa, 9)):
        _lib.chk(hx.hipxMatUpdateValues(A, orc.P(vals)))
        _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, LSYM | ZERO, 0.0, 1, 1, X.ptr))
        _lib.chk(hx.hipxMatGetSORMode(A, C.byref(used)))
        assert used.value == MODES["strand"]
        o = sor_cpu(ai, aj, vals, b, 1.0, LSYM | ZERO, 0.0, 1, 1, np.zeros(N))
        g = X.get()
        assert np.array_equal(g, o), np.abs(g - o).max()
    B.free()
    X.free()
    _lib.mat_destroy(A)


def test_sor_default_schedule_selection(hx):
    """The library's own choice: the plane march for the zero-guess sweeps of constant-coefficient boxes (round 5), strands for the other sweeps
    of stencil matrices -- from the row templates, or from the pattern templates with
    streamed coefficients when the values are arbitrary (then only for the sweeps without old-value lists; the level-ordered
    sweep otherwise) -- and after hipxMatUpdateValues the choice follows the new values."""
    from petsc_amd import _lib
    ai, aj, aa = orc.stencil("7pt", 20)
    N = len(ai) - 1
    rng = np.random.default_rng(2)
    b, x0 = rng.standard_normal(N), rng.standard_normal(N)
    g = sor_gpu(hx, ai, aj, aa, b, 1.0, LSYM | ZERO, 0.0, 1, 1, x0, want_mode="box")  # (constant coefficients, zero-guess sweep: the plane march)
    assert np.array_equal(g, sor_cpu(ai, aj, aa, b, 1.0, LSYM | ZERO, 0.0, 1, 1, x0))
    g = sor_gpu(hx, ai, aj, aa, b, 1.2, SYM, 0.0, 2, 1, x0, want_mode="strand")  # (sweeps with old values: the strands)
    assert np.array_equal(g, sor_cpu(ai, aj, aa, b, 1.2, SYM, 0.0, 2, 1, x0))
    aav = aa * (1.0 + 0.01 * rng.standard_normal(aa.size))
    g = sor_gpu(hx, ai, aj, aav, b, 1.0, LSYM | ZERO, 0.0, 1, 1, x0, want_mode="strand")  # (variable coefficients: streamed per row)
    assert np.array_equal(g, sor_cpu(ai, aj, aav, b, 1.0, LSYM | ZERO, 0.0, 1, 1, x0))
    g = sor_gpu(hx, ai, aj, aav, b, 1.2, SYM, 0.0, 2, 1, x0, want_mode="dep")  # (sweeps with old-value lists: level-ordered there)
    assert np.array_equal(g, sor_cpu(ai, aj, aav, b, 1.2, SYM, 0.0, 2, 1, x0))
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    B, X = _lib.DVec(N, b), _lib.DVec(N, x0)
    used = C.c_int()
    for vals, want in [(aa, 4), (aav, 2), (aa * 2.0, 4)]:
        _lib.chk(hx.hipxMatUpdateValues(A, orc.P(np.ascontiguousarray(vals))))
        _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, LSYM | ZERO, 0.0, 1, 1, X.ptr))
        _lib.chk(hx.hipxMatGetSORMode(A, C.byref(used)))
        assert used.value == want
        assert np.array_equal(X.get(), sor_cpu(ai, aj, np.ascontiguousarray(vals), b, 1.0, LSYM | ZERO, 0.0, 1, 1, x0))
    B.free()
    X.free()
    _lib.mat_destroy(A)


def test_sor_long_dependency_chain_uses_level_launches(hx):
    """Tridiagonal matrix: one row per level.  Padding every level to a wave would need 64 slots per row; the library runs one
    launch per level there (and still matches the CPU sweep bit for bit)."""
    m = 3000
    ai = np.zeros(m + 1, np.int32)
    cols, vals = [], []
    rng = np.random.default_rng(4)
    for r in range(m):
        for c in (r - 1, r, r + 1):
            if 0 <= c < m:
                cols.append(c)
                vals.append(4.0 + rng.random() if c == r else -1.0 - rng.random())
        ai[r + 1] = len(cols)
    aj, aa = np.array(cols, np.int32), np.array(vals)
    b, x0 = rng.standard_normal(m), rng.standard_normal(m)
    g = sor_gpu(hx, ai, aj, aa, b, 1.0, LSYM | ZERO, 0.0, 1, 1, x0, want_mode="levels")
    assert np.array_equal(g, sor_cpu(ai, aj, aa, b, 1.0, LSYM | ZERO, 0.0, 1, 1, x0))


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
    o = exact_solve("gmres", ai, aj, aa, b, pc="sor", rtol=1e-8)
    compare(g, o, 1e-8)
    g = solve_gpu("cg", ai, aj, aa, b, pc="sor", rtol=1e-8)
    o = exact_solve("cg", ai, aj, aa, b, pc="sor", rtol=1e-8)
    compare(g, o, 1e-8)


@pytest.mark.parametrize("env", [{}, {"HIPX_SOR_SPLIT": "0"}, {"HIPX_SOR_SPLIT": "2"}, {"HIPX_SOR_STAGGER": "0"}, {"HIPX_SOR_SPLIT": "2", "HIPX_SOR_WG_PER_CU": "1"},
                                 {"HIPX_SOR_LOCKSTEP": "1"}, {"HIPX_SOR_LOCKSTEP": "1", "HIPX_SOR_STAGGER": "0"}, {"HIPX_SOR_LOCKSTEP": "0", "HIPX_SOR_ROLEMAP": "0"}],
                         ids=["default", "nosplit", "split-all", "nostagger", "split-all-1wg", "lockstep", "lockstep-nostagger", "free-running-roles-by-wave"])
def test_strand_kernel_variants_bit_exact(env):
    """The strand kernels' variants (two-wave / split C-F-F-loader kernel for forward only or for every kind, staggered panel
    boundaries on / off, one or two workgroups per CU) all give the reference's bits: tests/_sor_variant_worker.py, one process per
    variant (the switches are read once per process)."""
    import subprocess
    import sys
    e = dict(os.environ, **env)
    e["PYTHONPATH"] = os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__))), e.get("PYTHONPATH", "")])
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_sor_variant_worker.py")], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0 and "SOR_VARIANT_OK" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("coop", ["0", "1"])
@pytest.mark.parametrize("flag,omega,its", [(SYM | ZERO, 1.0, 1), (SYM, 1.3, 2), (BWD, 0.9, 2), (FWD | ZERO, 1.0, 2), (EISENSTAT, 1.0, 1)])
def test_both_forms_of_the_dependency_driven_sweep(hx, coop, flag, omega, its):
    """HIPX_SOR_DEP_COOP: 1 = sor_dep_coop_kernel (16 lanes per row: narrow levels), 0 = sor_dep_kernel (one lane per row: wide levels) --
    the library picks by the average level width; both are MatSOR_SeqAIJ bit for bit (an unstructured matrix and a stencil)."""
    rng = np.random.default_rng(23)
    for ai, aj, aa in (sor_ready(*random_csr(3000, 3000, rng, 40, empty_frac=0.0), rng), orc.stencil("27pt", 11)):
        N = len(ai) - 1
        b, x0 = rng.standard_normal(N), rng.standard_normal(N)
        os.environ["HIPX_SOR_DEP_COOP"] = coop
        try:
            g = sor_gpu(hx, ai, aj, aa, b, omega, flag, 0.0, its, 1, x0, mode="dep")
        finally:
            os.environ.pop("HIPX_SOR_DEP_COOP", None)
        assert np.array_equal(g, sor_cpu(ai, aj, aa, b, omega, flag, 0.0, its, 1, x0))


def sor_ready(ai, aj, aa, rng):
    """random_csr with a nonzero diagonal entry in every row (MatSOR needs one: aij.c:1809)."""
    m = len(ai) - 1
    rows = np.repeat(np.arange(m), np.diff(ai))
    import scipy.sparse as sp
    A = sp.csr_matrix((aa, aj, ai), shape=(m, m)) + sp.diags(4.0 + rng.random(m))
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


def box_csr(nx, ny, nz, full27, seed=3):
    """constant-coefficient box stencil in natural ordering, nx x ny x nz (x fastest): 27-point (one value per position of the 3 x 3 x 3 cube, all
    couplings negative) or the 7-point star"""
    rng = np.random.default_rng(seed)
    N = nx * ny * nz
    I, Jg, K = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    i, j, k = I.ravel(order="F"), Jg.ravel(order="F"), K.ravel(order="F")
    rows, cols, vals = [], [], []
    for dk in (-1, 0, 1):
        for dj in (-1, 0, 1):
            for di in (-1, 0, 1):
                nd = abs(di) + abs(dj) + abs(dk)
                if not full27 and nd > 1:
                    continue
                ok = (i + di >= 0) & (i + di < nx) & (j + dj >= 0) & (j + dj < ny) & (k + dk >= 0) & (k + dk < nz)
                r = (i + nx * (j + ny * k))[ok]
                rows.append(r)
                cols.append(r + di + nx * dj + nx * ny * dk)
                vals.append(np.full(len(r), 7.5 + rng.random() if nd == 0 else -(0.125 + 0.5 * rng.random())))
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    o = np.lexsort((cols, rows))
    ai = np.zeros(N + 1, np.int32)
    ai[1:] = np.cumsum(np.bincount(rows, minlength=N))
    return ai, cols[o].astype(np.int32), vals[o]


@pytest.mark.parametrize("nx,ny,nz,full27", [(8, 70, 6, True), (8, 70, 6, False), (16, 66, 9, False), (6, 130, 3, True), (12, 12, 12, True), (10, 5, 70, True), (4, 64, 4, True),
                                             (32, 40, 5, True), (64, 64, 64, True), (64, 64, 64, False), (130, 9, 11, True), (33000, 3, 5, True)])
@pytest.mark.parametrize("flag,omega,shift", [(LSYM | ZERO, 1.0, 0.0), (SYM | ZERO, 1.0, 0.0), (FWD | ZERO, 1.0, 0.0), (BWD | ZERO, 1.0, 0.0), (LSYM | ZERO, 1.3, 0.25), (LFWD | ZERO, 0.8, 0.0),
                                              (LBWD | ZERO, 1.3, 0.0)])
def test_plane_march_bit_exact(hx, nx, ny, nz, full27, flag, omega, shift):
    """Round 5: the plane-march schedule (csrc/hipx_sorbox.hip) -- zero-guess forward / backward / symmetric sweeps of constant-coefficient box
    stencils: blocks that cross the grid's edges (the skewed block boundaries leave lanes without a line), grids of one block and of several, chunks
    of planes that do not fill a workgroup, lines shorter than a staging group, lines of more than 32768 rows (round 6: the packed readiness check of
    the main loop needs counters below 0x8000 -- longer lines take the spelled-out checks all the way), omega / shift -- bit-identical to MatSOR_SeqAIJ
    (aij.c:1930-1958)."""
    ai, aj, aa = box_csr(nx, ny, nz, full27)
    N = nx * ny * nz
    b = np.random.default_rng(5).standard_normal(N)
    g = sor_gpu(hx, ai, aj, aa, b, omega, flag, shift, 1, 1, np.zeros(N), mode="box")
    o = sor_cpu(ai, aj, aa, b, omega, flag, shift, 1, 1, np.zeros(N))
    assert np.array_equal(g, o), np.abs(g - o).max()


@pytest.mark.parametrize("nx,ny,nz,full27", [(64, 30, 32, True), (256, 40, 20, True), (40, 70, 21, False), (10, 5, 70, True)])
def test_plane_march_mailbox_is_reusable(hx, nx, ny, nz, full27):
    """Round 6: the hop between chunks goes through a mailbox that is filled with the sentinel ONCE per matrix -- the poller that reads an entry puts the
    sentinel back.  One matrix, eight applications of different kinds and right-hand sides in a row (forward, backward, symmetric; the forward sweep
    inside a symmetric application sends its planes through the same mailbox as the backward one): an entry left behind by one sweep would be taken for
    the next sweep's row.  5 to 18 chunks; one block of lines and several; a grid whose upper chunks have no line of block 0."""
    from petsc_amd import _lib
    ai, aj, aa = box_csr(nx, ny, nz, full27)
    N = nx * ny * nz
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    rng = np.random.default_rng(11)
    try:
        with sor_mode("box"):
            for rep, (flag, omega, shift) in enumerate([(LSYM | ZERO, 1.0, 0.0), (FWD | ZERO, 1.0, 0.0), (BWD | ZERO, 1.3, 0.0), (LSYM | ZERO, 1.3, 0.25), (LSYM | ZERO, 1.0, 0.0), (LBWD | ZERO, 0.8, 0.0),
                                                        (LFWD | ZERO, 1.0, 0.0), (SYM | ZERO, 1.0, 0.0)]):
                b = rng.standard_normal(N)
                B, X = _lib.DVec(N, b), _lib.DVec(N, np.zeros(N))
                _lib.chk(hx.hipxMatSOR(A, B.ptr, omega, flag, shift, 1, 1, X.ptr))
                used = C.c_int(-2)
                _lib.chk(hx.hipxMatGetSORMode(A, C.byref(used)))
                assert used.value == MODES["box"]
                g = X.get()
                B.free()
                X.free()
                o = sor_cpu(ai, aj, aa, b, omega, flag, shift, 1, 1, np.zeros(N))
                assert np.array_equal(g, o), (rep, flag, np.abs(g - o).max())
    finally:
        _lib.mat_destroy(A)


def test_plane_march_declines_what_it_does_not_cover(hx):
    """Couplings of both signs, a second diagonal value, an odd line length, arbitrary values, nonzero-guess sweeps: the plane march says no and the
    strands / levels run (still bit-identical)."""
    ai, aj, aa = box_csr(12, 10, 9, True)
    N = 12 * 10 * 9
    rng = np.random.default_rng(8)
    b, x0 = rng.standard_normal(N), rng.standard_normal(N)
    rows = np.repeat(np.arange(N), np.diff(ai))
    mixed = aa.copy()
    mixed[(aj == rows + 1) | (aj == rows - 1)] *= -1.0  # the x-couplings positive, the others negative
    twodiag = aa.copy()
    twodiag[(aj == rows) & (rows % 12 == 0)] += 1.0     # another diagonal value on the rows of one face
    for vals in (mixed, twodiag):
        g = sor_gpu(hx, ai, aj, vals, b, 1.0, LSYM | ZERO, 0.0, 1, 1, x0, want_mode="strand")
        assert np.array_equal(g, sor_cpu(ai, aj, vals, b, 1.0, LSYM | ZERO, 0.0, 1, 1, x0))
    ai2, aj2, aa2 = box_csr(11, 10, 9, True)  # odd line length
    N2 = 11 * 10 * 9
    b2 = rng.standard_normal(N2)
    g = sor_gpu(hx, ai2, aj2, aa2, b2, 1.0, LSYM | ZERO, 0.0, 1, 1, np.zeros(N2), want_mode="strand")
    assert np.array_equal(g, sor_cpu(ai2, aj2, aa2, b2, 1.0, LSYM | ZERO, 0.0, 1, 1, np.zeros(N2)))
    g = sor_gpu(hx, ai, aj, aa, b, 1.0, LSYM, 0.0, 2, 1, x0, want_mode="strand")  # nonzero guess, two iterations
    assert np.array_equal(g, sor_cpu(ai, aj, aa, b, 1.0, LSYM, 0.0, 2, 1, x0))
    with pytest.raises(Exception):
        sor_gpu(hx, ai, aj, aa, b, 1.0, LSYM, 0.0, 2, 1, x0, mode="box")


@pytest.mark.parametrize("stencil,dims", [(27, (128, 128, 128)), (7, (128, 128, 128)), (7, (512, 256, 96)), (27, (256, 256, 256))])
def test_plane_march_at_scale_equals_the_level_ordered_sweep(hx, stencil, dims):
    """2 M - 16.8 M rows (several rounds of workgroups per CU, 64 chunks of planes): the plane march against the level-ordered sweep of the same
    library (itself held to the CPU loop by test_sor_bit_exact_at_scale), PCSOR's default application."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from petsc_amd import _lib
    _, ks = _lib.load()
    N = dims[0] * dims[1] * dims[2]
    ai, aj, aa = bench.assemble(ks, stencil, dims, 0, N)
    b = 1.0 + (np.arange(N) % 17) / 17.0
    g = sor_gpu(hx, ai, aj, aa, b, 1.0, LSYM | ZERO, 0.0, 1, 1, np.zeros(N), mode="box")
    o = sor_gpu(hx, ai, aj, aa, b, 1.0, LSYM | ZERO, 0.0, 1, 1, np.zeros(N), mode="dep")
    assert np.array_equal(g, o), np.abs(g - o).max()
