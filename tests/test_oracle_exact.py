"""CPU: pins the "exact reduction" yardstick of the history parity tests.

1. Dot2 (oracle/exactblas.c `exactblas_dot2`, and the same algorithm inside oracle/petsc_oracle.c's exact mode) against EXACT
   rational arithmetic (python Fractions, rounded once) on adversarial vectors: heavy cancellation, 1e+-150 magnitude mixes,
   long vectors of one sign.  Dot2 is "as if in twice the working precision" (Ogita-Rump-Oishi Prop. 5.5): the test asserts the
   published bound and, where the condition number is moderate, that the result IS the correctly rounded value.
2. The reference's OWN KSPSolve (oracle/_ref/bin/ref_driver) with oracle/libexactblas.so LD_PRELOADed -- its cg.c / gmres.c
   control flow, its MatMult_SeqAIJ, its Vec loops, only the BLAS reductions swapped -- against the restated oracle's exact
   mode: CG histories are BIT-IDENTICAL, i.e. the restated yardstick the GPU tests use is the reference itself.
"""
import ctypes as C
import os
import subprocess
import sys
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402

SHIM = os.path.join(ROOT, "oracle", "libexactblas.so")
REF = os.path.join(ROOT, "oracle", "_ref", "bin", "ref_driver")


def shim():
    if not os.path.exists(SHIM):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    L = C.CDLL(SHIM)
    L.exactblas_dot2.restype = C.c_double
    L.exactblas_dot2.argtypes = [C.c_long, C.c_void_p, C.c_void_p]
    return L


def exact_dot(x, y):
    """(correctly rounded x.y, exact x.y, exact |x|.|y|) through rational arithmetic"""
    s = sum((Fraction(float(a)) * Fraction(float(b)) for a, b in zip(x, y)), Fraction(0))
    sa = sum((abs(Fraction(float(a)) * Fraction(float(b))) for a, b in zip(x, y)), Fraction(0))
    return float(s), s, sa  # float(Fraction) rounds to nearest even: the correctly rounded value


def cases():
    rng = np.random.default_rng(7)
    out = {}
    n = 4000
    out["same_sign_long"] = (rng.random(n) + 0.5, rng.random(n) + 0.5)
    x = rng.standard_normal(n)
    out["random_signs"] = (x, rng.standard_normal(n))
    # cancellation: the sum is ~1e-10 of the sum of magnitudes
    x = rng.standard_normal(n)
    y = rng.standard_normal(n)
    x2 = np.concatenate([x, x, [1e-10]])
    y2 = np.concatenate([y, -y, [1.0]])
    out["cancellation_1e10"] = (x2, y2)
    # magnitude mix 1e+-150: products span 600 orders of magnitude, the big ones cancel exactly
    big = np.array([1e150, -1e150, 3e149, -3e149])
    out["magnitude_mix"] = (np.concatenate([big, rng.random(500) * 1e-150, rng.random(500)]), np.concatenate([np.array([1e150, 1e150, 2e148, 2e148]), rng.random(500) * 1e-150, rng.random(500)]))
    # a CG-like inner product: p.Ap of a smooth vector on the 1-D Laplacian (all partial sums positive, terms tiny vs the sum)
    k = np.arange(n)
    p = np.sin(np.pi * (k + 1) / (n + 1))
    ap = 2 * p - np.concatenate([[0], p[:-1]]) - np.concatenate([p[1:], [0]])
    out["p_dot_Ap"] = (p, ap)
    return out


@pytest.mark.parametrize("name", list(cases().keys()))
def test_dot2_against_exact_rational_arithmetic(name):
    x, y = cases()[name]
    x = np.ascontiguousarray(x, np.float64)
    y = np.ascontiguousarray(y, np.float64)
    n = len(x)
    rounded, s, sa = exact_dot(x, y)
    got_shim = shim().exactblas_dot2(n, x.ctypes.data, y.ctypes.data)
    L = orc.lib()
    L.orc_set_exact_reductions(1)
    try:
        got_orc = L.orc_VecDot_Seq(n, orc.P(x), orc.P(y))
    finally:
        L.orc_set_exact_reductions(0)
    assert got_shim == got_orc  # the two implementations of Dot2 agree bit for bit
    eps = 2.0 ** -53
    gamma = n * eps / (1 - n * eps)
    bound = eps * abs(s) + Fraction(gamma) ** 2 * sa  # Ogita-Rump-Oishi Prop. 5.5
    assert abs(Fraction(got_shim) - s) <= bound * Fraction(1001, 1000), (name, got_shim, rounded)
    cond = float(2 * sa / abs(s)) if s != 0 else np.inf
    if cond * n * n * eps < 0.01:  # well inside the regime where the second term of the bound is invisible: correctly rounded
        assert got_shim == rounded, (name, got_shim, rounded, cond)
    # and plain left-to-right double summation is NOT good enough on the hard cases (the test would be vacuous otherwise)
    if name in ("cancellation_1e10", "magnitude_mix"):
        plain = 0.0
        for a, b in zip(x, y):
            plain += a * b
        assert plain != rounded


def run_ref(args, exact):
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1")
    if exact:
        env["LD_PRELOAD"] = SHIM
    out = subprocess.run([REF] + args + ["-mat_type", "aij", "-vec_type", "standard", "-history"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600).stdout
    return np.array([float(l.split()[2]) for l in out.splitlines() if l.startswith("hist ")]), out


needs_ref = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (needs /root/reference at build time)")


@needs_ref
@pytest.mark.parametrize("stencil,n,pc,its", [("7pt", 40, "jacobi", 40), ("27pt", 20, "jacobi", 30), ("7pt", 24, "none", 30), ("7pt", 24, "sor", 20)])
def test_reference_with_exact_blas_equals_restated_exact_oracle_cg(stencil, n, pc, its):
    """KSPSolve_CG of the REFERENCE (its own executable, BLAS reductions swapped for Dot2 by LD_PRELOAD) == the oracle's exact
    mode, bit for bit: every elementwise step and every row sum of the restatement is the reference's."""
    ai, aj, aa = orc.stencil(stencil, n)
    b = orc.matmult(ai, aj, aa, np.ones(n ** 3))
    h_orc = orc.ksp_solve("cg", ai, aj, aa, b, pc=pc, rtol=1e-50, max_it=its, exact=True)[3]
    h_ref, out = run_ref(["-stencil", stencil[:-2], "-n", str(n), "-ksp_type", "cg", "-pc_type", pc, "-ksp_rtol", "1e-50", "-ksp_max_it", str(its), "-ksp_norm_type", "preconditioned"], True)
    assert len(h_ref) == len(h_orc) == its + 1, out[-800:]
    assert np.array_equal(h_ref, h_orc), np.abs(h_ref - h_orc).max()
    # and the shim is really in the path: the MKL run differs (by rounding only)
    h_mkl, _ = run_ref(["-stencil", stencil[:-2], "-n", str(n), "-ksp_type", "cg", "-pc_type", pc, "-ksp_rtol", "1e-50", "-ksp_max_it", str(its), "-ksp_norm_type", "preconditioned"], False)
    assert not np.array_equal(h_mkl, h_ref) and np.abs(h_mkl - h_ref).max() <= 1e-11 * h_ref[0]


@needs_ref
@pytest.mark.parametrize("ksp", ["pipecg", "groppcg"])
@pytest.mark.parametrize("stencil,n,pc,norm,its", [("7pt", 32, "jacobi", "preconditioned", 40), ("27pt", 16, "none", "unpreconditioned", 30), ("7pt", 20, "jacobi", "natural", 30),
                                                    ("27pt", 14, "sor", "preconditioned", 15)])
def test_reference_with_exact_blas_equals_restated_exact_oracle_pipelined_cg(ksp, stencil, n, pc, norm, its):
    """Round 6: KSPSolve_PIPECG / KSPSolve_GROPPCG of the REFERENCE (pipecg.c:20-160, groppcg.c:23-140; VecDotBegin/NormBegin ... End, comb.c) with exact BLAS
    reductions == the oracle's restatements in exact mode, bit for bit (history and iteration count, including pipecg.c:160's `i <= max_it`)."""
    ai, aj, aa = orc.stencil(stencil, n)
    b = orc.matmult(ai, aj, aa, np.ones(n ** 3))
    nt = {"preconditioned": 1, "unpreconditioned": 2, "natural": 3}[norm]
    _, its_o, reason_o, h_orc = orc.ksp_solve(ksp, ai, aj, aa, b, pc=pc, rtol=1e-50, max_it=its, normtype=nt, exact=True)
    h_ref, out = run_ref(["-stencil", stencil[:-2], "-n", str(n), "-ksp_type", ksp, "-pc_type", pc, "-ksp_rtol", "1e-50", "-ksp_max_it", str(its), "-ksp_norm_type", norm], True)
    assert len(h_ref) == len(h_orc) == its + 1, out[-800:]
    assert np.array_equal(h_ref, h_orc), np.abs(h_ref - h_orc).max()
    import re
    m = re.search(r"iterations (\d+) reason (-?\d+)", out)
    assert (its_o, reason_o) == (int(m.group(1)), int(m.group(2))) == (its + 1 if ksp == "pipecg" else its, -3)


@needs_ref
def test_reference_with_exact_blas_vs_restated_exact_oracle_gmres_sor():
    """GMRES(30) + SOR: dgemv 'T' (VecMDot) is Dot2 per column in both; dgemv 'N' (VecMAXPY) is rounded once per element in the
    shim and in VecMAXPY_Seq's grouping in the oracle -- a last-bit elementwise difference, so the histories agree to rounding
    (and are far closer to each other than the MKL run is to either)."""
    n, its = 16, 45
    ai, aj, aa = orc.stencil("27pt", n)
    b = orc.matmult(ai, aj, aa, np.ones(n ** 3))
    h_orc = orc.ksp_solve("gmres", ai, aj, aa, b, pc="sor", rtol=1e-50, max_it=its, exact=True)[3]
    a = ["-stencil", "27", "-n", str(n), "-ksp_type", "gmres", "-pc_type", "sor", "-ksp_rtol", "1e-50", "-ksp_max_it", str(its)]
    h_ref, out = run_ref(a, True)
    assert len(h_ref) == len(h_orc), out[-800:]
    d = np.abs(h_ref - h_orc) / h_ref[0]
    assert d.max() <= 1e-13, d.max()


@pytest.mark.parametrize("kind,n,pc", [("7pt", 24, "jacobi"), ("27pt", 14, "jacobi"), ("7pt_box", (12, 10, 7), "none"), ("7pt", 16, "none")])
def test_streamed_exact_cg_equals_oracle_exact_mode(kind, n, pc):
    """oracle/stream_cg.py (the matrix-free-by-slabs exact CG that makes the goldens for systems beyond 32-bit nonzero counts:
    27-pt 512^3, the config-5 boxes) == the C oracle's exact mode, bit for bit, where both fit."""
    import stream_cg
    N = int(np.prod(n)) if isinstance(n, tuple) else n ** 3
    ai, aj, aa = orc.stencil(kind, n)
    b = orc.matmult(ai, aj, aa, np.ones(N))
    xo, _, _, h_orc = orc.ksp_solve("cg", ai, aj, aa, b, pc=pc, rtol=1e-50, max_it=25, exact=True)
    op = stream_cg.StreamOperator(kind, n, N, slab_rows=1000, threads=3)  # ragged slabs on purpose
    h, err = stream_cg.cg_exact(op, pc, 25)
    assert np.array_equal(h, h_orc), np.abs(h - h_orc).max()
    assert abs(err - np.linalg.norm(xo - 1.0)) <= 1e-12 * err


def test_committed_exact_goldens_are_reproducible_at_their_smallest_size():
    """tests/golden/exact_histories.json (made by tests/golden/make_exact_golden.py) carries one small entry that this test
    regenerates with the C oracle: a stale or hand-edited golden file fails here."""
    import json
    p = os.path.join(ROOT, "tests", "golden", "exact_histories.json")
    if not os.path.exists(p):
        pytest.skip("goldens not generated yet")
    g = json.load(open(p))
    e = g["cg_jacobi_7pt_32"]
    ai, aj, aa = orc.stencil("7pt", 32)
    b = orc.matmult(ai, aj, aa, np.ones(32 ** 3))
    h = orc.ksp_solve("cg", ai, aj, aa, b, pc="jacobi", rtol=1e-50, max_it=len(e["history"]) - 1, exact=True)[3]
    assert np.array_equal(h, np.array([float.fromhex(v) for v in e["history_hex"]]))
    assert np.array_equal(h, np.array(e["history"]))


@pytest.mark.parametrize("n,nranks", [(24, 3), (32, 8), (40, 5)])
def test_streamed_gmres_sor_equals_the_stored_matrix_oracle(n, nranks):
    """oracle/stream_gmres.py (round 5: the yardstick of BASELINE config 3 at its real shape, 27-pt 512^3 on 8 ranks, 3.6e9 nonzeros never held as one
    matrix): the C oracle's GMRES loop over per-rank products (MatMult_MPIAIJ's order: diagonal block, then the off-diagonal terms added) and per-rank
    local sweeps assembled slab by slab -- bit-identical, over a restart, to the same loop on the stored matrix (orc.ksp_solve with nranks), whose
    products take the same order since this round (orc.matmult_mpi)."""
    import stream_gmres as sg
    ai, aj, aa = orc.stencil("27pt", n)
    op = sg.StreamPartitionedOperator("27pt", n, nranks, sub_rows=7000)
    b = np.empty(n ** 3)
    op.mult(np.ones(n ** 3), b)
    assert np.array_equal(b, orc.matmult_mpi(ai, aj, aa, np.ones(n ** 3), nranks))
    h0 = orc.ksp_solve("gmres", ai, aj, aa, b, pc="sor", rtol=1e-50, max_it=35, nranks=nranks, exact=True)[3]
    h1 = sg.gmres_sor_exact(op, 35)
    assert len(h0) == len(h1) == 36 and np.array_equal(h0, h1)
    x = np.random.default_rng(n).standard_normal(n ** 3)
    y = np.empty(n ** 3)
    assert np.array_equal(op.mult(x, y), orc.matmult_mpi(ai, aj, aa, x, nranks))


def test_committed_np_goldens_are_the_oracles_with_the_partitioned_product():
    """tests/golden/exact_histories.json gmres_sor_27pt_128_np{2,4,8} (regenerated in round 5): the oracle's exact mode with b = A * 1 and every product
    formed as MatMult_MPIAIJ forms them; the 512^3 entries come from the streamed form of the same loop."""
    import json
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "exact_histories.json")))
    ai, aj, aa = orc.stencil("27pt", 128)
    b = orc.matmult_mpi(ai, aj, aa, np.ones(128 ** 3), 8)
    h = orc.ksp_solve("gmres", ai, aj, aa, b, pc="sor", rtol=1e-50, max_it=12, nranks=8, exact=True)[3]
    want = [float.fromhex(v) for v in g["gmres_sor_27pt_128_np8"]["history_hex"]][:len(h)]
    assert list(h) == want
    for k in ("gmres_sor_27pt_512_np8", "gmres_sor_27pt_512_np4"):
        assert g[k]["source"] == "stream" and len(g[k]["history_hex"]) >= 17
