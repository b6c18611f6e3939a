"""CPU: the oracle (C restatement) against outputs of the REFERENCE ITSELF (tests/golden/ref_runs.json, produced by
oracle/_ref/bin/ref_driver = the reference's libpetsc compiled by oracle/build_ref.py; script: tests/golden/make_golden.py).
SpMV must agree bit for bit; Krylov histories to rounding of the BLAS reductions (the reference calls MKL ddot/dgemv)."""
import json
import os

import numpy as np
import pytest

import oracle as orc

R = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_runs.json")))


def parse(args):
    t = args.split()
    d, flags, i = {}, [], 0
    while i < len(t):
        if i + 1 < len(t) and not (t[i + 1].startswith("-") and not t[i + 1][1:2].isdigit()):
            d[t[i]] = t[i + 1]
            i += 2
        else:
            flags.append(t[i])
            i += 1
    return d, flags


def build(d):
    st = int(d["-stencil"])
    n = int(d["-n"])
    if st == 5:
        return orc.stencil("5pt", n, m=int(d.get("-m", n)))
    return orc.stencil("7pt" if st == 7 else "27pt", n)


def solve_kwargs(d, flags):
    kw = dict(pc=d.get("-pc_type", "jacobi"), rtol=float(d["-ksp_rtol"]))
    kw["normtype"] = {"preconditioned": 1, "unpreconditioned": 2, "natural": 3}[d.get("-ksp_norm_type", "preconditioned")]
    kw["restart"] = int(d.get("-ksp_gmres_restart", 30))
    kw["refine"] = 2 if d.get("-ksp_gmres_cgs_refinement_type") == "refine_always" else 0
    kw["sor_flag"] = 3 if "-pc_sor_symmetric" in flags else 12
    if "-ksp_max_it" in d:
        kw["max_it"] = int(d["-ksp_max_it"])
    return d.get("-ksp_type", "gmres"), kw


@pytest.mark.parametrize("name", sorted(R["spmv"]))
def test_spmv_bit_exact_vs_reference_matmult(name):
    d, _ = parse(R["spmv"][name]["args"])
    ai, aj, aa = build(d)
    N = len(ai) - 1
    x = 1.0 + (np.arange(N) % 17) / 17.0
    y = orc.matmult(ai, aj, aa, x)
    yref = np.array([float(v) for v in R["spmv"][name]["y"]])
    assert np.array_equal(y, yref)


@pytest.mark.parametrize("name", sorted(R["ksp"]))
def test_krylov_history_vs_reference(name):
    g = R["ksp"][name]
    d, flags = parse(g["args"])
    ai, aj, aa = build(d)
    b = orc.matmult(ai, aj, aa, np.ones(len(ai) - 1))
    kind, kw = solve_kwargs(d, flags)
    x, its, reason, hist = orc.ksp_solve(kind, ai, aj, aa, b, **kw)
    href = np.array([float(v) for v in g["history"]])
    assert its == g["iterations"] and reason == g["reason"]
    assert len(hist) == len(href)
    # the pipelined recurrences (pipecg.c, groppcg.c) carry the rounding of every reduction forward (r, u = B r, w = A u are all recurred): the distance between
    # MKL's ddot and the plain loop grows to a few 1e-8 of the residual at 50 iterations; their tight pin is the exact-reduction history (test_oracle_exact.py)
    loose = kind in ("pipecg", "groppcg")
    assert np.abs(hist - href).max() <= (1e-11 if loose else 1e-12) * href[0]
    assert (np.abs(hist - href) / href).max() <= (1e-6 if loose else 1e-8)
    assert abs(np.linalg.norm(x - 1) - g["error"]) <= (1e-5 if loose else 1e-9) * max(g["error"], 1e-30) + 1e-13


# ---- matrices with inodes, LIVE against the reference (where oracle/_ref is built: this container and the GPU box; the committed vectors
# of tests/golden/inode_sor.json cover every other place, tests/test_oracle.py)
REF_EXE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "bin", "ref_driver")


def _ref_lines(args, tag):
    import subprocess
    r = subprocess.run([REF_EXE] + args + ["-mat_type", "aij", "-vec_type", "standard"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300,
                       env=dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-2000:]
    return np.array([float(ln.split()[2]) for ln in r.stdout.splitlines() if ln.startswith(tag)])


@pytest.mark.parametrize("seed,nnodes,sizes", [(101, 90, (1, 2, 3, 4, 5)), (102, 150, (3,)), (103, 120, (2, 5)), (104, 80, (4, 1))])
def test_inode_restatements_against_the_live_reference(tmp_path, seed, nnodes, sizes):
    """Other matrices than the golden one (node-size mixes, seeds): MatMult (inode.c:356), MatMultAdd-free y = A x, and MatSOR in five sweep
    configurations -- the reference's executable run here, the oracle's dispatching restatements held to it bit for bit."""
    if not os.path.exists(REF_EXE):
        pytest.skip("oracle/_ref is not built here")
    import ctypes as C
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from petsc_amd import matio
    from surrogates import inode_matrix
    ai, aj, aa = inode_matrix(nnodes=nnodes, seed=seed, sizes=sizes, long_run=len(sizes) > 2)
    aa = aa * (1.0 + 1e-3 * np.random.default_rng(seed).standard_normal(len(aa)))  # no longer multiples of 2^-10: the summation orders show
    N = len(ai) - 1
    f = str(tmp_path / "m.bin")
    matio.write_petsc_binary(f, ai, aj, aa)
    x = 1.0 + (np.arange(N) % 17) / 17.0
    y = _ref_lines(["-f", f, "-dump_y", "-ksp_max_it", "1"], "y ")
    assert np.array_equal(y, orc.matmult_ref(ai, aj, aa, x)) and not np.array_equal(y, orc.matmult_ref(ai, aj, aa, x, no_inode=True))
    b = orc.matmult_ref(ai, aj, aa, np.ones(N))  # ref_driver's b = MatMult(A, 1): the inode product
    for flag, its in ((16 | 12, 1), (3, 2), (2, 2), (16 | 1, 2), (32, 1)):
        xr = _ref_lines(["-f", f, "-dump_sor", str(flag), "-sor_its", str(its), "-ksp_max_it", "1"], "sor ")
        xo = 0.5 + (np.arange(N) % 7) / 7.0
        assert orc.lib().orc_MatSOR_SeqAIJ_dispatch(N, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(b), C.c_double(1.0), flag, C.c_double(0.0), its, 1, orc.P(xo), 0) == 0
        assert len(xr) == N and np.array_equal(xr, xo), (flag, its, np.abs(xr - xo).max())
