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
    assert np.abs(hist - href).max() <= 1e-12 * href[0]
    assert (np.abs(hist - href) / href).max() <= 1e-8
    assert abs(np.linalg.norm(x - 1) - g["error"]) <= 1e-9 * max(g["error"], 1e-30) + 1e-13
