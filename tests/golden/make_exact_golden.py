#!/usr/bin/env python
"""Generates tests/golden/exact_histories.json: residual histories with EXACT (twice-working-precision) reductions for the
configurations bench.py and the GPU tests time at BASELINE scale -- the yardstick a multi-GPU run checks itself against when no
CPU reference can be run beside it (N > 1 ranks, 27-pt 512^3, the config-5 boxes).  Run in the build container (CPU only):

    python tests/golden/make_exact_golden.py [--only KEY ...]

Sources, all pinned by tests/test_oracle_exact.py:
  * "reference+shim": the REFERENCE's own KSPSolve (oracle/_ref/bin/ref_driver) with oracle/libexactblas.so LD_PRELOADed;
  * "oracle-exact":   the C oracle's exact mode (bit-identical to reference+shim for CG; per-rank local SOR through nranks);
  * "stream":         oracle/stream_cg.py, for systems beyond the 32-bit nonzero counts of both (bit-identical to oracle-exact
                      wherever both fit).
Values are stored as decimal repr AND as C99 hex floats (exact).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402
import stream_cg  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "exact_histories.json")
SHIM = os.path.join(ROOT, "oracle", "libexactblas.so")
REF = os.path.join(ROOT, "oracle", "_ref", "bin", "ref_driver")


def entry(hist, source, what, extra=None):
    e = {"what": what, "source": source, "history": [float(v) for v in hist], "history_hex": [float(v).hex() for v in hist]}
    if extra:
        e.update(extra)
    return e


def ref_shim(stencil, n, ksp, pc, its, m=None):
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1", LD_PRELOAD=SHIM)
    a = [REF, "-stencil", str(stencil), "-n", str(n), "-ksp_type", ksp, "-pc_type", pc, "-ksp_rtol", "1e-50", "-ksp_max_it", str(its), "-history", "-mat_type", "aij", "-vec_type", "standard"]
    if m is not None:  # the 2-D 5-point operator of ex2.c on an m x n grid
        a += ["-m", str(m)]
    if ksp == "cg":
        a += ["-ksp_norm_type", "preconditioned"]
    out = subprocess.run(a, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=7200).stdout
    return np.array([float(l.split()[2]) for l in out.splitlines() if l.startswith("hist ")])


def jobs():
    J = {}

    def stream(kind, n, N, pc, its, what):
        op = stream_cg.StreamOperator(kind, n, N)
        h, err = stream_cg.cg_exact(op, pc, its)
        return entry(h, "stream", what, {"error_norm": err})
    J["cg_jacobi_7pt_32"] = lambda: entry(orc.ksp_solve("cg", *orc.stencil("7pt", 32), orc.matmult(*orc.stencil("7pt", 32), np.ones(32 ** 3)), pc="jacobi", rtol=1e-50, max_it=20, exact=True)[3],
                                          "oracle-exact", "7-pt 32^3 CG+Jacobi (self-check entry of tests/test_oracle_exact.py)")
    J["cg_jacobi_7pt_64"] = lambda: entry(ref_shim(7, 64, "cg", "jacobi", 40), "reference+shim", "7-pt Poisson 64^3, KSPCG + PCJACOBI; 40 iterations (small multi-rank smoke runs of bench.py)")
    J["cg_none_7pt_64x64x16"] = lambda: stream("7pt_box", (64, 64, 16), 64 * 64 * 16, "none", 20, "7-pt 64 x 64 x 16 box, KSPCG + PCNONE; 20 iterations (weak-mode smoke runs: --grid 64 --scaling weak on 2 ranks)")
    J["cg_jacobi_7pt_256"] = lambda: entry(ref_shim(7, 256, "cg", "jacobi", 60), "reference+shim", "BASELINE config 2: 7-pt Poisson 256^3, KSPCG + PCJACOBI, b = A*1, x0 = 0; 60 iterations")
    # round 6: BASELINE config 1's operator (ex2.c:70-94) at HBM size -- north_star's 5-point leg
    J["cg_jacobi_5pt_4096x4096x1"] = lambda: entry(ref_shim(5, 4096, "cg", "jacobi", 40, m=4096), "reference+shim", "2-D 5-pt Laplacian (ex2.c) 4096 x 4096 = 16.8 M rows, KSPCG + PCJACOBI, b = A*1, x0 = 0; 40 iterations")
    # round 6: the pipelined variants (pipecg.c, groppcg.c) from the REFERENCE with exact BLAS reductions: the yardsticks of pipecghipx / the batched lazy queue
    for kk in ("pipecg", "groppcg"):
        J["%s_jacobi_7pt_64" % kk] = (lambda kk=kk: entry(ref_shim(7, 64, kk, "jacobi", 40), "reference+shim", "7-pt Poisson 64^3, KSP%s + PCJACOBI; 40 iterations" % kk.upper()))
        J["%s_jacobi_7pt_128" % kk] = (lambda kk=kk: entry(ref_shim(7, 128, kk, "jacobi", 40), "reference+shim", "7-pt Poisson 128^3, KSP%s + PCJACOBI; 40 iterations" % kk.upper()))
        J["%s_jacobi_27pt_96" % kk] = (lambda kk=kk: entry(ref_shim(27, 96, kk, "jacobi", 30), "reference+shim", "27-pt (bench_kspsolve.c) 96^3, KSP%s + PCJACOBI; 30 iterations" % kk.upper()))
    J["groppcg_jacobi_7pt_256"] = lambda: entry(ref_shim(7, 256, "groppcg", "jacobi", 40), "reference+shim", "BASELINE config 2's system under KSPGROPPCG: 7-pt Poisson 256^3 + PCJACOBI; 40 iterations")
    J["pipecg_jacobi_7pt_256"] = lambda: entry(ref_shim(7, 256, "pipecg", "jacobi", 40), "reference+shim", "BASELINE config 2's system under KSPPIPECG: 7-pt Poisson 256^3 + PCJACOBI; 40 iterations")
    J["cg_jacobi_27pt_160"] = lambda: entry(ref_shim(27, 160, "cg", "jacobi", 40), "reference+shim", "27-pt (bench_kspsolve.c) 160^3, KSPCG + PCJACOBI; 40 iterations")
    J["cg_jacobi_7pt_512"] = lambda: stream("7pt", 512, 512 ** 3, "jacobi", 24, "7-pt Poisson 512^3 (134 M rows), KSPCG + PCJACOBI; 24 iterations")
    J["cg_jacobi_27pt_512"] = lambda: stream("27pt", 512, 512 ** 3, "jacobi", 16, "north_star scaling target: 27-pt 512^3 (3.6e9 nonzeros), KSPCG + PCJACOBI; 16 iterations")
    for g in (1, 2, 4):
        J["cg_none_7pt_1024x1024x%d" % (128 * g)] = (lambda g=g: stream("7pt_box", (1024, 1024, 128 * g), 1024 * 1024 * 128 * g, "none", 12,
                                                                      "BASELINE config 5, weak-scaled share of %d GPU(s): 7-pt 1024 x 1024 x %d, KSPCG + PCNONE; 12 iterations" % (g, 128 * g)))

    def gmres_sor(n, nranks, its):
        ai, aj, aa = orc.stencil("27pt", n)
        b = orc.matmult_mpi(ai, aj, aa, np.ones(n ** 3), nranks)  # b = A * 1 as the drivers form it under mpiexec: MatMult_MPIAIJ (round 5; the products of the solve likewise)
        h = orc.ksp_solve("gmres", ai, aj, aa, b, pc="sor", rtol=1e-50, max_it=its, nranks=nranks, exact=True)[3]
        return entry(h, "oracle-exact", "config 3's solver: 27-pt %d^3, KSPGMRES(30) + PCSOR (local symmetric sweep per rank, %d rank(s): mpiaij.c:1408-1412; products as "
                                        "MatMult_MPIAIJ forms them: diagonal block, then the off-diagonal terms added, mpiaij.c:1056-1059); %d iterations" % (n, nranks, its))
    for g in (1, 2, 4, 8):
        J["gmres_sor_27pt_256_np%d" % g] = (lambda g=g: gmres_sor(256, g, 35))
        J["gmres_sor_27pt_128_np%d" % g] = (lambda g=g: gmres_sor(128, g, 35))
    # round 4: the one-rank histories of config 3's solver come from the REFERENCE itself (its KSPSolve_GMRES, MatSOR_SeqAIJ, VecMDot_Seq_GEMV) with
    # exact BLAS reductions, not from the restated oracle (the two agree to ~1e-13: tests/test_oracle_exact.py)
    for n in (256, 128):
        J["gmres_sor_27pt_%d_np1" % n] = (lambda n=n: entry(ref_shim(27, n, "gmres", "sor", 35), "reference+shim",
                                                            "config 3's solver: 27-pt %d^3, KSPGMRES(30) + PCSOR, one rank; 35 iterations" % n))

    # round 5: BASELINE config 4's stand-in (tests/surrogates.py flan_surrogate_spd: deterministic) under the REFERENCE's MatLoad + KSPSolve with exact
    # BLAS reductions -- what bench.py's config-4 legs used to run beside themselves (a 1.5 GB file written and read back: tens of seconds per run)
    def flan(pc, its):
        import tempfile
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sys.path.insert(0, ROOT)
        from surrogates import flan_surrogate_spd
        from petsc_amd import matio
        d = tempfile.mkdtemp(prefix="hipx_flan_")
        f = os.path.join(d, "matrix.bin")
        try:
            matio.write_petsc_binary(f, *flan_surrogate_spd())
            env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1", LD_PRELOAD=SHIM)
            a = [REF, "-f", f, "-ksp_type", "cg", "-pc_type", pc, "-ksp_rtol", "1e-50", "-ksp_max_it", str(its), "-ksp_norm_type", "preconditioned", "-history", "-mat_type", "aij", "-vec_type", "standard"]
            out = subprocess.run(a, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=7200).stdout
            h = np.array([float(l.split()[2]) for l in out.splitlines() if l.startswith("hist ")])
            assert len(h) == its + 1, out[-2000:]
        finally:
            import shutil
            shutil.rmtree(d, ignore_errors=True)
        return entry(h, "reference+shim", "BASELINE config 4's stand-in (tests/surrogates.py flan_surrogate_spd(): 1,536,000 rows, 121 M nonzeros, 3 unknowns per node = a matrix with inodes), "
                                          "the reference's MatLoad + KSPCG + PC%s (MatMult_SeqAIJ_Inode%s); %d iterations" % (pc.upper(), " / MatSOR_SeqAIJ_Inode" if pc == "sor" else "", its))
    J["cg_jacobi_flan_standin"] = lambda: flan("jacobi", 20)
    J["cg_sor_flan_standin"] = lambda: flan("sor", 10)

    # round 5: BASELINE config 3 at its real shape -- 27-pt 512^3 over 8 (4, 2) ranks: 3.6e9 nonzeros, streamed (oracle/stream_gmres.py: the C oracle's
    # GMRES loop with exact reductions over per-rank products and local sweeps assembled on the fly)
    def gmres_sor_stream(n, nranks, its):
        import stream_gmres
        op = stream_gmres.StreamPartitionedOperator("27pt", n, nranks, sub_rows=1 << 20, log=lambda *a: print("   ", *a, flush=True))
        h = stream_gmres.gmres_sor_exact(op, its, log=lambda *a: print("   ", *a, flush=True))
        return entry(h, "stream", "BASELINE config 3: 27-pt %d^3 (%.2e rows), KSPGMRES(30) + PCSOR on %d ranks (local symmetric sweep per rank, MatMult_MPIAIJ products); %d iterations"
                     % (n, float(n) ** 3, nranks, its))
    for g in (8, 4):  # (2 ranks: a rank's diagonal block alone has 1.8e9 nonzeros, beyond the oracle's 32-bit counts)
        J["gmres_sor_27pt_512_np%d" % g] = (lambda g=g: gmres_sor_stream(512, g, 35 if g == 8 else 16))
    return J


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*")
    a = ap.parse_args()
    data = json.load(open(OUT)) if os.path.exists(OUT) else {}
    J = jobs()
    for k in (a.only or list(J)):
        if k in data and not a.only:
            continue
        t0 = time.time()
        data[k] = J[k]()
        data[k]["seconds_to_make"] = round(time.time() - t0, 1)
        print("%-34s %3d entries  %.1f s   last %.17g" % (k, len(data[k]["history"]), time.time() - t0, data[k]["history"][-1]), flush=True)
        json.dump(data, open(OUT + ".tmp", "w"), indent=1)
        os.replace(OUT + ".tmp", OUT)


if __name__ == "__main__":
    main()
