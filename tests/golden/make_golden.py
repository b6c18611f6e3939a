"""Regenerates tests/golden/*.json.  Run in the build container (needs /root/reference):
      python tests/golden/make_golden.py
  * ref_outputs.json  -- known-answer outputs of the reference's own tests for this path, parsed from
    /root/reference/src/ksp/ksp/tutorials/output/{ex2_3,bench_kspsolve_matmult,bench_kspsolve_ksp}.out
    (residual histories / iteration counts / error norms printed with %g, i.e. 6 significant digits).
  * ref_runs.json     -- outputs of the reference library itself (oracle/_ref, built by oracle/build_ref.py)
    run here with -ksp_monitor at full precision, when oracle/_ref exists (see oracle/README.md).
"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/ksp/ksp/tutorials/output"


def parse_monitor(path):
    hist, tail = [], None
    for line in open(path):
        m = re.match(r"\s*(\d+) KSP Residual norm (\S+)", line)
        if m:
            hist.append(float(m.group(2)))
        m = re.match(r"Norm of error (\S+) iterations (\d+)", line)
        if m:
            tail = {"error": float(m.group(1)), "iterations": int(m.group(2))}
    return hist, tail


def main():
    out = {}
    # ex2 suffix 3: -pc_type sor -pc_sor_symmetric -ksp_monitor -ksp_gmres_cgs_refinement_type refine_always (ex2.c TEST block),
    # default m = 8, n = 7, GMRES(30), rtol = 1e-2/((m+1)(n+1))
    h, t = parse_monitor(os.path.join(REF, "ex2_3.out"))
    out["ex2_3"] = {"args": "-pc_type sor -pc_sor_symmetric -ksp_monitor -ksp_gmres_cgs_refinement_type refine_always", "m": 8, "n": 7, "history": h, **t}
    for name in ("bench_kspsolve_matmult", "bench_kspsolve_ksp"):
        txt = open(os.path.join(REF, name + ".out")).read()
        out[name] = {"dofs": int(re.search(r"DoFs = (\d+)", txt).group(1)), "nnz": int(re.search(r"Number of nonzeros = (\d+)", txt).group(1)),
                     "n": int(re.search(r"-n (\d+)", txt).group(1))}
        m = re.search(r"Error norm:\s+(\S+)", txt)
        if m:
            out[name]["error_norm"] = float(m.group(1))
        m = re.search(r"KSP iters:\s+(\d+)", txt)
        if m:
            out[name]["iterations"] = int(m.group(1))
    json.dump(out, open(os.path.join(HERE, "ref_outputs.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
    kats()
    ref_runs()


# The reference's own known-answer tests for this path: (executable built by oracle/build_ref.py, args, the filter of the
# reference's TEST block restated as python, golden file).  The hipx run appends -vec_type hipx / replaces the mat type.
KATS = [
    ("vec_tut_ex1", "kat_vec_tut_ex1", "", "none", "vec/vec/tutorials/output/ex1_1.out"),
    ("vec_ex43", "kat_vec_ex43", "", "none", "vec/vec/tests/output/ex43_1.out"),
    ("vec_ex34", "kat_vec_ex34", "", "none", "vec/vec/tests/output/ex34_1.out"),
    ("vec_ex60", "kat_vec_ex60", "", "seqname", "vec/vec/tests/output/ex60_1.out"),
    ("vec_ex21", "kat_vec_ex21", "", "ex21", "vec/vec/tests/output/ex21_1.out"),
    ("vec_ex28", "kat_vec_ex28", "", "none", "vec/vec/tests/output/empty.out"),
    ("vec_ex31", "kat_vec_ex31", "", "none", "vec/vec/tests/output/empty.out"),
    ("vec_ex52", "kat_vec_ex52", "", "none", "vec/vec/tests/output/empty.out"),
    ("vec_ex63", "kat_vec_ex63", "", "none", "vec/vec/tests/output/empty.out"),
    ("mat_ex5_11_A", "kat_mat_ex5", "-mat_type seqaij -rectA", "notype", "mat/tests/output/ex5_11_A.out"),
    ("mat_ex5_11_B", "kat_mat_ex5", "-mat_type seqaij -rectB", "notype", "mat/tests/output/ex5_11_B.out"),
    ("mat_ex5_21", "kat_mat_ex5", "-mat_type mpiaij", "notype", "mat/tests/output/ex5_21.out"),
    ("mat_ex5_31", "kat_mat_ex5", "-mat_type mpiaij -test_diagonalscale", "notype", "mat/tests/output/ex5_31.out"),
    # the reference's PetscSF tests with the PetscSF type hipx (sfhipx.c).  ex1 / ex4: host buffers -> the parent's path
    # under the new type; ex2 is the reference's own device test (a VecScatter out of a device vector): -vec_hipx_memtype hands the
    # SF the device mirror, the type hipx runs it on the device
    ("sf_ex1_basic_1", "kat_sf_ex1", "-user_sf_type hipx -sf_type hipx -options_left no", "notype", "vec/is/sf/tests/output/ex1_basic_1.out"),
    ("sf_ex4_1", "kat_sf_ex4", "-sf_type hipx -options_left no", "notype", "vec/is/sf/tests/output/ex4_1.out"),
    ("sf_ex2_device", "kat_sf_ex2", "-vec_hipx_memtype -sf_type hipx", "none", "vec/is/sf/tests/output/ex2_1.out"),
    # src/ksp/pc/tests/ex3.c, suffix sor_aij ("SOR / Richardson-SOR"): GMRES + symmetric PCSOR on the 1-D Laplacian; the vectors
    # are VecCreateSeq's (host): MatMult / MatSOR of the hipx matrix stage them.  The golden was written with %g
    ("pc_ex3_sor_aij", "kat_pc_ex3", "-ksp_type gmres -ksp_monitor -pc_type sor -pc_sor_symmetric -mat_type seqaij -options_left no", "monitor", "ksp/pc/tests/output/ex3_1.out"),
] + [("mat_ex123_1_%s_l%d_n%d" % (mt, la, ng), "kat_mat_ex123", "-mat_type %s -localapi %d -neg %d -options_left no" % (mt, la, ng), "ex123", "mat/tests/output/ex123_1.out")
     for mt in ("seqaij", "mpiaij") for la in (0, 1) for ng in (0, 1)]


# np > 1 variants of the same tests (TEST blocks: `nsize: N`); run with mpiexec against oracle/_ref/mpich (MPICH build of the
# same reference sources).  ex2 suffix 2 prints through -ksp_monitor (%g in the golden, 14 digits in today's reference).
KATS_MPI = [
    ("vec_tut_ex1_np2", "kat_vec_tut_ex1", 2, "", "none", "vec/vec/tutorials/output/ex1_1.out"),
    ("vec_ex21_np2", "kat_vec_ex21", 2, "", "type", "vec/vec/tests/output/ex21_2.out"),
    ("vec_ex28_np3", "kat_vec_ex28", 3, "", "none", "vec/vec/tests/output/empty.out"),
    ("vec_ex28_np3_async", "kat_vec_ex28", 3, "-splitreduction_async -options_left no", "none", "vec/vec/tests/output/empty.out"),
    ("vec_ex31_np2", "kat_vec_ex31", 2, "", "none", "vec/vec/tests/output/empty.out"),
    ("vec_ex52_np2", "kat_vec_ex52", 2, "", "none", "vec/vec/tests/output/empty.out"),
    ("mat_ex5_23", "kat_mat_ex5", 3, "-mat_type mpiaij", "notype", "mat/tests/output/ex5_23.out"),
    ("mat_ex5_33", "kat_mat_ex5", 3, "-mat_type mpiaij -test_diagonalscale", "notype", "mat/tests/output/ex5_33.out"),
    ("ksp_ex2_2", "ex2", 2, "-ksp_monitor -m 5 -n 5 -ksp_gmres_cgs_refinement_type refine_always", "monitor", "ksp/ksp/tutorials/output/ex2_2.out"),
    ("sf_ex1_basic_2", "kat_sf_ex1", 2, "-user_sf_type hipx -sf_type hipx -options_left no", "type", "vec/is/sf/tests/output/ex1_basic_2.out"),
    ("sf_ex1_basic_3", "kat_sf_ex1", 3, "-user_sf_type hipx -sf_type hipx -options_left no", "type", "vec/is/sf/tests/output/ex1_basic_3.out"),
] + [("mat_ex123_3_l%d_n%d" % (la, ng), "kat_mat_ex123", 3, "-mat_type mpiaij -loc -localapi %d -neg %d -options_left no" % (la, ng), "ex123", "mat/tests/output/ex123_3.out") for la in (0, 1) for ng in (0, 1)
] + [  # rectangular local blocks, every entry in the off-diagonal block (suite 4, nsize 4)
    ("mat_ex123_4_l%d_n%d" % (la, ng), "kat_mat_ex123", 4, "-mat_type mpiaij -loc -locdiag 0 -localapi %d -neg %d -options_left no" % (la, ng), "ex123", "mat/tests/output/ex123_4.out") for la in (0, 1) for ng in (0, 1)]


def kats():
    out = {}
    for name, exe, np_, args, filt, gold in KATS_MPI:
        out[name] = {"exe": exe, "nsize": np_, "args": args, "filter": filt, "golden_file": "src/" + gold, "golden": open(os.path.join("/root/reference/src", gold)).read()}
    json.dump(out, open(os.path.join(HERE, "kats_mpi.json"), "w"), indent=0)
    print("kats_mpi.json:", sorted(out))
    out = {}
    for name, exe, args, filt, gold in KATS:
        out[name] = {"exe": exe, "args": args, "filter": filt, "golden_file": "src/" + gold, "golden": open(os.path.join("/root/reference/src", gold)).read()}
    json.dump(out, open(os.path.join(HERE, "kats.json"), "w"), indent=0)
    print("kats.json:", sorted(out))


RUNS = {
    # name: (driver args)  -- all with b = A*1, x0 = 0 (oracle/ref_driver.c)
    "config1_ex2_100x100_cg_jacobi": "-stencil 5 -m 100 -n 100 -ksp_type cg -pc_type jacobi -ksp_rtol 9.8029604940692086e-07",
    "p7_n20_cg_jacobi": "-stencil 7 -n 20 -ksp_type cg -pc_type jacobi -ksp_rtol 1e-8",
    "p7_n20_cg_none_unpre": "-stencil 7 -n 20 -ksp_type cg -pc_type none -ksp_norm_type unpreconditioned -ksp_rtol 1e-8",
    "p7_n20_cg_jacobi_natural": "-stencil 7 -n 20 -ksp_type cg -pc_type jacobi -ksp_norm_type natural -ksp_rtol 1e-8",
    "p27_n16_cg_jacobi": "-stencil 27 -n 16 -ksp_type cg -pc_type jacobi -ksp_rtol 1e-8",
    "p27_n12_gmres_jacobi": "-stencil 27 -n 12 -ksp_type gmres -pc_type jacobi -ksp_rtol 1e-8",
    "p27_n12_gmres5_jacobi": "-stencil 27 -n 12 -ksp_type gmres -ksp_gmres_restart 5 -pc_type jacobi -ksp_rtol 1e-8",
    "p7_n16_gmres_sor": "-stencil 7 -n 16 -ksp_type gmres -pc_type sor -ksp_rtol 1e-8",
    "p27_n12_gmres_sor": "-stencil 27 -n 12 -ksp_type gmres -pc_type sor -ksp_rtol 1e-8",
    "p7_n16_cg_ssor": "-stencil 7 -n 16 -ksp_type cg -pc_type sor -ksp_rtol 1e-8",
    # round 6: the pipelined CG variants (pipecg.c, groppcg.c) -- every norm type, and the loop bound (pipecg.c:160 `i <= max_it`)
    "p7_n20_pipecg_jacobi": "-stencil 7 -n 20 -ksp_type pipecg -pc_type jacobi -ksp_rtol 1e-8",
    "p7_n20_pipecg_none_unpre": "-stencil 7 -n 20 -ksp_type pipecg -pc_type none -ksp_norm_type unpreconditioned -ksp_rtol 1e-8",
    "p7_n20_pipecg_jacobi_natural": "-stencil 7 -n 20 -ksp_type pipecg -pc_type jacobi -ksp_norm_type natural -ksp_rtol 1e-8",
    "p27_n16_pipecg_jacobi": "-stencil 27 -n 16 -ksp_type pipecg -pc_type jacobi -ksp_rtol 1e-8",
    "p7_n20_pipecg_jacobi_maxit7": "-stencil 7 -n 20 -ksp_type pipecg -pc_type jacobi -ksp_rtol 1e-30 -ksp_max_it 7",
    "ex2_100x100_pipecg_jacobi": "-stencil 5 -m 100 -n 100 -ksp_type pipecg -pc_type jacobi -ksp_rtol 1e-6",
    "p7_n20_groppcg_jacobi": "-stencil 7 -n 20 -ksp_type groppcg -pc_type jacobi -ksp_rtol 1e-8",
    "p7_n20_groppcg_none_unpre": "-stencil 7 -n 20 -ksp_type groppcg -pc_type none -ksp_norm_type unpreconditioned -ksp_rtol 1e-8",
    "p7_n20_groppcg_jacobi_natural": "-stencil 7 -n 20 -ksp_type groppcg -pc_type jacobi -ksp_norm_type natural -ksp_rtol 1e-8",
    "p27_n16_groppcg_jacobi": "-stencil 27 -n 16 -ksp_type groppcg -pc_type jacobi -ksp_rtol 1e-8",
    "p7_n20_groppcg_jacobi_maxit7": "-stencil 7 -n 20 -ksp_type groppcg -pc_type jacobi -ksp_rtol 1e-30 -ksp_max_it 7",
    "ex2_3_gmres_ssor": "-stencil 5 -m 8 -n 7 -pc_type sor -pc_sor_symmetric -ksp_gmres_cgs_refinement_type refine_always -ksp_rtol 1.3888888888888889e-04",
}
SPMV = {"p7_n8": "-stencil 7 -n 8", "p27_n6": "-stencil 27 -n 6", "p5_9x7": "-stencil 5 -m 9 -n 7"}


def ref_runs():
    """Outputs of the reference library itself (oracle/_ref, built by oracle/build_ref.py) at full precision."""
    import subprocess
    exe = os.path.join(HERE, "..", "..", "oracle", "_ref", "bin", "ref_driver")
    if not os.path.exists(exe):
        print("oracle/_ref not built: ref_runs.json not regenerated")
        return
    out = {"_how": "oracle/_ref/bin/ref_driver <args> -history (reference libpetsc built by oracle/build_ref.py: gcc -O2, MPIUNI, libmkl_rt)", "ksp": {}, "spmv": {}}
    for name, args in RUNS.items():
        txt = subprocess.check_output([exe] + args.split() + ["-history"], text=True)
        hist = [float(l.split()[2]) for l in txt.splitlines() if l.startswith("hist ")]
        m = re.search(r"iterations (\d+) reason (-?\d+) error (\S+)", txt)
        out["ksp"][name] = {"args": args, "iterations": int(m.group(1)), "reason": int(m.group(2)), "error": float(m.group(3)), "history": [repr(h) for h in hist]}
    for name, args in SPMV.items():
        txt = subprocess.check_output([exe] + args.split() + ["-dump_y", "-ksp_max_it", "1"], text=True)
        out["spmv"][name] = {"args": args, "y": [l.split()[2] for l in txt.splitlines() if l.startswith("y ")]}
    json.dump(out, open(os.path.join(HERE, "ref_runs.json"), "w"), indent=0)
    print("ref_runs.json:", {k: v["iterations"] for k, v in out["ksp"].items()})


if __name__ == "__main__":
    main()
