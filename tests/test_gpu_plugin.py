"""GPU: the drop-in itself.  UNMODIFIED reference executables (the reference's tutorials ex2 and bench_kspsolve and our
thin ref_driver, all linked against the reference's libpetsc built by oracle/build_ref.py) are run with
    -dll_prepend petsc_amd/lib/libpetschipx.so -vec_type hipx -mat_type aijhipx
and must reproduce the reference's golden outputs and the CPU run of the same binary."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "bin")
PLUGIN = os.path.join(ROOT, "petsc_amd", "lib", "libpetschipx.so")
HIPX = ["-dll_prepend", PLUGIN, "-vec_type", "hipx", "-mat_type", "aijhipx"]


def run(exe, args, exact_blas=False):
    p = os.path.join(BIN, exe)
    assert os.path.exists(p) and os.path.exists(PLUGIN), "oracle/_ref or the plugin is not built: run __graft_entry__.build() where /root/reference exists"
    env = dict(os.environ, HIPX_NO_TORCH="1")
    if exact_blas:  # the CPU run with the published (unfused) daxpy and twice-working-precision reductions: oracle/exactblas.c
        env["LD_PRELOAD"] = os.path.join(ROOT, "oracle", "libexactblas.so")
    r = subprocess.run([p] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


def hist_of(txt):
    return np.array([float(l.split()[2]) for l in txt.splitlines() if l.startswith("hist ")])


def test_ex2_golden_suffix3_with_hipx_types():
    """output/ex2_3.out (GMRES + symmetric SOR, 8x7 grid): the reference's harness compares the %g rendering of the
    monitor lines (petscdiff); reproduce every printed digit."""
    args = ["-pc_type", "sor", "-pc_sor_symmetric", "-ksp_monitor", "-ksp_gmres_cgs_refinement_type", "refine_always"]
    out = run("ex2", args + HIPX)
    golden = ["2.98499", "1.13133", "0.575925", "0.108871", "0.0213225", "0.00325239", "0.000874208", "0.000179613"]
    got = ["%g" % float(m) for m in re.findall(r"KSP Residual norm (\S+)", out)]
    assert got == golden
    assert "Norm of error 0.000300302 iterations 7" in out
    cpu = run("ex2", args)
    assert [float(m) for m in re.findall(r"KSP Residual norm (\S+)", cpu)] == pytest.approx([float(m) for m in re.findall(r"KSP Residual norm (\S+)", out)], rel=1e-11)


def test_ex2_config1_cg_jacobi_and_jacobihipx():
    """BASELINE config 1: 160 iterations, Norm of error 5.70785e-05."""
    base = ["-m", "100", "-n", "100", "-ksp_type", "cg"]
    cpu = run("ex2", base + ["-pc_type", "jacobi"])
    assert cpu.strip() == "Norm of error 5.70785e-05 iterations 160"
    assert run("ex2", base + ["-pc_type", "jacobi"] + HIPX) == cpu
    assert run("ex2", base + ["-pc_type", "jacobihipx"] + HIPX) == cpu


def test_view_reports_hipx_types_and_kernels_ran():
    out = run("ex2", ["-m", "20", "-n", "20", "-ksp_type", "cg", "-pc_type", "jacobi", "-ksp_view"] + HIPX)
    assert "type: seqaijhipx" in out
    out = run("ex2", ["-m", "20", "-n", "20", "-ksp_type", "cg", "-pc_type", "jacobi", "-vec_view", "::ascii_info"] + HIPX)
    assert "error" not in out.lower().replace("norm of error", "")


# (driver args, number of leading history entries compared, pointwise relative tolerance).  Krylov recurrences amplify the
# rounding of the (non-bit-exact) reductions; well-conditioned runs are compared over their whole history, the two
# notoriously sensitive ones (short-restart GMRES over 200+ iterations, BiCGStab) over their first entries only --
# the reference's own CPU path shows the same sensitivity (oracle vs reference differ by 1.6e-3 there, and a 1e-15
# relative perturbation of b changes the iteration count).
CASES = [("-stencil 7 -n 24 -ksp_type cg -pc_type jacobi -ksp_rtol 1e-8", None, 1e-8),
         ("-stencil 27 -n 16 -ksp_type cg -pc_type jacobi -ksp_rtol 1e-8", None, 1e-8),
         ("-stencil 27 -n 14 -ksp_type gmres -pc_type sor -ksp_rtol 1e-8", None, 1e-8),
         ("-stencil 7 -n 20 -ksp_type gmres -pc_type jacobi -ksp_rtol 1e-8", None, 1e-8),
         ("-stencil 7 -n 16 -ksp_type cg -pc_type sor -ksp_rtol 1e-8", None, 1e-8),
         ("-stencil 7 -n 16 -ksp_type cg -pc_type sor -ksp_rtol 1e-8 -dup_mat", None, 1e-8),  # operator = MatDuplicate(A)
         ("-stencil 7 -n 16 -ksp_type cg -pc_type bjacobi -sub_pc_type sor -ksp_rtol 1e-8", None, 1e-8),  # local-vector views
         # PCApply_BJacobi_Multiblock (bjacobi.c:886-895): VecPlaceArray on hipx work vectors, sub-solve on the device, VecResetArray
         ("-stencil 7 -n 16 -ksp_type cg -pc_type bjacobi -pc_bjacobi_local_blocks 2 -sub_pc_type sor -ksp_rtol 1e-8", None, 1e-8),
         ("-stencil 7 -n 16 -ksp_type gmres -pc_type bjacobi -pc_bjacobi_local_blocks 3 -sub_pc_type jacobi -ksp_rtol 1e-8", None, 1e-6),  # (absolute 1e-12 |r0|; measured 3e-8 relative at the 1e-9 tail)
         ("-stencil 7 -n 16 -ksp_type cg -ksp_cg_single_reduction -pc_type jacobi -ksp_rtol 1e-8", None, 1e-8),
         # SURVEY 8(f2)/(f4) callers that only need the Vec/Mat ops: pipelined and Gropp CG, Chebyshev with fixed bounds, PCPBJACOBI
         ("-stencil 7 -n 16 -ksp_type pipecg -pc_type jacobi -ksp_rtol 1e-8", 25, 1e-9),
         ("-stencil 7 -n 16 -ksp_type groppcg -pc_type jacobi -ksp_rtol 1e-8", 25, 1e-9),
         ("-stencil 7 -n 16 -ksp_type chebyshev -ksp_chebyshev_eigenvalues 0.2,2.0 -pc_type jacobi -ksp_rtol 1e-6 -ksp_max_it 400", None, 1e-8),
         ("-stencil 27 -n 12 -ksp_type cg -pc_type pbjacobi -ksp_rtol 1e-8", None, 1e-8),
         ("-stencil 7 -n 16 -ksp_type fgmres -pc_type jacobi -ksp_rtol 1e-8", None, 1e-6),
         ("-stencil 7 -n 16 -ksp_type cr -pc_type jacobi -ksp_rtol 1e-8", None, 1e-7),
         ("-stencil 7 -n 20 -ksp_type gmres -ksp_gmres_restart 7 -pc_type jacobi -ksp_rtol 1e-8", 40, 1e-9),
         ("-stencil 7 -n 16 -ksp_type bcgs -pc_type jacobi -ksp_rtol 1e-8", 12, 1e-8)]


@pytest.mark.parametrize("args,nlead,tol", CASES)
def test_ref_driver_histories_cpu_vs_hipx(args, nlead, tol):
    """Any KSP of the reference runs on the HIPX types unchanged (they only call Vec/Mat ops): CG, single-reduction CG,
    CR, GMRES, FGMRES, BiCGStab ..."""
    a = args.split() + ["-history"]
    cpu, gpu = run("ref_driver", a), run("ref_driver", a + HIPX)
    hc, hg = hist_of(cpu), hist_of(gpu)
    ic = re.search(r"iterations (\d+) reason (-?\d+) error (\S+)", cpu)
    ig = re.search(r"iterations (\d+) reason (-?\d+) error (\S+)", gpu)
    if nlead is None:
        assert ic.group(1, 2) == ig.group(1, 2)
        assert len(hc) == len(hg)
        assert np.abs(hc - hg).max() <= 1e-12 * hc[0]
        assert abs(float(ic.group(3)) - float(ig.group(3))) <= 1e-5 * float(ic.group(3)) + 1e-13
    else:
        assert ic.group(2) == ig.group(2) and abs(int(ic.group(1)) - int(ig.group(1))) <= max(3, int(ic.group(1)) // 20)
        hc, hg = hc[:nlead], hg[:nlead]
    assert (np.abs(hc - hg) / hc).max() <= tol


def test_matmult_bit_exact_cpu_vs_hipx():
    a = "-stencil 27 -n 10 -dump_y -ksp_max_it 1".split()
    cpu, gpu = run("ref_driver", a), run("ref_driver", a + HIPX)
    yc = [l for l in cpu.splitlines() if l.startswith("y ")]
    yg = [l for l in gpu.splitlines() if l.startswith("y ")]
    assert yc == yg and len(yc) == 1000


@pytest.mark.parametrize("args", ["-stencil 27 -n 10 -mat_ops", "-stencil 7 -n 12 -mat_ops", "-stencil 5 -m 30 -n 17 -mat_ops", "-stencil 27 -n 10"])
def test_matmulttranspose_on_the_device_bit_exact(args):
    """MatMultTranspose_SeqAIJ / MatMultTransposeAdd_SeqAIJ (aij.c:1383-1440) through MATSEQAIJHIPX (round 4): the transposed matrix as its own
    device CSR -- column c's contributions in ascending row order, the reference's order -- so A^T x, y + A^T x (separate and in place) carry the
    CPU loop's bits, also for the non-symmetric D_l A D_r."""
    a = args.split() + ["-dump_y", "-dump_yt", "-ksp_max_it", "1"]
    cpu, gpu = run("ref_driver", a), run("ref_driver", a + HIPX)
    for tag in ("yt ", "yta "):
        c = [l for l in cpu.splitlines() if l.startswith(tag)]
        g = [l for l in gpu.splitlines() if l.startswith(tag)]
        assert len(c) > 0 and c == g, tag
    nc = [float(l.split()[-1]) for l in cpu.splitlines() if l.startswith("MatMultTransposeAdd in place")]
    ng = [float(l.split()[-1]) for l in gpu.splitlines() if l.startswith("MatMultTransposeAdd in place")]
    assert len(nc) == 1 and abs(nc[0] - ng[0]) <= 1e-14 * nc[0]  # (a norm: the reduction's rounding, not the product's)


def test_matscale_diagonalscale_on_device_then_host_update_bit_exact():
    """MatScale / MatDiagonalScale run on the device copy (hipxMatScale, hipxMatDiagonalScale: (a l_i) r_j like
    aij.c:2333-2371), a host-side MatSetValue + assembly and another MatDiagonalScale follow: the product is bit-identical to the
    CPU types', and so is the solve's history."""
    a = "-stencil 27 -n 10 -mat_ops -dump_y -ksp_type gmres -pc_type jacobi -ksp_rtol 1e-8 -history".split()
    cpu, gpu = run("ref_driver", a), run("ref_driver", a + HIPX)
    yc = [l for l in cpu.splitlines() if l.startswith("y ")]
    yg = [l for l in gpu.splitlines() if l.startswith("y ")]
    assert yc == yg and len(yc) == 1000
    hc, hg = hist_of(cpu), hist_of(gpu)
    assert len(hc) == len(hg) and np.abs(hc - hg).max() <= 1e-12 * hc[0]


def test_device_value_op_then_in_place_host_edit_is_uploaded():
    """After a device-side MatDiagonalScale the host edits a value in place (MatSeqAIJGetArray / RestoreArray): the next product
    must see it (the device copy is re-uploaded, never trusted on a predicted object state)."""
    a = "-stencil 7 -n 9 -mat_ops -mat_ops_block_edit -dump_y -ksp_max_it 1".split()
    cpu, gpu = run("ref_driver", a), run("ref_driver", a + HIPX)
    yc = [l for l in cpu.splitlines() if l.startswith("y ")]
    yg = [l for l in gpu.splitlines() if l.startswith("y ")]
    assert yc == yg and len(yc) == 729


def test_bench_kspsolve_matmult_and_ksp_goldens_with_aijhipx():
    out = run("bench_kspsolve", ["-print_timing", "false", "-matmult", "-its", "10", "-n", "8", "-mat_type", "aijhipx", "-dll_prepend", PLUGIN])
    ref = run("bench_kspsolve", ["-print_timing", "false", "-matmult", "-its", "10", "-n", "8"])
    assert out.split() == ref.split() and "Number of nonzeros = 10648" in out


def test_ksp_cghipx_fused_solve_inside_petsc():
    """-ksp_type cghipx: KSPCG subclass whose solve runs the fused device kernels (C host layer) under PETSc's own monitors,
    history and convergence test.  Must agree with the reference's KSPSolve_CG on the CPU and on the hipx types."""
    base = ["-m", "100", "-n", "100", "-pc_type", "jacobi"]
    cpu = run("ex2", base + ["-ksp_type", "cg"])
    assert cpu.strip() == "Norm of error 5.70785e-05 iterations 160"          # BASELINE config 1
    assert run("ex2", base + ["-ksp_type", "cghipx"] + HIPX) == cpu
    # monitors / converged reason come from PETSc itself, every iteration
    mon_cpu = run("ex2", base + ["-ksp_type", "cg", "-ksp_monitor", "-ksp_converged_reason"])
    mon_gpu = run("ex2", base + ["-ksp_type", "cghipx", "-ksp_monitor", "-ksp_converged_reason"] + HIPX)
    rc = [float(m) for m in re.findall(r"KSP Residual norm (\S+)", mon_cpu)]
    rg = [float(m) for m in re.findall(r"KSP Residual norm (\S+)", mon_gpu)]
    assert len(rc) == len(rg) == 161 and np.abs(np.array(rc) - np.array(rg)).max() <= 1e-12 * rc[0]
    assert "CONVERGED_RTOL iterations 160" in mon_gpu and "CONVERGED_RTOL iterations 160" in mon_cpu
    # full-precision histories through ref_driver (3-D 7-pt and 27-pt), max_it stop, non-default configuration falls back
    for args in ("-stencil 7 -n 24 -pc_type jacobi -ksp_rtol 1e-8", "-stencil 27 -n 16 -pc_type jacobi -ksp_rtol 1e-8",
                 "-stencil 7 -n 16 -pc_type jacobi -ksp_rtol 1e-30 -ksp_max_it 9", "-stencil 7 -n 16 -pc_type sor -ksp_rtol 1e-8",
                 "-stencil 7 -n 16 -pc_type jacobi -ksp_norm_type unpreconditioned -ksp_rtol 1e-8"):
        a = args.split() + ["-history"]
        c, g = run("ref_driver", a + ["-ksp_type", "cg"]), run("ref_driver", a + ["-ksp_type", "cghipx"] + HIPX)
        hc, hg = hist_of(c), hist_of(g)
        ic = re.search(r"iterations (\d+) reason (-?\d+) error (\S+)", c)
        ig = re.search(r"iterations (\d+) reason (-?\d+) error (\S+)", g)
        assert ic.group(1, 2) == ig.group(1, 2), (args, ic.groups(), ig.groups())
        assert len(hc) == len(hg) and np.abs(hc - hg).max() <= 1e-12 * hc[0], args
        assert abs(float(ic.group(3)) - float(ig.group(3))) <= 1e-5 * float(ic.group(3)) + 1e-13


def test_pc_eisenstat_on_hipx_types():
    """PCEISENSTAT (src/ksp/pc/impls/eisens/eisen.c) drives MatSOR with SOR_EISENSTAT and SOR_APPLY_UPPER: the CG history of the
    reference over the hipx types follows its CPU run."""
    a = "-stencil 7 -n 20 -ksp_type cg -pc_type eisenstat -ksp_rtol 1e-8 -history".split()
    c, g = run("ref_driver", a), run("ref_driver", a + HIPX)
    hc, hg = hist_of(c), hist_of(g)
    ic = re.search(r"iterations (\d+) reason (-?\d+) error (\S+)", c)
    ig = re.search(r"iterations (\d+) reason (-?\d+) error (\S+)", g)
    assert ic.group(1, 2) == ig.group(1, 2) and len(hc) == len(hg) > 5
    assert (np.abs(hc - hg) / hc).max() <= 1e-9


@pytest.mark.parametrize("bs,n", [(2, 8), (3, 9), (4, 8), (8, 8)])
def test_pbjacobi_apply_on_device_bit_exact(bs, n):
    """-pc_type pbjacobihipx = PCPBJACOBI (host set-up: MatInvertBlockDiagonal) with PCApply / PCApplyTranspose as one
    device kernel (pbjacobi.c:4-124,126-241: column-major inverted blocks, products added left to right): z = PCApply(b) and
    z' = PCApplyTranspose(b) bit-identical to the CPU types' PCPBJACOBI, and the preconditioned solve follows it."""
    a = ("-stencil 27 -n %d -mat_block_size %d -dump_pc -ksp_type gmres -ksp_rtol 1e-8 -history" % (n, bs)).split()
    cpu = run("ref_driver", a + ["-pc_type", "pbjacobi"])
    gpu = run("ref_driver", a + HIPX + ["-pc_type", "pbjacobihipx"])
    for tag in ("z ", "zt "):
        zc = [l for l in cpu.splitlines() if l.startswith(tag)]
        zg = [l for l in gpu.splitlines() if l.startswith(tag)]
        assert zc == zg and len(zc) == n ** 3
    hc, hg = hist_of(cpu), hist_of(gpu)
    assert len(hc) == len(hg) > 3 and np.abs(hc - hg).max() <= 1e-10 * hc[0]


def test_mataxpy_same_pattern_on_device_bit_exact():
    """MatAXPY(Y, a, X, SAME_NONZERO_PATTERN) = a daxpy over the value arrays (aij.c:2926-2945) runs on the device
    copies (hipxMatAXPY).  An optimised host BLAS fuses the multiply-add; the published daxpy (and every plain-C loop of the
    reference) rounds twice -- the CPU side therefore runs with oracle/libexactblas.so, which pins that definition: the product
    y = (A + 0.37 D A) x is then bit-identical."""
    a = "-stencil 27 -n 9 -mat_axpy -dump_y -ksp_max_it 1".split()
    cpu, gpu = run("ref_driver", a, exact_blas=True), run("ref_driver", a + HIPX)
    yc = [l for l in cpu.splitlines() if l.startswith("y ")]
    yg = [l for l in gpu.splitlines() if l.startswith("y ")]
    assert yc == yg and len(yc) == 729


def test_matload_of_a_file_into_the_hipx_types(tmp_path):
    """BASELINE config 4's route (a SuiteSparse file -> MatLoad -> KSPCG + PCJACOBI): `ref_driver -f <PETSc binary>` with
    -mat_type aijhipx -vec_type hipx against the CPU types -- a matrix with arbitrary distinct values (no templates, no dictionary):
    y = A x bit-identical, the history within 1e-12 of the CPU run made with exact BLAS reductions."""
    import sys
    sys.path.insert(0, ROOT)
    from petsc_amd import matio
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from surrogates import flan_surrogate_spd
    ai, aj, aa = flan_surrogate_spd(16)
    f = str(tmp_path / "spd.bin")
    matio.write_petsc_binary(f, ai, aj, aa)
    a = ["-f", f, "-ksp_type", "cg", "-pc_type", "jacobi", "-ksp_rtol", "1e-50", "-ksp_max_it", "12", "-ksp_norm_type", "preconditioned", "-history", "-dump_y"]
    # The reference takes its INODE MatMult on this matrix (3 x 3 blocks: consecutive rows share their column pattern; inode.c sums a row
    # two columns at a time: sum += a0 x0 + a1 x1), which rounds differently from MatMult_SeqAIJ's one-by-one sum.  libhipx follows the
    # matrix it wraps (round 4: the partition PETSc found is handed over, hipxMatSetInodes): bit-identical y in BOTH settings, and with exact
    # reductions on both sides the same history to the last bit.
    ys = lambda t: [l.split()[2] for l in t.splitlines() if l.startswith("y ")]  # noqa: E731
    seen = []
    for extra in ([], ["-mat_no_inode"]):
        cpu = run("ref_driver", a + extra, exact_blas=True)
        gpu = run("ref_driver", a + extra + HIPX + ["-hipx_reductions", "exact"])
        y_cpu, y_gpu = ys(cpu), ys(gpu)
        assert len(y_cpu) == len(ai) - 1 and y_gpu == y_cpu, extra
        hc, hg = hist_of(cpu), hist_of(gpu)
        assert len(hc) == 13 and len(hg) == 13
        assert np.array_equal(hc, hg), (extra, max(abs(g - c) / abs(c) for g, c in zip(hg, hc)))
        seen.append(y_cpu)
    assert seen[0] != seen[1]  # (the two orders do differ on this matrix)


def test_matsor_on_a_matrix_with_inodes_is_the_reference_s_node_sweep(tmp_path):
    """A MATSEQAIJ matrix with inodes is relaxed node by node (aij.c:1852 -> MatSOR_SeqAIJ_Inode, inode.c:2494): MatSOR of the aijhipx type
    must print what the CPU type prints -- every sweep kind, with the partition PETSc found at assembly handed to libhipx
    (hipxMatSetInodes), under -mat_no_inode (point sweeps on both sides) and for omega != 1 (the reference falls back to the point
    routine) -- and KSPCG + PCSOR on the 3-unknowns-per-node stand-in must follow the CPU history."""
    import sys
    sys.path.insert(0, ROOT)
    from petsc_amd import matio
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from surrogates import flan_surrogate_spd, inode_matrix
    ai, aj, aa = inode_matrix(nnodes=150, seed=33)
    f = str(tmp_path / "inode.bin")
    matio.write_petsc_binary(f, ai, aj, aa)
    sor = lambda t: [l.split()[2] for l in t.splitlines() if l.startswith("sor ")]  # noqa: E731
    for extra in (["-dump_sor", "28"], ["-dump_sor", "3"], ["-dump_sor", "17", "-sor_its", "2"], ["-dump_sor", "2", "-sor_its", "2"], ["-dump_sor", "32"],
                  ["-dump_sor", "28", "-mat_no_inode"], ["-dump_sor", "28", "-sor_omega", "1.3"], ["-dump_sor", "28", "-mat_inode_limit", "2"]):
        a = ["-f", f, "-ksp_max_it", "1"] + extra
        cpu, gpu = sor(run("ref_driver", a)), sor(run("ref_driver", a + HIPX))
        assert len(cpu) == len(ai) - 1 and cpu == gpu, extra
    ai, aj, aa = flan_surrogate_spd(8)
    f = str(tmp_path / "flan8.bin")
    matio.write_petsc_binary(f, ai, aj, aa)
    a = ["-f", f, "-ksp_type", "cg", "-pc_type", "sor", "-ksp_rtol", "1e-50", "-ksp_max_it", "12", "-ksp_norm_type", "preconditioned", "-history"]
    hc, hg = hist_of(run("ref_driver", a, exact_blas=True)), hist_of(run("ref_driver", a + HIPX + ["-hipx_reductions", "exact"]))
    assert len(hc) == 13 and len(hg) == 13
    assert np.array_equal(hc, hg), max(abs(g - c) / abs(c) for g, c in zip(hg, hc))  # product, relaxation and (exact) reductions: every kernel bit-identical


@pytest.mark.parametrize("args", ["-stencil 7 -n 12 -pc_type jacobi -ksp_max_it 8", "-stencil 27 -n 10 -pc_type jacobi -ksp_max_it 5", "-stencil 5 -m 31 -n 17 -pc_type none -ksp_max_it 12",
                                  "-stencil 7 -n 12 -pc_type jacobi -ksp_max_it 1", "-stencil 7 -n 12 -pc_type jacobi -ksp_max_it 7 -ksp_initial_guess_nonzero"])
def test_ksp_chebyshevhipx_fused_smoother_bit_identical(args):
    """KSPCHEBYSHEV as a smoother (first kind, -ksp_norm_type none, given eigenvalue bounds): `-ksp_type chebyshevhipx` runs every iteration
    as the SpMV plus ONE fused kernel (residual + PCJACOBI / PCNONE + three-term update of cheby.c:475-511) -- no reductions anywhere, so
    the solution is bit-identical to the reference's KSPSolve_Chebyshev on the CPU types, entry by entry; the reference's own chebyshev
    over the hipx types gives the same bits through four kernels per iteration.  With a norm requested the type falls back to the parent."""
    common = args.split() + ["-ksp_chebyshev_eigenvalues", "0.15,1.95", "-dump_x"]
    a = common + ["-ksp_norm_type", "none"]
    xs = lambda t: [l.split()[2] for l in t.splitlines() if l.startswith("x ")]  # noqa: E731
    cpu = run("ref_driver", a + ["-ksp_type", "chebyshev"])
    gpu_ref = run("ref_driver", a + ["-ksp_type", "chebyshev"] + HIPX)
    gpu = run("ref_driver", a + ["-ksp_type", "chebyshevhipx", "-info"] + HIPX)
    assert len(xs(cpu)) > 100 and xs(gpu_ref) == xs(cpu) and xs(gpu) == xs(cpu)
    assert "outside the fused path" not in gpu
    assert re.search(r"iterations (\d+) reason (-?\d+)", gpu).groups() == re.search(r"iterations (\d+) reason (-?\d+)", cpu).groups()
    # a norm type: the parent's solve (monitors, history) over the hipx types
    b = common + ["-ksp_norm_type", "preconditioned", "-history"]
    cpu_n, gpu_n = run("ref_driver", b + ["-ksp_type", "chebyshev"]), run("ref_driver", b + ["-ksp_type", "chebyshevhipx", "-info"] + HIPX)
    assert "outside the fused path" in gpu_n
    hc, hg = hist_of(cpu_n), hist_of(gpu_n)
    assert len(hc) == len(hg) and len(hc) > 1 and (np.abs(hc - hg) / hc).max() <= 1e-12


@pytest.mark.parametrize("args", ["-stencil 7 -n 48 -ksp_type cg -pc_type jacobi -ksp_norm_type preconditioned -ksp_rtol 1e-50 -ksp_max_it 60",
                                  "-stencil 27 -n 24 -ksp_type cg -pc_type jacobihipx -ksp_norm_type natural -ksp_rtol 1e-50 -ksp_max_it 40",
                                  "-stencil 7 -n 32 -ksp_type cg -pc_type jacobi -ksp_norm_type unpreconditioned -ksp_rtol 1e-50 -ksp_max_it 40",
                                  "-stencil 7 -n 24 -ksp_type cg -ksp_cg_single_reduction -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 40",
                                  "-stencil 7 -n 24 -ksp_type gmres -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 45",
                                  "-stencil 7 -n 24 -ksp_type bcgs -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 12"])
def test_reduction_cache_returns_what_the_separate_kernels_return(args):
    """Round 5: the reference's UNMODIFIED Krylov loops over the hipx types get p . A p from the product's epilogue and z . z,
    z . r from the kernel that applies PCJACOBI (VecPointwiseMult), keyed on the vectors (plugin/vechipx.c: reduction cache) -- three vector passes
    and three launches fewer per CG iteration.  A cached sum must be the sum the separate kernel returns on the same data: with EXACT reductions
    the histories with and without the cache are the same doubles; with the default reductions they differ by the association of the partial sums."""
    a = args.split() + ["-history"]
    on = hist_of(run("ref_driver", a + HIPX + ["-hipx_reductions", "exact"]))
    off = hist_of(run("ref_driver", a + HIPX + ["-hipx_reductions", "exact", "-hipx_reduction_cache", "0"]))
    assert len(on) == len(off) > 10 and np.array_equal(on, off), np.abs(on - off).max()
    on_f = hist_of(run("ref_driver", a + HIPX))
    off_f = hist_of(run("ref_driver", a + HIPX + ["-hipx_reduction_cache", "0"]))
    assert len(on_f) == len(off_f) == len(on)
    tol = 1e-12 if "bcgs" not in args and "gmres" not in args else 1e-9
    assert (np.abs(on_f - off_f) / off_f).max() <= tol
    assert (np.abs(on_f - on) / on).max() <= (1e-11 if tol == 1e-12 else 1e-8)


LAZY_CASES = ["-stencil 7 -n 48 -ksp_type cg -pc_type jacobi -ksp_norm_type preconditioned -ksp_rtol 1e-50 -ksp_max_it 60",
              "-stencil 7 -n 64 -ksp_type cg -pc_type jacobi -ksp_norm_type preconditioned -ksp_rtol 1e-50 -ksp_max_it 40 -mat_aijhipx_spmv_variant 30",   # march form: product prologue
              "-stencil 7 -n 48 -ksp_type cg -pc_type jacobi -ksp_norm_type natural -ksp_rtol 1e-50 -ksp_max_it 40 -mat_aijhipx_spmv_variant 30",          # planes of 2304 = 2 x 1024 + 256 rows
              "-stencil 27 -n 24 -ksp_type cg -pc_type jacobihipx -ksp_norm_type natural -ksp_rtol 1e-50 -ksp_max_it 40",
              "-stencil 7 -n 32 -ksp_type cg -pc_type jacobi -ksp_norm_type preconditioned -ksp_rtol 1e-50 -ksp_max_it 30 -mat_axpy",  # a diagonal that is NOT one value: streamed
              "-stencil 7 -n 32 -ksp_type cg -pc_type jacobi -ksp_norm_type natural -ksp_rtol 1e-50 -ksp_max_it 30 -mat_ops",
              "-stencil 7 -n 32 -ksp_type cg -pc_type jacobi -ksp_norm_type unpreconditioned -ksp_rtol 1e-50 -ksp_max_it 40",
              "-stencil 7 -n 32 -ksp_type cg -pc_type none -ksp_rtol 1e-50 -ksp_max_it 40 -mat_aijhipx_spmv_variant 30",
              "-stencil 7 -n 24 -ksp_type cg -pc_type sor -ksp_rtol 1e-50 -ksp_max_it 30",
              "-stencil 7 -n 24 -ksp_type cg -ksp_cg_single_reduction -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 40",
              "-stencil 7 -n 24 -ksp_type cr -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 40",
              "-stencil 7 -n 24 -ksp_type gmres -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 45",
              "-stencil 7 -n 24 -ksp_type bcgs -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 12",
              "-stencil 7 -n 24 -ksp_type chebyshev -pc_type jacobi -ksp_max_it 30 -ksp_norm_type preconditioned -ksp_rtol 1e-50",
              "-stencil 7 -n 24 -ksp_type richardson -pc_type jacobi -ksp_max_it 30 -ksp_rtol 1e-50 -ksp_richardson_scale 0.1",
              # round 6: the pipelined variants -- their update blocks run as ONE batch kernel (hipxVecBatchAXPYDotsBegin), the sums of the next iteration with it
              "-stencil 7 -n 32 -ksp_type pipecg -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 40",
              "-stencil 27 -n 20 -ksp_type pipecg -pc_type none -ksp_norm_type unpreconditioned -ksp_rtol 1e-50 -ksp_max_it 30",
              "-stencil 7 -n 24 -ksp_type pipecg -pc_type jacobi -ksp_norm_type natural -ksp_rtol 1e-50 -ksp_max_it 30 -mat_axpy",
              "-stencil 7 -n 24 -ksp_type pipecg -pc_type sor -ksp_rtol 1e-50 -ksp_max_it 20",
              "-stencil 7 -n 32 -ksp_type groppcg -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 40",
              "-stencil 27 -n 20 -ksp_type groppcg -pc_type none -ksp_norm_type unpreconditioned -ksp_rtol 1e-50 -ksp_max_it 30",
              "-stencil 7 -n 24 -ksp_type groppcg -pc_type jacobi -ksp_norm_type natural -ksp_rtol 1e-50 -ksp_max_it 30",
              "-stencil 7 -n 24 -ksp_type pipecr -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 30"]


@pytest.mark.parametrize("args", LAZY_CASES)
def test_lazy_fusion_leaves_every_history_as_it_was(args):
    """Round 5: VecAXPY / VecAYPX on hipx vectors are recorded and run later -- fused into VecPointwiseMult (PCApply_Jacobi), into the product
    kernel's prologue (march-form matrices; the direction vector's two device buffers swap), or as one direction kernel (plugin/vechipx.c: lazy
    fusion).  Every fused kernel does the separate kernels' operations element by element: with EXACT reductions the residual histories of the
    reference's UNMODIFIED Krylov loops with and without it are the same doubles (the iterates feed every later residual: one wrong or stale element
    would show), for every solver here; with the default reductions they differ by the association of the partial sums only."""
    a = args.split() + ["-history", "-hipx_lazy_min_size", "1"]
    on_out = run("ref_driver", a + HIPX + ["-hipx_reductions", "exact", "-hipx_lazy_view"])
    on = hist_of(on_out)
    off = hist_of(run("ref_driver", a + HIPX + ["-hipx_reductions", "exact", "-hipx_lazy_fusion", "0"]))
    assert len(on) == len(off) > (3 if ("-mat_axpy" in args or "-mat_ops" in args) else 10) and np.array_equal(on, off), np.abs(on - off).max()
    line = [ln for ln in on_out.splitlines() if ln.startswith("hipx lazy fusion:")]
    if "-ksp_type cg" in args and "single_reduction" not in args:
        assert line, on_out[-400:]
        nums = [int(t) for t in line[0].replace(";", " ").replace(",", " ").split() if t.isdigit()]
        rec, alone, pairs, inpw, inmm = nums[:5]
        its = len(on) - 1
        assert rec >= 3 * its - 3
        if "-mat_axpy" in args or "-mat_ops" in args:
            assert inpw >= its - 1, line
        elif ("jacobi" in args or "-pc_type none" in args) and "unpreconditioned" not in args:
            assert inpw >= its - 1, line  # "r -= a w" inside PCApply_Jacobi's kernel, every iteration
        if "spmv_variant 30" in args:
            assert inmm >= its - 2, line  # "x += a p; p = z + b p" as the product's prologue
        else:
            assert pairs >= its - 2, line  # ... or as one direction kernel
    if any(k in args for k in ("pipecg", "groppcg", "pipecr")):  # every iteration's update block as batch kernels (one for PIPECG / PIPECR, two for GROPPCG)
        bl = [ln for ln in on_out.splitlines() if "batch kernels" in ln]
        assert bl, on_out[-400:]
        nops, nb = [int(t) for t in bl[0].split() if t.isdigit()][:2]
        its = len(on) - 1
        per = {"pipecg": (8, 1), "groppcg": (5, 2), "pipecr": (6, 1)}[[k for k in ("pipecg", "groppcg", "pipecr") if k in args][0]]
        assert nb >= per[1] * (its - 2) and nops >= per[0] * (its - 2), bl
    on_f = hist_of(run("ref_driver", a + HIPX))
    off_f = hist_of(run("ref_driver", a + HIPX + ["-hipx_lazy_fusion", "0"]))
    assert len(on_f) == len(off_f) == len(on)
    tol = 1e-12 if all(k not in args for k in ("bcgs", "gmres", "cr", "pipecg", "groppcg")) else (1e-9 if all(k not in args for k in ("pipecg", "groppcg", "pipecr")) else 1e-6)
    assert (np.abs(on_f - off_f) / off_f).max() <= tol  # (the pipelined recurrences carry the reductions' rounding forward: the reference's own MKL run is 4e-8 from exact)


@pytest.mark.parametrize("args", ["-stencil 7 -n 40 -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 40",
                                  "-stencil 27 -n 20 -pc_type none -ksp_norm_type unpreconditioned -ksp_rtol 1e-50 -ksp_max_it 30",
                                  "-stencil 7 -n 24 -pc_type jacobi -ksp_norm_type natural -ksp_rtol 1e-50 -ksp_max_it 30",
                                  "-stencil 7 -n 24 -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 25 -mat_axpy",
                                  "-stencil 5 -m 100 -n 100 -pc_type jacobi -ksp_rtol 1e-6"])
def test_ksp_pipecghipx_and_batched_pipecg_equal_the_reference_with_exact_blas(args):
    """Round 6.  Three runs of the same executable: (a) the REFERENCE's KSPSolve_PIPECG on the CPU types with exact BLAS reductions
    (oracle/libexactblas.so), (b) the reference's unmodified KSPSolve_PIPECG over the hipx types (update block = one batch kernel) and (c) -ksp_type pipecghipx
    (the host layer's launch-ahead loop: one fused update kernel + one product per iteration), both with -hipx_reductions exact: the same residual history,
    iteration count and error norm, bit for bit (north_star's 1e-12 met with 0.0).  Default reductions: within rounding."""
    a = args.split() + ["-history"]
    ref = run("ref_driver", a + ["-ksp_type", "pipecg"], exact_blas=True)
    stock = run("ref_driver", a + ["-ksp_type", "pipecg"] + HIPX + ["-hipx_reductions", "exact"])
    fused = run("ref_driver", a + ["-ksp_type", "pipecghipx"] + HIPX + ["-hipx_reductions", "exact", "-info", ":ksp"])
    assert "outside the fused path" not in fused
    h = hist_of(ref)
    assert len(h) > 10 and np.array_equal(hist_of(stock), h) and np.array_equal(hist_of(fused), h)
    tail = lambda t: re.search(r"iterations (\d+) reason (-?\d+) error (\S+)", t).groups()  # noqa: E731
    assert tail(stock) == tail(ref) == tail(fused)
    fast = hist_of(run("ref_driver", a + ["-ksp_type", "pipecghipx"] + HIPX))
    m = min(len(fast), len(h), 12)
    assert (np.abs(fast[:m] - h[:m]) / h[:m]).max() <= 1e-12
    assert abs(len(fast) - len(h)) <= 1 and (np.abs(fast[:min(len(fast), len(h))] - h[:min(len(fast), len(h))]) / h[:min(len(fast), len(h))]).max() <= 1e-6


@pytest.mark.parametrize("args", ["-stencil 7 -n 40 -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 40",
                                  "-stencil 27 -n 20 -pc_type none -ksp_norm_type unpreconditioned -ksp_rtol 1e-50 -ksp_max_it 30",
                                  "-stencil 7 -n 24 -pc_type jacobi -ksp_norm_type natural -ksp_rtol 1e-50 -ksp_max_it 30",
                                  "-stencil 7 -n 24 -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 25 -mat_axpy",
                                  "-stencil 5 -m 100 -n 100 -pc_type jacobi -ksp_rtol 1e-6"])
def test_ksp_groppcghipx_and_batched_groppcg_equal_the_reference_with_exact_blas(args):
    """Round 6: the reference's KSPSolve_GROPPCG with exact BLAS on the CPU types, its unmodified loop over the hipx types (two batch kernels per iteration) and
    -ksp_type groppcghipx (the host layer's two fused passes + one product per iteration), both with -hipx_reductions exact: the same history, iteration count
    and error norm, bit for bit."""
    a = args.split() + ["-history"]
    ref = run("ref_driver", a + ["-ksp_type", "groppcg"], exact_blas=True)
    stock = run("ref_driver", a + ["-ksp_type", "groppcg"] + HIPX + ["-hipx_reductions", "exact"])
    fused = run("ref_driver", a + ["-ksp_type", "groppcghipx"] + HIPX + ["-hipx_reductions", "exact", "-info", ":ksp"])
    assert "outside the fused path" not in fused
    h = hist_of(ref)
    assert len(h) > 10 and np.array_equal(hist_of(stock), h) and np.array_equal(hist_of(fused), h)
    tail = lambda t: re.search(r"iterations (\d+) reason (-?\d+) error (\S+)", t).groups()  # noqa: E731
    assert tail(stock) == tail(ref) == tail(fused)
    fast = hist_of(run("ref_driver", a + ["-ksp_type", "groppcghipx"] + HIPX))
    m = min(len(fast), len(h), 12)
    assert (np.abs(fast[:m] - h[:m]) / h[:m]).max() <= 1e-12


def test_ksp_pipecghipx_falls_back_to_the_reference_loop_when_somebody_watches():
    """-ksp_monitor (or any non-default convergence test) takes the reference's KSPSolve_PIPECG over the hipx types: same text as the CPU run's monitor lines."""
    a = "-stencil 7 -n 16 -pc_type jacobi -ksp_rtol 1e-6 -ksp_monitor -history".split()
    out = run("ref_driver", a + ["-ksp_type", "pipecghipx"] + HIPX + ["-info", ":ksp"])
    assert "outside the fused path" in out
    cpu = run("ref_driver", a + ["-ksp_type", "pipecg"])
    mon = lambda t: np.array([float(v) for v in re.findall(r"KSP Residual norm (\S+)", t)])  # noqa: E731
    assert len(mon(out)) == len(mon(cpu)) > 10 and (np.abs(mon(out) - mon(cpu)) / mon(cpu)).max() <= 1e-8


def test_lazy_fusion_is_invisible_to_the_vector_interface():
    """The reference's own Vec known-answer test (exact-text golden: norms, dots and printed entries taken right after VecAXPY / VecAYPX / VecWAXPY /
    VecPointwiseMult on the same vectors) with every operation recorded (-hipx_lazy_min_size 1) and with the feature off: the same text."""
    on = run("kat_vec_tut_ex1", HIPX[:4] + ["-hipx_lazy_min_size", "1"])
    off = run("kat_vec_tut_ex1", HIPX[:4] + ["-hipx_lazy_fusion", "0"])
    assert on == off and "error" not in on.lower().replace("norm of error", "")


def test_reduction_cache_entries_die_with_the_first_write():
    """The reference's own Vec tests with the cache on (it is on by default in every other test of this file too): ex1 of the tutorials prints
    norms and dots taken right after VecPointwiseMult / VecScale / VecAXPY on the same vectors -- a stale entry would show up in its exact-text golden."""
    on = run("kat_vec_tut_ex1", HIPX[:4])
    off = run("kat_vec_tut_ex1", HIPX[:4] + ["-hipx_reduction_cache", "0"])
    assert on == off and "error" not in on.lower().replace("norm of error", "")
