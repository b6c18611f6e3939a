"""GPU: the plugin's MPI types on REAL MPI ranks.  The reference (same C sources, oracle/build_ref.py arch "mpich") is run under
mpiexec with -vec_type hipx / -mat_type (mpi)aijhipx on 2-3 ranks that share the box's one GPU, and held to
  * the reference's own np > 1 golden outputs (tests/golden/kats_mpi.json: ex1 np 2 exact text, ex21_2, ex28 np 3 with and
    without -splitreduction_async, ex31/ex52 np 2, mat ex5_23 / ex5_33 np 3, ksp ex2_2 np 2), and
  * the CPU MPI run of the same executable on the same ranks, launched here beside it (bit-exact y = A x through
    MatMult_MPIAIJ with hipx diagonal/off-diagonal blocks; residual histories to 1e-12 * ||r0||).
This exercises VecMPIHIPX reductions (device partial + MPI_Allreduce), VecDotBegin/End split reductions, ghost scatter into a
hipx lvec and the A_d x + B x_ghost composition -- the pieces that were covered by construction only with one rank."""
import json
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "mpich", "bin")
PLUGIN = os.path.join(ROOT, "petsc_amd", "lib", "libpetschipx_mpich.so")
MPIEXEC = "/opt/conda/bin/mpiexec"
K = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kats_mpi.json")))
ENV = dict(os.environ, HIPX_NO_TORCH="1")


def mpirun(np_, exe, args, hipx, timeout=150, env=None):
    assert os.path.exists(os.path.join(BIN, exe)) and os.path.exists(PLUGIN), "oracle/_ref/mpich or the MPICH plugin is not built"
    cmd = [MPIEXEC, "-n", str(np_), os.path.join(BIN, exe)] + args
    if hipx:
        cmd += ["-dll_prepend", PLUGIN, "-vec_type", "hipx"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout, env=dict(ENV, **(env or {})))
    assert r.returncode == 0, "%s\n%s" % (" ".join(cmd), r.stdout[-3000:])
    return r.stdout


def apply_filter(kind, text):
    lines = text.splitlines()
    if kind == "notype":  # filter: grep -v type, no diff_args -j: white space is not significant
        lines = [" ".join(ln.split()) for ln in lines if "type" not in ln]
    elif kind == "type":
        lines = [ln for ln in lines if "type" not in ln]
    elif kind == "monitor":  # the golden was written with %g; today's monitor prints 14 digits: reformat the numbers
        out = []
        for ln in lines:
            m = re.match(r"(\s*\d+ KSP Residual norm )(\S+)\s*$", ln)
            out.append(m.group(1) + "%g" % float(m.group(2)) if m else ln)
        lines = out
    elif kind == "ex123":  # grep -v type | grep -v "Mat Object"; diff_args -j = plain `diff -w` (lib/petsc/bin/petscdiff:73): numbers exact,
        # white space not significant (MatView of MPIAIJ indents its rows, the golden was written by SeqAIJ)
        lines = [" ".join(ln.split()) for ln in lines if "type" not in ln and "Mat Object" not in ln]
    return "\n".join(lines).strip()


@pytest.mark.parametrize("name", sorted(K))
def test_reference_np_kat_with_hipx_types(name):
    k = K[name]
    args = k["args"].replace("-mat_type mpiaij", "-mat_type mpiaijhipx").split()
    if k["exe"] == "ex2":
        args += ["-mat_type", "aijhipx"]
    got = apply_filter(k["filter"], mpirun(k["nsize"], k["exe"], args, True))
    want = apply_filter(k["filter"], k["golden"])
    assert got == want, "np-%d KAT %s differs from %s:\n--- got\n%s\n--- want\n%s" % (k["nsize"], name, k["golden_file"], got[-1500:], want[-1500:])


def parse_driver(txt):
    hist = [float(l.split()[2]) for l in txt.splitlines() if l.startswith("hist ")]
    y = sorted((int(l.split()[1]), l.split()[2]) for l in txt.splitlines() if l.startswith("y "))  # ranks print in any order
    m = re.search(r"iterations (\d+) reason (-?\d+) error (\S+)", txt)
    return hist, y, (int(m.group(1)), int(m.group(2)), float(m.group(3))) if m else None


@pytest.mark.parametrize("np_,args", [(2, "-stencil 7 -n 8"), (3, "-stencil 27 -n 6"), (3, "-stencil 5 -m 9 -n 7"), (2, "-stencil 27 -n 6 -dup_mat"),
                                      (2, "-stencil 27 -n 6 -mat_ops"), (3, "-stencil 7 -n 8 -mat_ops"),  # MatScale / MatDiagonalScale on the device blocks
                                      # ... followed by an in-place host edit of the diagonal block (one state increase right after
                                      # MatDiagonalScale_MPIAIJ's un-counted block ops: ADVICE r2, mathipx.c state bookkeeping)
                                      (2, "-stencil 27 -n 6 -mat_ops -mat_ops_block_edit"), (3, "-stencil 7 -n 8 -mat_ops -mat_ops_block_edit")])
def test_matmult_mpiaijhipx_bit_exact_vs_cpu_mpi(np_, args):
    a = args.split() + ["-dump_y", "-ksp_max_it", "1"]
    _, y_cpu, _ = parse_driver(mpirun(np_, "ref_driver", a, False))
    _, y_gpu, _ = parse_driver(mpirun(np_, "ref_driver", a + ["-mat_type", "aijhipx"], True))
    assert len(y_cpu) > 0 and y_gpu == y_cpu  # printed with %.17g: string equality is bit equality


@pytest.mark.parametrize("np_,args", [(2, "-stencil 27 -n 6 -mat_ops"), (3, "-stencil 7 -n 8 -mat_ops"), (3, "-stencil 5 -m 9 -n 7 -mat_ops")])
def test_matmulttranspose_mpiaijhipx_bit_exact_vs_cpu_mpi(np_, args):
    """MatMultTranspose / MatMultTransposeAdd_MPIAIJ (mpiaij.c) over hipx blocks (round 4: the blocks' transposes are device CSR matrices of their
    own, MatMultTranspose_SeqAIJHIPX) on a NON-symmetric operator (D_l A D_r): y^T = A^T x and y + A^T x equal to the CPU MPI run's, digit for digit."""
    a = args.split() + ["-dump_y", "-dump_yt", "-ksp_max_it", "1"]
    cpu = mpirun(np_, "ref_driver", a, False)
    gpu = mpirun(np_, "ref_driver", a + ["-mat_type", "aijhipx"], True)
    for tag in ("yt ", "yta "):
        c = sorted(l for l in cpu.splitlines() if l.startswith(tag))
        g = sorted(l for l in gpu.splitlines() if l.startswith(tag))
        assert len(c) > 0 and c == g, tag
    nc = [float(l.split()[-1]) for l in cpu.splitlines() if l.startswith("MatMultTransposeAdd in place")]
    ng = [float(l.split()[-1]) for l in gpu.splitlines() if l.startswith("MatMultTransposeAdd in place")]
    assert len(nc) == 1 and abs(nc[0] - ng[0]) <= 1e-14 * nc[0]  # (a norm: the reduction's rounding, not the product's)


@pytest.mark.parametrize("np_", [2, 3])
def test_bench_kspsolve_coo_assembly_on_device_np(np_):
    """bench_kspsolve.c:301-302 assembles with MatSetPreallocationCOO / MatSetValuesCOO, every rank its own rows: with
    MATMPIAIJHIPX the values go through the device COO kernel of both blocks (MatSetValuesCOO_MPIAIJHIPX, no entry travels
    between ranks); the -matmult leg and the KSP leg print what the CPU types print."""
    mm = ["-print_timing", "false", "-matmult", "-its", "10", "-n", "8"]
    assert mpirun(np_, "bench_kspsolve", mm + ["-mat_type", "aijhipx", "-options_left", "no"], True).split() == mpirun(np_, "bench_kspsolve", mm, False).split()
    traced = mpirun(np_, "bench_kspsolve", mm + ["-mat_type", "aijhipx", "-options_left", "no"], True, env={"HIPX_TRACE_COO": "1"})
    assert traced.count("MatSetValuesCOO_MPIAIJHIPX: device path") == np_  # every rank took the device path
    ks = ["-print_timing", "false", "-n", "8", "-ksp_type", "cg", "-pc_type", "jacobi", "-ksp_rtol", "1e-8"]
    assert mpirun(np_, "bench_kspsolve", ks + ["-mat_type", "aijhipx", "-options_left", "no"], True).split() == mpirun(np_, "bench_kspsolve", ks, False).split()


@pytest.mark.parametrize("np_,args", [(2, "-stencil 27 -n 6"), (3, "-stencil 7 -n 8"), (3, "-stencil 27 -n 6 -coo_device_values -vec_hipx_memtype"),
                                      (2, "-stencil 5 -m 9 -n 7 -coo_device_values -vec_hipx_memtype")])
def test_coo_entries_travelling_between_ranks_on_the_device(np_, args):
    """MatSetValuesCOO_MPIAIJ's remote part (mpiaij.c:6798-6822): `ref_driver -coo_assemble 2` has every rank set half of each value of
    the first rows of the NEXT rank (the owner sets the other half).  With MATMPIAIJHIPX the entries are packed on the device
    (hipxVecScatterIndexed), sent through a PetscSF of type hipx (device PetscSFReduce, MPI_REPLACE) and added to both blocks by
    the indexed COO kernel -- with host values and with a device-resident value array.  The assembled operator must equal the
    MatSetValues one bit for bit: y = A x and a CG history are compared with the CPU run of the standard assembly."""
    tail = ["-dump_y", "-history", "-ksp_type", "cg", "-pc_type", "jacobi", "-ksp_max_it", "8"]
    base = [x for x in args.split() if x not in ("-coo_device_values", "-vec_hipx_memtype")] + tail
    h_cpu, y_cpu, _ = parse_driver(mpirun(np_, "ref_driver", base, False))
    h_coo_cpu, y_coo_cpu, _ = parse_driver(mpirun(np_, "ref_driver", base + ["-coo_assemble", "2"], False))
    assert y_coo_cpu == y_cpu and h_coo_cpu == h_cpu  # the driver's travelling assembly reproduces the operator on the CPU types
    out = mpirun(np_, "ref_driver", args.split() + tail + ["-coo_assemble", "2", "-mat_type", "aijhipx", "-options_left", "no"], True, env={"HIPX_TRACE_COO": "1"})
    h_gpu, y_gpu, _ = parse_driver(out)
    assert out.count("entries travel between ranks") == 2 * np_, out[-2000:]  # both MatSetValuesCOO calls of every rank took the device path
    if "-coo_device_values" in args:
        assert "coo values memtype 3" in out  # PETSC_MEMTYPE_HIP: the value array was device memory
    assert len(y_cpu) > 0 and y_gpu == y_cpu
    assert len(h_gpu) == len(h_cpu) and max(abs(a - b) for a, b in zip(h_gpu, h_cpu)) <= 1e-12 * h_cpu[0]


def test_hipx_vectors_with_host_mpiaij_and_default_pc():
    """-vec_type hipx only: CPU MPIAIJ matrix, block Jacobi/ILU(0) default PC -- duplicates of MPI hipx vectors, local-vector
    views into them (PCApply_BJacobi_Singleblock), clean exit (heap-checked by glibc)."""
    args = "-m 5 -n 5 -ksp_monitor -ksp_gmres_cgs_refinement_type refine_always".split()
    assert mpirun(2, "ex2", args, True) == mpirun(2, "ex2", args, False)


@pytest.mark.parametrize("np_,args,tol", [
    (2, "-stencil 7 -n 20 -ksp_type cg -pc_type jacobi -ksp_rtol 1e-8", 1e-12),
    (3, "-stencil 27 -n 16 -ksp_type cg -pc_type jacobi -ksp_rtol 1e-8", 1e-12),
    (2, "-stencil 27 -n 12 -ksp_type gmres -pc_type jacobi -ksp_rtol 1e-8", 1e-10),
    (3, "-stencil 7 -n 16 -ksp_type cg -pc_type jacobi -ksp_norm_type natural -ksp_rtol 1e-8", 1e-12),
])
def test_ksp_history_np_vs_cpu_mpi(np_, args, tol):
    a = args.split() + ["-history"]
    h_cpu, _, t_cpu = parse_driver(mpirun(np_, "ref_driver", a, False))
    h_gpu, _, t_gpu = parse_driver(mpirun(np_, "ref_driver", a + ["-mat_type", "aijhipx"], True))
    assert t_gpu[0] == t_cpu[0] and t_gpu[1] == t_cpu[1]
    assert len(h_gpu) == len(h_cpu)
    r0 = h_cpu[0]
    for g, c in zip(h_gpu, h_cpu):
        assert abs(g - c) <= tol * r0 + 1e-9 * abs(c)
    assert abs(t_gpu[2] - t_cpu[2]) <= 1e-10 * max(1.0, abs(t_cpu[2])) + 1e-6 * abs(t_cpu[2])


@pytest.mark.parametrize("np_,args", [(2, "-stencil 7 -n 20 -ksp_type cg -pc_type jacobi -ksp_norm_type preconditioned -ksp_rtol 1e-50 -ksp_max_it 40"),
                                      (3, "-stencil 27 -n 16 -ksp_type cg -pc_type jacobi -ksp_norm_type natural -ksp_rtol 1e-50 -ksp_max_it 30"),
                                      (2, "-stencil 7 -n 16 -ksp_type cg -pc_type jacobi -ksp_norm_type unpreconditioned -ksp_rtol 1e-50 -ksp_max_it 30"),
                                      (2, "-stencil 27 -n 12 -ksp_type gmres -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 35"),
                                      (3, "-stencil 7 -n 16 -ksp_type bcgs -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 12"),
                                      # round 6: the pipelined variants' update blocks as batch kernels on the ranks' local parts
                                      (2, "-stencil 7 -n 20 -ksp_type pipecg -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 30"),
                                      (3, "-stencil 27 -n 12 -ksp_type groppcg -pc_type jacobi -ksp_norm_type unpreconditioned -ksp_rtol 1e-50 -ksp_max_it 25"),
                                      (2, "-stencil 7 -n 16 -ksp_type pipecr -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 25")])
def test_lazy_fusion_on_mpi_vectors(np_, args):
    """Round 5: the vector type's lazy fusion (VecAXPY / VecAYPX recorded, run fused) with VECMPIHIPX on real MPI ranks: the recorded operations are the
    ranks' local parts, the MPIAIJ product reaches the vectors through the accessors (which run what is recorded: x += a p and p = z + b p as one kernel),
    PCJACOBI's VecPointwiseMult takes r -= a w into its kernel.  With exact reductions the histories with and without it are the same doubles."""
    a = args.split() + ["-history", "-mat_type", "aijhipx", "-hipx_reductions", "exact", "-hipx_lazy_min_size", "1"]
    on_txt = mpirun(np_, "ref_driver", a + ["-hipx_lazy_view"], True)
    h_on, _, t_on = parse_driver(on_txt)
    h_off, _, t_off = parse_driver(mpirun(np_, "ref_driver", a + ["-hipx_lazy_fusion", "0"], True))
    assert len(h_on) == len(h_off) > 10 and h_on == h_off and t_on[:2] == t_off[:2]
    if "-ksp_type cg" in args:
        lines = [ln for ln in on_txt.splitlines() if ln.startswith("hipx lazy fusion:")]
        assert len(lines) == np_, on_txt[-600:]
        for ln in lines:
            nums = [int(t) for t in ln.replace(";", " ").replace(",", " ").split() if t.isdigit()]
            its = len(h_on) - 1
            assert nums[0] >= 3 * its - 3 and nums[2] >= its - 2, ln  # every rank: the direction pair as one kernel
            if "unpreconditioned" not in args:
                assert nums[3] >= its - 1, ln  # ... and r -= a w inside PCApply_Jacobi's kernel


@pytest.mark.parametrize("halo", ["ipc", "host"])
def test_mpiaijhipx_ghost_exchange_transports(halo):
    """MatMult_MPIAIJ over hipx blocks with the ghost exchange ON THE DEVICE (ranks share this box's GPU, so the transport is
    the IPC one: peer stores, no host staging) and through the stock host VecScatter: both bit-identical to the CPU MPI run;
    -info names the transport; a 60-iteration CG solve (60 back-to-back exchanges + reductions) follows the CPU history."""
    a = "-stencil 27 -n 10 -dump_y -ksp_max_it 1".split()
    _, y_cpu, _ = parse_driver(mpirun(3, "ref_driver", a, False))
    out = mpirun(3, "ref_driver", a + ["-mat_type", "aijhipx", "-info", ":mat"], True, env={"HIPX_HALO": halo})
    _, y_gpu, _ = parse_driver(out)
    assert len(y_cpu) > 0 and y_gpu == y_cpu
    assert ("transport ipc" in out) == (halo == "ipc"), out[-1500:]
    a = "-stencil 7 -n 24 -ksp_type cg -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 60 -history".split()
    h_cpu, _, t_cpu = parse_driver(mpirun(2, "ref_driver", a, False))
    h_gpu, _, t_gpu = parse_driver(mpirun(2, "ref_driver", a + ["-mat_type", "aijhipx"], True, env={"HIPX_HALO": halo}))
    assert t_gpu[:2] == t_cpu[:2] and len(h_gpu) == len(h_cpu) == 61
    assert max(abs(g - c) / c for g, c in zip(h_gpu, h_cpu)) <= 1e-11


@pytest.mark.parametrize("np_,args", [(2, "-stencil 7 -n 24 -pc_type jacobi -ksp_rtol 1e-8"), (3, "-stencil 27 -n 16 -pc_type jacobi -ksp_rtol 1e-8"), (2, "-stencil 7 -n 16 -pc_type jacobi -ksp_rtol 1e-30 -ksp_max_it 9")])
def test_cghipx_on_mpiaijhipx_fused_solve(np_, args):
    """-ksp_type cghipx on an MPI operator: the fused device CG (SpMV with the device ghost exchange, fused update with its two
    sums all-reduced on the device) under PETSc's monitors / convergence test, against the CPU MPI run of KSPCG."""
    a = args.split() + ["-history"]
    h_cpu, _, t_cpu = parse_driver(mpirun(np_, "ref_driver", a + ["-ksp_type", "cg"], False))
    out = mpirun(np_, "ref_driver", a + ["-ksp_type", "cghipx", "-mat_type", "aijhipx", "-info", ":ksp"], True)
    h_gpu, _, t_gpu = parse_driver(out)
    assert "running the reference KSPSolve_CG" not in out  # the fused path was taken
    assert t_gpu[:2] == t_cpu[:2] and len(h_gpu) == len(h_cpu)
    assert max(abs(g - c) / c for g, c in zip(h_gpu, h_cpu)) <= 1e-9
    assert abs(t_gpu[2] - t_cpu[2]) <= 1e-6 * abs(t_cpu[2]) + 1e-12


@pytest.mark.parametrize("np_,args,kw", [(2, "-stencil 7 -n 24 -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 30", dict(pc="jacobi")),
                                         (3, "-stencil 27 -n 16 -pc_type none -ksp_norm_type unpreconditioned -ksp_rtol 1e-50 -ksp_max_it 25", dict(pc="none", normtype=2)),
                                         (4, "-stencil 7 -n 20 -pc_type jacobi -ksp_norm_type natural -ksp_rtol 1e-8", dict(pc="jacobi", normtype=3)),
                                         (2, "-stencil 7 -n 16 -pc_type jacobi -ksp_rtol 1e-30 -ksp_max_it 7", dict(pc="jacobi"))])
def test_pipecghipx_on_mpiaijhipx(np_, args, kw):
    """Round 6: -ksp_type pipecghipx on real MPI ranks (sharing this box's GPU: IPC transport).  Per iteration and rank: the fused update kernel, whose three
    local sums START their all-reduce (peer stores), the product with its ghost exchange, then the END of the all-reduce (hipxAllreduceEnd) -- PIPECG's single
    reduction hidden behind the product.  Exact reduction mode: the history is the oracle's exact PIPECG on the same row partition (MatMult_MPIAIJ's
    association, sums over the whole vectors rounded once), bit for bit -- the restatement tests/test_oracle_exact.py pins to the reference.  Default mode:
    the CPU MPI run of the reference's KSPSolve_PIPECG to rounding."""
    import numpy as np
    import oracle as orc
    d = dict(zip(args.split()[::2], args.split()[1::2]))
    n = int(d["-n"])
    ai, aj, aa = orc.stencil("7pt" if d["-stencil"] == "7" else "27pt", n)
    b = orc.matmult_mpi(ai, aj, aa, np.ones(n ** 3), np_)  # b = A * 1 as the driver forms it under mpiexec
    rtol = float(d["-ksp_rtol"])
    xo, its_o, reason_o, ho = orc.ksp_solve("pipecg", ai, aj, aa, b, rtol=rtol, max_it=int(d.get("-ksp_max_it", 10000)), nranks=np_, exact=True, **kw)
    a = args.split() + ["-history", "-ksp_type", "pipecghipx", "-mat_type", "aijhipx", "-info", ":ksp"]
    out = mpirun(np_, "ref_driver", a + ["-hipx_reductions", "exact"], True)
    assert "outside the fused path" not in out
    h, _, t = parse_driver(out)
    assert t[:2] == (its_o, reason_o) and len(h) == len(ho)
    assert h == [float(v) for v in ho], max(abs(g - c) / c for g, c in zip(h, ho))
    h_cpu, _, t_cpu = parse_driver(mpirun(np_, "ref_driver", args.split() + ["-history", "-ksp_type", "pipecg"], False))
    h_f, _, t_f = parse_driver(mpirun(np_, "ref_driver", a, True))
    assert abs(t_f[0] - t_cpu[0]) <= 1 and t_f[1] == t_cpu[1]
    m = min(len(h_f), len(h_cpu))
    assert max(abs(g - c) / c for g, c in zip(h_f[:12], h_cpu[:12])) <= 1e-12
    assert max(abs(g - c) / c for g, c in zip(h_f[:m], h_cpu[:m])) <= 1e-6


@pytest.mark.parametrize("np_,args,kw", [(2, "-stencil 7 -n 24 -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 30", dict(pc="jacobi")),
                                         (3, "-stencil 27 -n 16 -pc_type none -ksp_norm_type unpreconditioned -ksp_rtol 1e-50 -ksp_max_it 25", dict(pc="none", normtype=2)),
                                         (4, "-stencil 7 -n 20 -pc_type jacobi -ksp_norm_type natural -ksp_rtol 1e-8", dict(pc="jacobi", normtype=3))])
def test_groppcghipx_on_mpiaijhipx(np_, args, kw):
    """Round 6: -ksp_type groppcghipx on real MPI ranks sharing the GPU: t all-reduced on the stream, {dp, gammaNew} as a split-phase all-reduce around the
    product.  Exact reduction mode: the oracle's exact GROPPCG on the same row partition, bit for bit."""
    import numpy as np
    import oracle as orc
    d = dict(zip(args.split()[::2], args.split()[1::2]))
    n = int(d["-n"])
    ai, aj, aa = orc.stencil("7pt" if d["-stencil"] == "7" else "27pt", n)
    b = orc.matmult_mpi(ai, aj, aa, np.ones(n ** 3), np_)
    xo, its_o, reason_o, ho = orc.ksp_solve("groppcg", ai, aj, aa, b, rtol=float(d["-ksp_rtol"]), max_it=int(d.get("-ksp_max_it", 10000)), nranks=np_, exact=True, **kw)
    out = mpirun(np_, "ref_driver", args.split() + ["-history", "-ksp_type", "groppcghipx", "-mat_type", "aijhipx", "-info", ":ksp", "-hipx_reductions", "exact"], True)
    assert "outside the fused path" not in out
    h, _, t = parse_driver(out)
    assert t[:2] == (its_o, reason_o) and len(h) == len(ho)
    assert h == [float(v) for v in ho], max(abs(g - c) / c for g, c in zip(h, ho))


@pytest.mark.parametrize("np_", [2, 3])
def test_mpiaijhipx_on_a_matrix_with_inodes(np_, tmp_path):
    """A blocked operator loaded on 2-3 ranks (MatLoad splits the rows wherever PetscSplitOwnership says, nodes cut or not): every rank's
    diagonal block finds its own inodes at assembly (the off-diagonal block never uses them: mpiaij.c:824) and the reference multiplies
    (MatMult_SeqAIJ_Inode) and relaxes (MatSOR_MPIAIJ -> MatSOR_SeqAIJ_Inode on the block) accordingly.  With the hipx blocks: y = A x,
    the local symmetric sweep and the CG + PCSOR history equal to the CPU MPI run's, digit for digit / to 1e-12 of the first residual."""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from petsc_amd import matio
    from surrogates import flan_surrogate_spd, inode_matrix
    ai, aj, aa = inode_matrix(nnodes=120, seed=44)
    f = str(tmp_path / "inode.bin")
    matio.write_petsc_binary(f, ai, aj, aa)
    for extra in (["-dump_y"], ["-dump_sor", "28"], ["-dump_sor", "28", "-sor_lits", "2"], ["-dump_sor", "20", "-mat_no_inode"]):
        a = ["-f", f, "-ksp_max_it", "1"] + extra
        cpu = mpirun(np_, "ref_driver", a, False)
        gpu = mpirun(np_, "ref_driver", a + ["-mat_type", "aijhipx"], True)
        tag = "y " if extra[0] == "-dump_y" else "sor "
        c = sorted(l for l in cpu.splitlines() if l.startswith(tag))
        g = sorted(l for l in gpu.splitlines() if l.startswith(tag))
        assert len(c) == len(ai) - 1 and c == g, extra
    ai, aj, aa = flan_surrogate_spd(8)
    f = str(tmp_path / "flan8.bin")
    matio.write_petsc_binary(f, ai, aj, aa)
    a = ["-f", f, "-ksp_type", "cg", "-pc_type", "sor", "-ksp_rtol", "1e-8", "-ksp_norm_type", "preconditioned", "-history"]
    h_cpu, _, t_cpu = parse_driver(mpirun(np_, "ref_driver", a, False))
    h_gpu, _, t_gpu = parse_driver(mpirun(np_, "ref_driver", a + ["-mat_type", "aijhipx"], True))
    assert t_gpu[0] == t_cpu[0] and t_gpu[1] == t_cpu[1] and len(h_gpu) == len(h_cpu)
    for g, c in zip(h_gpu, h_cpu):
        assert abs(g - c) <= 1e-12 * h_cpu[0] + 1e-9 * abs(c)
