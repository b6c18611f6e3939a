"""GPU: the PetscSF type "hipx" (petsc_amd/plugin/sfhipx.c) on device buffers.

  * oracle/ref_driver.c -scatter_test: general VecScatters (self + remote edges, out-of-order leaves with a hole, several leaves
    per root) forward INSERT / forward ADD / reverse ADD / one-to-one forward + reverse INSERT between vectors of uneven
    ownership.  The CPU types over PETSCSFBASIC and the hipx types over -sf_type hipx with -vec_hipx_memtype (VecScatterBegin
    then hands the SF the DEVICE mirrors: pack, exchange -- IPC peer stores here, the ranks share this box's GPU -- and unpack
    all run on the device) must print the same 17-digit text, np 1-3; -info names the device plan.
  * MATMPIAIJHIPX with -mat_mpiaijhipx_halo sf: MatMult's Mvctx re-typed to hipx and driven through
    PetscSFBcastWithMemTypeBegin / End on device pointers: y bit-identical to the CPU MPI run, CG history follows it.
The reference's own PetscSF tests with the new type are in tests/golden/kats*.json (sf_ex1 np 1-3, sf_ex4, sf_ex2 = the
reference's device VecScatter test)."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MPIEXEC = "/opt/conda/bin/mpiexec"
ENV = dict(os.environ, HIPX_NO_TORCH="1")


def run(np_, args, hipx, extra_env=None):
    mp = np_ > 1
    exe = os.path.join(ROOT, "oracle", "_ref", "mpich" if mp else "", "bin", "ref_driver")
    plugin = os.path.join(ROOT, "petsc_amd", "lib", "libpetschipx_mpich.so" if mp else "libpetschipx.so")
    assert os.path.exists(exe) and os.path.exists(plugin), "oracle/_ref or the plugin is not built"
    cmd = ([MPIEXEC, "-n", str(np_)] if mp else []) + [exe] + args
    if hipx:
        cmd += ["-dll_prepend", plugin, "-vec_type", "hipx"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=dict(ENV, **(extra_env or {})))
    assert r.returncode == 0, "%s\n%s" % (" ".join(cmd), r.stdout[-3000:])
    return r.stdout


def data_lines(txt):
    return [l for l in txt.splitlines() if re.match(r"(fwd_insert|fwd_add|rev_add|one_fwd|one_rev) \d+ ", l)]


@pytest.mark.parametrize("np_", [1, 2, 3])
def test_general_vecscatter_on_device_buffers_bit_identical(np_):
    cpu = run(np_, ["-scatter_test", "37"], False)
    gpu = run(np_, ["-scatter_test", "37", "-vec_hipx_memtype", "-sf_type", "hipx", "-info", ":sf"], True)
    want, got = data_lines(cpu), data_lines(gpu)
    assert len(want) > 5 * 37 * np_ and got == want  # %.17g text: string equality is bit equality
    assert "scatter type hipx" in gpu and "scatter type basic" in cpu
    assert "PetscSF hipx: device plan built" in gpu, gpu[-2000:]
    if np_ > 1:
        assert "transport ipc" in gpu  # ranks share the GPU here


@pytest.mark.parametrize("np_", [2, 3])
def test_general_vecscatter_device_vectors_but_basic_arguments_fall_back(np_):
    """hipx vectors WITHOUT -vec_hipx_memtype under -sf_type hipx: VecScatter sees host pointers, the type hipx forwards to the
    parent's path: same text again."""
    cpu = run(np_, ["-scatter_test", "19"], False)
    gpu = run(np_, ["-scatter_test", "19", "-sf_type", "hipx"], True)
    assert data_lines(gpu) == data_lines(cpu) and len(data_lines(cpu)) > 0


@pytest.mark.parametrize("np_,args", [(2, "-stencil 7 -n 10"), (3, "-stencil 27 -n 8")])
def test_matmult_mpiaijhipx_through_petscsf_hipx(np_, args):
    a = args.split() + ["-dump_y", "-ksp_type", "cg", "-pc_type", "jacobi", "-ksp_rtol", "1e-8", "-history"]
    cpu = run(np_, a, False)
    gpu = run(np_, a + ["-mat_type", "aijhipx", "-mat_mpiaijhipx_halo", "sf", "-info", ":mat,sf"], True)
    yc = sorted((int(l.split()[1]), l.split()[2]) for l in cpu.splitlines() if l.startswith("y "))
    yg = sorted((int(l.split()[1]), l.split()[2]) for l in gpu.splitlines() if l.startswith("y "))
    assert len(yc) > 0 and yg == yc
    assert "ghost exchange through PetscSF type hipx" in gpu and "PetscSF hipx: device plan built" in gpu
    hc = [float(l.split()[2]) for l in cpu.splitlines() if l.startswith("hist ")]
    hg = [float(l.split()[2]) for l in gpu.splitlines() if l.startswith("hist ")]
    assert len(hc) == len(hg) > 5 and max(abs(g - c) for g, c in zip(hg, hc)) <= 1e-12 * hc[0]
