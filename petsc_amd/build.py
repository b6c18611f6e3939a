"""Build recipe for the native libraries (in-tree, gfx950 only).

  petsc_amd/lib/libhipx.so      HIP kernels + C ABI (include/hipx.h)          hipcc --offload-arch=gfx950
  petsc_amd/lib/libhipxksp.so   C host layer (include/hipx_ksp.h)            gcc -std=c11
  oracle/liboracle.so           CPU oracle, test infrastructure only          gcc (oracle/Makefile)
  oracle/_ref/...               the reference itself, when /root/reference exists (oracle/build_ref.py)
  petsc_amd/lib/libpetschipx.so PETSc plugin, when oracle/_ref was built (needs the reference headers)

hipcc cross-compiles without a GPU, so this runs in the CPU-only container.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "petsc_amd", "csrc")
HOST = os.path.join(ROOT, "petsc_amd", "host")
LIB = os.path.join(ROOT, "petsc_amd", "lib")
INC = os.path.join(ROOT, "include")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

HIP_SOURCES = ["hipx_runtime.hip", "hipx_vec.hip", "hipx_pipe.hip", "hipx_mat.hip", "hipx_sell.hip", "hipx_sor.hip", "hipx_sorbox.hip", "hipx_comm.hip"]
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
             "-DHIPX_BUILD", "-I" + INC, "-I" + CSRC, "-Wall", "-Wno-unused-function"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build step failed: " + " ".join(cmd))
    return r.stdout


def build_hipx(force=False, verbose=False):
    os.makedirs(LIB, exist_ok=True)
    objdir = os.path.join(LIB, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(INC, "hipx.h")] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    objs = []
    procs = []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + headers):
            cmd = [HIPCC] + HIP_FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    target = os.path.join(LIB, "libhipx.so")
    if force or _newer(target, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs +
             # nodelete: a host that dlclose()s its plugins at finalize (PetscFinalize does) must not unmap code objects the
             # HIP runtime still has registered; the library stays resident until process exit, as under ctypes
             ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-z,nodelete"])
    return target


def build_host(force=False):
    target = os.path.join(LIB, "libhipxksp.so")
    srcs = [os.path.join(HOST, f) for f in ("hipx_ksp.c", "hipx_mpiaij.c", "hipx_drivers.c")]
    hdrs = [os.path.join(INC, "hipx.h"), os.path.join(INC, "hipx_ksp.h")]
    if force or _newer(target, srcs + hdrs + [os.path.join(LIB, "libhipx.so")]):
        _run(["gcc", "-std=c11", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-I" + INC, "-o", target] + srcs +
             ["-L" + LIB, "-lhipx", "-Wl,-rpath,$ORIGIN", "-lm"])
    return target


def build_oracle(force=False):
    if force:
        _run(["make", "-C", os.path.join(ROOT, "oracle"), "clean"])
    _run(["make", "-C", os.path.join(ROOT, "oracle")])
    return os.path.join(ROOT, "oracle", "liboracle.so")


def build_diag(force=False):
    """scripts/diag/*.hip: stand-alone hardware probes (request-path cost, wave placement, DPP wave shift) that
    scripts/gpu_final.sh runs on the GPU box; built in-tree like the libraries (git-ignored, shipped by gpurun)."""
    d = os.path.join(ROOT, "scripts", "diag")
    out = []
    for src in sorted(f for f in os.listdir(d) if f.endswith(".hip")) if os.path.isdir(d) else []:
        s, t = os.path.join(d, src), os.path.join(d, src[:-4])
        if force or _newer(t, [s]):
            _run([HIPCC, "--offload-arch=gfx950", "-O2", "-w", "-o", t, s])
        out.append(t)
    return out


def build_all(force=False, verbose=False):
    out = {"hipx": build_hipx(force, verbose), "host": build_host(force), "oracle": build_oracle(force), "diag": build_diag(force)}
    ref = os.path.join(ROOT, "oracle", "build_ref.py")
    if os.path.exists(ref) and os.path.isdir("/root/reference/src"):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import build_ref  # noqa: E402
        out["ref"] = build_ref.build(verbose=verbose)
        out["ref_mpich"] = build_ref.build(verbose=verbose, arch="mpich")  # same sources against the image's MPICH, for np > 1 parity
        out["ref_int64"] = build_ref.build(verbose=verbose, arch="int64")  # same sources with 64-bit PetscInt (systems beyond 2^31 nonzeros through the drop-in)
        plug = os.path.join(ROOT, "petsc_amd", "plugin", "build_plugin.py")
        if os.path.exists(plug):
            sys.path.insert(0, os.path.join(ROOT, "petsc_amd", "plugin"))
            import build_plugin  # noqa: E402
            out["plugin"] = build_plugin.build(verbose=verbose)
            out["plugin_mpich"] = build_plugin.build(verbose=verbose, arch="mpich")
            out["plugin_int64"] = build_plugin.build(verbose=verbose, arch="int64")
    return out


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose="-v" in sys.argv))
