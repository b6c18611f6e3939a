"""Builds petsc_amd/lib/libpetschipx.so: the PETSc plugin (C, gcc) against the reference's headers where they lie
(/root/reference/include and the private impl headers under /root/reference/src) and the configuration in
oracle/ref_conf.  Needs oracle/_ref (oracle/build_ref.py) for -lpetsc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
LIB = os.path.join(ROOT, "petsc_amd", "lib")
SRCS = ["vechipx.c", "mathipx.c", "matmpihipx.c", "commhipx.c", "sfhipx.c", "pchipx.c", "ksphipx.c", "register.c"]


def build(verbose=False, arch="mpiuni"):
    """arch 'mpiuni': libpetschipx.so against oracle/_ref/lib; arch 'mpich': libpetschipx_mpich.so against oracle/_ref/mpich/lib
    (MPI types in the ops signatures differ between the reference's MPI stub and a real MPI, hence two builds of one source)."""
    mpich, int64 = arch == "mpich", arch == "int64"
    refdir = os.path.join(ROOT, "oracle", "_ref", arch) if (mpich or int64) else os.path.join(ROOT, "oracle", "_ref")
    target = os.path.join(LIB, "libpetschipx_mpich.so" if mpich else "libpetschipx_int64.so" if int64 else "libpetschipx.so")
    srcs = [os.path.join(HERE, s) for s in SRCS]
    deps = srcs + [os.path.join(HERE, "hipxplugin.h"), os.path.join(ROOT, "include", "hipx.h"), os.path.join(ROOT, "include", "hipx_ksp.h"), os.path.join(LIB, "libhipx.so"), os.path.join(LIB, "libhipxksp.so"),
                   os.path.join(refdir, "lib", "libpetsc.so")]
    if os.path.exists(target) and all(os.path.getmtime(d) <= os.path.getmtime(target) for d in deps):
        return target
    extra = ["-DHIPX_REF_MPICH", "-DHIPX_PLUGIN_REGISTER=PetscDLLibraryRegister_petschipx_mpich", "-I/opt/conda/include"] if mpich else []
    if int64:  # the same sources against the 64-bit-PetscInt build of the reference (oracle/build_ref.py int64)
        extra = ["-DHIPX_REF_INT64", "-DHIPX_PLUGIN_REGISTER=PetscDLLibraryRegister_petschipx_int64"]
    cmd = ["gcc", "-std=gnu11", "-O2", "-fPIC", "-shared", "-Wall", "-Wno-unused-parameter"] + extra + \
          ["-I" + os.path.join(ROOT, "oracle", "ref_conf"), "-I" + os.path.join(REF, "include"), "-I" + REF, "-I" + os.path.join(REF, "include", "petsc"),
           "-I" + os.path.join(ROOT, "include"), "-o", target] + srcs + \
          ["-L" + LIB, "-lhipxksp", "-lhipx", "-L" + os.path.join(refdir, "lib"), "-lpetsc",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + ("$ORIGIN/../../oracle/_ref/mpich/lib" if mpich else "$ORIGIN/../../oracle/_ref/int64/lib" if int64 else "$ORIGIN/../../oracle/_ref/lib"), "-Wl,-rpath,/opt/conda/lib", "-Wl,-z,nodelete"] + \
          (["-L/opt/conda/lib", "-lmpi"] if mpich else [])
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        sys.stderr.write(r.stdout[-6000:])
        raise RuntimeError("building libpetschipx.so failed")
    return target


if __name__ == "__main__":
    print(build(verbose=True), build(verbose=True, arch="mpich"), build(verbose=True, arch="int64"))
