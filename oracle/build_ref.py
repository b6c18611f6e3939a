"""oracle/build_ref.py -- compile the REFERENCE itself (PETSc, C sources where they lie under /root/reference) into
oracle/_ref/ with plain gcc.  Test infrastructure: the result is the strongest oracle (the reference's own MatMult_SeqAIJ,
KSPSolve_CG, ... ) and the CPU baseline ("kind": "reference"), and the host library the plugin libpetschipx.so loads into.

The reference's own build system (./configure, gmakefile) is NOT run.  This script:
  1. uses the hand-written configuration oracle/ref_conf/petscconf.h (facts about this image + our choices),
  2. walks /root/reference/src with the reference's documented selection rule -- a directory is compiled unless its
     `makefile` carries a `#requires<kind> 'X'` line that the configuration does not satisfy (doc/developers/buildsystem.md);
     tests/, tutorials/, benchmarks/, ftn-* are never part of the library,
  3. compiles every selected .c with gcc -O2 (parallel), links oracle/_ref/lib/libpetsc.so against libmkl_rt,
  4. builds the reference's tutorial drivers ex2 and bench_kspsolve (their sources, unmodified) plus oracle/ref_driver.c.
No reference source is copied into the repository; outputs go only to oracle/_ref/ (git-ignored, shipped by gpurun).
"""
import concurrent.futures as cf
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.path.join(HERE, "_ref")
CONF = os.path.join(HERE, "ref_conf")
PKGS = "sys vec mat dm ksp snes ts tao ml".split()
SKIPDIRS = {"benchmarks", "build", "mex-scripts", "tests", "tutorials"}
BLAS_DIR = "/opt/conda/lib"
KATS = {  # name -> source (reference tree)
    "vec_tut_ex1": "vec/vec/tutorials/ex1.c",   # Max/Min/Scale/Copy/AXPY/AYPX/Swap/WAXPY/PointwiseMult/Divide/MAXPY/Dot/MDot/Norm, exact text
    "vec_ex21": "vec/vec/tests/ex21.c",         # VecMax with index, VecSetStdBasis
    "vec_ex28": "vec/vec/tests/ex28.c",         # repeated VecDotBegin/End
    "vec_ex31": "vec/vec/tests/ex31.c",
    "vec_ex34": "vec/vec/tests/ex34.c",         # norm caching semantics
    "vec_ex43": "vec/vec/tests/ex43.c",         # VecMDot/Dot/MTDot/TDot
    "vec_ex52": "vec/vec/tests/ex52.c",
    "vec_ex60": "vec/vec/tests/ex60.c",         # VecPlaceArray + VecReciprocal
    "vec_ex63": "vec/vec/tests/ex63.c",         # VecExp (parent op through our array hooks)
    "mat_ex5": "mat/tests/ex5.c",               # MatMult/MultAdd/MultTranspose (+ diagonal scale)
    "mat_ex123": "mat/tests/ex123.c",           # MatSetPreallocationCOO / MatSetValuesCOO (repeated and negative indices, ADD/INSERT)
    "sf_ex1": "vec/is/sf/tests/ex1.c",          # PetscSF Bcast / Reduce / FetchAndOp / Gather / Scatter / Compose ... (-sf_type hipx: SURVEY 8(f3))
    "sf_ex2": "vec/is/sf/tests/ex2.c",          # VecScatter from a device vector into a host-resident one (the reference's own hip test)
    "sf_ex4": "vec/is/sf/tests/ex4.c",          # PetscSFCompose
    "pc_ex3": "ksp/pc/tests/ex3.c",             # GMRES + symmetric PCSOR on a tridiagonal MATSEQAIJ (the reference's MatSOR test), monitor golden
}
CFLAGS = ["-fPIC", "-O2", "-fstack-protector", "-fvisibility=hidden", "-w", "-I" + CONF, "-I" + os.path.join(REF, "include")]


MPI_DIR = "/opt/conda"  # MPICH 3.3.2 of the image (its mpicc wrapper points at a missing compiler: use include/lib directly)


def conf_defines(arch):
    """PETSC_* macros of the configuration for this arch, through the preprocessor (the header has an #if for mpich)."""
    cmd = ["gcc", "-dM", "-E", "-I" + CONF, os.path.join(CONF, "petscconf.h")] + (["-DHIPX_REF_MPICH"] if arch == "mpich" else []) + (["-DHIPX_REF_INT64"] if arch == "int64" else [])
    txt = subprocess.check_output(cmd, text=True)
    return set(re.findall(r"^#define\s+(PETSC_\w+)", txt, flags=re.M))


def dir_selected(makefile, defs):
    """The reference's rule: conditions on separate lines are AND-ed, values on one line are OR-ed."""
    for line in open(makefile, errors="replace"):
        if not line.startswith("#requires"):
            continue
        toks = line[len("#requires"):].replace("'", "").split()
        if not toks:
            continue
        key, vals = toks[0], toks[1:]
        if key in ("package", "define", "function"):
            ok = any(v in defs for v in vals)
        elif key == "precision":
            ok = "double" in vals
        elif key == "scalar":
            ok = "real" in vals
        elif key == "language":
            ok = "C" in vals or "c" in vals
        else:
            raise RuntimeError("unknown #requires kind in %s: %s" % (makefile, line))
        if not ok:
            return False
    return True


def select_sources(arch="mpiuni"):
    defs = conf_defines(arch)
    srcs = []
    for pkg in PKGS:
        for root, dirs, files in os.walk(os.path.join(REF, "src", pkg)):
            dirs.sort()
            if os.path.basename(root).startswith("ftn-"):
                dirs[:] = []
                continue
            dirs[:] = [d for d in dirs if d not in SKIPDIRS]
            mk = os.path.join(root, "makefile")
            if os.path.isfile(mk) and not dir_selected(mk, defs):
                dirs[:] = []
                continue
            srcs += [os.path.join(root, f) for f in sorted(files) if f.endswith(".c")]
    return srcs


def _compile(job):
    src, obj, flags = job
    if os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src):
        return None
    os.makedirs(os.path.dirname(obj), exist_ok=True)
    r = subprocess.run(["gcc"] + flags + ["-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return (src, r.stdout) if r.returncode else None


def arch_flags(arch):
    """(output dir, compile flags, link flags) -- mpiuni: the reference's single-process MPI stub; mpich: the image's MPICH."""
    if arch == "mpich":
        return (os.path.join(OUT, "mpich"), CFLAGS + ["-DHIPX_REF_MPICH", "-I" + os.path.join(MPI_DIR, "include")],
                ["-L" + os.path.join(MPI_DIR, "lib"), "-Wl,-rpath," + os.path.join(MPI_DIR, "lib"), "-lmpi"])
    if arch == "int64":  # MPIUNI with 64-bit PetscInt (--with-64-bit-indices): systems beyond 2^31 nonzeros through the drop-in (27-pt 512^3)
        return os.path.join(OUT, "int64"), CFLAGS + ["-DHIPX_REF_INT64"], []
    return OUT, CFLAGS, []


def build(verbose=False, jobs=None, arch="mpiuni"):
    if not os.path.isdir(os.path.join(REF, "src")):
        raise RuntimeError("reference tree not present: oracle/_ref can only be (re)built where /root/reference exists")
    OUTA, cflags, mpilink = arch_flags(arch)
    lib = os.path.join(OUTA, "lib", "libpetsc.so")
    os.makedirs(os.path.join(OUTA, "lib"), exist_ok=True)
    os.makedirs(os.path.join(OUTA, "bin"), exist_ok=True)
    stamp = os.path.join(OUTA, "conf.stamp")
    conf_txt = "".join(open(os.path.join(CONF, f)).read() for f in sorted(os.listdir(CONF))) + " ".join(cflags)
    if os.path.exists(stamp) and open(stamp).read() != conf_txt:
        shutil.rmtree(os.path.join(OUTA, "obj"), ignore_errors=True)  # configuration changed: rebuild everything
    srcs = select_sources(arch)
    pairs = [(s, os.path.join(OUTA, "obj", os.path.relpath(s, REF)[:-2] + ".o"), cflags) for s in srcs]
    jobs = jobs or max(1, (os.cpu_count() or 2))
    failed = []
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        for res in ex.map(_compile, pairs):
            if res:
                failed.append(res)
    if failed:
        for src, out in failed[:5]:
            sys.stderr.write("FAILED %s\n%s\n" % (src, out[-2000:]))
        raise RuntimeError("%d reference files failed to compile" % len(failed))
    objs = [o for _, o, _ in pairs]
    if not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        rsp = os.path.join(OUTA, "objs.rsp")
        open(rsp, "w").write("\n".join(objs))
        cmd = ["gcc", "-shared", "-fPIC", "-Wl,-soname,libpetsc.so", "-o", lib, "@" + rsp, "-L" + BLAS_DIR, "-Wl,-rpath," + BLAS_DIR, "-lmkl_rt", "-lm", "-ldl", "-lpthread"] + mpilink
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            sys.stderr.write(r.stdout[-4000:])
            raise RuntimeError("linking libpetsc.so failed")
    open(stamp, "w").write(conf_txt)
    # drivers: the reference's own tutorials (sources unmodified, compiled in place) + our thin driver
    tut = os.path.join(REF, "src", "ksp", "ksp", "tutorials")
    drivers = {"ex2": os.path.join(tut, "ex2.c"), "bench_kspsolve": os.path.join(tut, "bench_kspsolve.c"), "ref_driver": os.path.join(HERE, "ref_driver.c")}
    # the reference's own known-answer tests for this path (sources unmodified, compiled in place): they are run with
    # -vec_type hipx / -mat_type aijhipx against the reference's golden outputs (tests/test_gpu_plugin_kats.py)
    for name, rel in KATS.items():
        drivers["kat_" + name] = os.path.join(REF, "src", rel)
    for name, src in drivers.items():
        exe = os.path.join(OUTA, "bin", name)
        if not os.path.exists(src):
            continue
        if os.path.exists(exe) and os.path.getmtime(exe) >= max(os.path.getmtime(src), os.path.getmtime(lib)):
            continue
        cmd = ["gcc", "-O2", "-w"] + [f for f in cflags if f.startswith("-I") or f.startswith("-D")] + [src, "-o", exe, "-L" + os.path.join(OUTA, "lib"), "-lpetsc",
               "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath," + BLAS_DIR, "-L" + BLAS_DIR, "-lmkl_rt", "-lm", "-rdynamic"] + mpilink
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            sys.stderr.write(r.stdout[-4000:])
            raise RuntimeError("building driver %s failed" % name)
    if verbose:
        print("reference built (%s): %d sources -> %s" % (arch, len(srcs), lib))
    return lib


if __name__ == "__main__":
    import time
    for a in (["mpiuni", "mpich"] if "--all" in sys.argv else [sys.argv[1]] if len(sys.argv) > 1 else ["mpiuni"]):
        t = time.time()
        print(build(verbose=True, arch=a), "%.1fs" % (time.time() - t))
