"""ctypes bindings of the C ABI (include/hipx.h, include/hipx_ksp.h).  Fails loudly when a library is
missing: there is no Python/NumPy fallback for any kernel."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.environ.get("HIPX_LIBDIR") or os.path.join(HERE, "lib")  # HIPX_LIBDIR: a developer's A/B build of both libraries in another directory

c_int_p = C.POINTER(C.c_int)
c_dbl_p = C.POINTER(C.c_double)


class HipxError(RuntimeError):
    pass


class HipxMat(C.Structure):
    _fields_ = [("m", C.c_int32), ("A", C.c_void_p), ("B", C.c_void_p), ("halo", C.c_void_p), ("lvec", C.c_void_p), ("nranks", C.c_int)]


class HipxPC(C.Structure):
    _fields_ = [("type", C.c_int), ("dinv", C.c_void_p), ("sor_flag", C.c_int), ("sor_omega", C.c_double), ("sor_shift", C.c_double),
                ("sor_its", C.c_int32), ("sor_lits", C.c_int32), ("dconst_valid", C.c_int), ("dconst", C.c_double)]


class HipxKSP(C.Structure):
    _fields_ = [("normtype", C.c_int), ("rtol", C.c_double), ("abstol", C.c_double), ("divtol", C.c_double), ("max_it", C.c_int32),
                ("min_it", C.c_int32), ("gmres_restart", C.c_int32), ("gmres_haptol", C.c_double), ("gmres_cgs_refine", C.c_int),
                ("guess_nonzero", C.c_int), ("fused", C.c_int), ("its", C.c_int32), ("reason", C.c_int), ("rnorm", C.c_double),
                ("rnorm0", C.c_double), ("ttol", C.c_double), ("history", C.c_void_p), ("hist_len", C.c_int32), ("hist_n", C.c_int32),
                ("R", C.c_void_p), ("Z", C.c_void_p), ("P", C.c_void_p), ("beta", C.c_double), ("betaold", C.c_double), ("dpi", C.c_double),
                ("a", C.c_double), ("i", C.c_int32), ("work_n", C.c_int32), ("x_pending", C.c_int), ("a_pending", C.c_double), ("defer_flush", C.c_int), ("external_test", C.c_int), ("pipeline", C.c_int), ("dscal", C.c_void_p), ("single_reduction", C.c_int), ("S", C.c_void_p), ("W", C.c_void_p), ("delta", C.c_double), ("gslab", C.c_void_p), ("gslab_len", C.c_double), ("P2", C.c_void_p),
                ("pipe_slab", C.c_void_p), ("pipe_slab_len", C.c_double)]


class MPIAIJSplit(C.Structure):
    _fields_ = [("m", C.c_int32), ("nghost", C.c_int32), ("nrows_c", C.c_int32), ("Ai", C.c_void_p), ("Aj", C.c_void_p), ("Aa", C.c_void_p),
                ("Bi", C.c_void_p), ("Bj", C.c_void_p), ("Ba", C.c_void_p), ("ridx", C.c_void_p), ("garray", C.c_void_p)]
    # field order must match HipxMPIAIJSplit in include/hipx_ksp.h


_libs = {}
INCDIR = os.path.join(os.path.dirname(HERE), "include")


def parse_header(path):
    """[(name, restype, [argtypes])] for every function prototype in a C header of this ABI."""
    import re
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", " ", txt)
    txt = re.sub(r"^\s*#.*$", " ", txt, flags=re.M)
    txt = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", " ", txt, flags=re.S)
    out = []

    def ctype(t):
        t = t.strip()
        if "*" in t or t in ("hipxMat", "hipxHalo", "hipxCOO"):
            return C.c_char_p if t.replace(" ", "") == "constchar*" else C.c_void_p
        base = t.replace("const", "").strip()
        return {"int": C.c_int, "hipx_int": C.c_int32, "double": C.c_double, "float": C.c_float, "size_t": C.c_size_t,
                "int64_t": C.c_int64, "void": None}[base]

    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b((?:hipx|Hipx)\w+)\s*\(([^;{}]*?)\)\s*;", txt):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef") or not ret:
            continue
        argt = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                a = re.sub(r"\[\s*\]", "*", a)
                if "*" in a:
                    argt.append(C.c_void_p)
                else:
                    toks = a.split()
                    argt.append(ctype(" ".join(toks[:-1]) if len(toks) > 1 else toks[0]))
        out.append((name, ctype(ret), argt))
    return out


def declared_functions():
    return {"hipx": parse_header(os.path.join(INCDIR, "hipx.h")), "ksp": parse_header(os.path.join(INCDIR, "hipx_ksp.h"))}


def _bind(lib, protos):
    for name, ret, argt in protos:
        f = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        f.restype = ret
        f.argtypes = argt


def lib_paths():
    return {"hipx": os.path.join(LIBDIR, "libhipx.so"), "ksp": os.path.join(LIBDIR, "libhipxksp.so")}


def load():
    """Returns (libhipx, libhipxksp) ctypes handles; raises HipxError if either is not built."""
    if _libs:
        return _libs["hipx"], _libs["ksp"]
    # One HIP runtime per process: torch bundles its own libamdhip64/librccl (same sonames as /opt/rocm's).  Import torch
    # FIRST so that libhipx binds to the runtime torch already loaded; the other order leaves two runtimes half-shared
    # and corrupts the heap at exit.  HIPX_NO_TORCH=1 keeps torch out (pure /opt/rocm runtime, e.g. the PETSc plugin).
    if os.environ.get("HIPX_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    p = lib_paths()
    for k, path in p.items():
        if not os.path.exists(path):
            raise HipxError("native library missing: %s -- run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
    hx = C.CDLL(p["hipx"], mode=C.RTLD_GLOBAL)
    ks = C.CDLL(p["ksp"], mode=C.RTLD_GLOBAL)
    d = declared_functions()
    _bind(hx, d["hipx"])
    _bind(ks, d["ksp"])
    sm, sp, sk = C.c_int(), C.c_int(), C.c_int()
    ks.HipxStructSizes(C.byref(sm), C.byref(sp), C.byref(sk))
    got = (C.sizeof(HipxMat), C.sizeof(HipxPC), C.sizeof(HipxKSP))
    if got != (sm.value, sp.value, sk.value):
        raise HipxError("ctypes mirrors of HipxMat/HipxPC/HipxKSP are out of date: %s vs C %s (include/hipx_ksp.h)" % (got, (sm.value, sp.value, sk.value)))
    _libs["hipx"], _libs["ksp"] = hx, ks
    return hx, ks


def chk(ierr):
    if ierr:
        hx, _ = load()
        raise HipxError("hipx call failed (%d): %s" % (ierr, hx.hipxGetErrorString().decode()))


def init(device=0):
    hx, _ = load()
    chk(hx.hipxInit(int(device)))
    return hx


class DVec:
    """A device array of doubles owned through hipxMalloc (the VECSEQHIPX device mirror)."""

    def __init__(self, n, data=None):
        self.hx, _ = load()
        self.n = int(n)
        p = C.c_void_p()
        chk(self.hx.hipxMalloc(C.byref(p), C.c_size_t(max(self.n, 1) * 8)))
        self.ptr = C.c_void_p(p.value)
        if data is not None:
            self.set(data)

    def set(self, data):
        a = np.ascontiguousarray(data, dtype=np.float64)
        assert a.size == self.n
        chk(self.hx.hipxMemcpyHtoD(self.ptr, a.ctypes.data_as(C.c_void_p), C.c_size_t(self.n * 8)))

    def get(self):
        out = np.empty(self.n, dtype=np.float64)
        chk(self.hx.hipxMemcpyDtoH(out.ctypes.data_as(C.c_void_p), self.ptr, C.c_size_t(self.n * 8)))
        return out

    def free(self):
        if self.ptr is not None and self.ptr.value:
            chk(self.hx.hipxFree(self.ptr))
            self.ptr = None

    def offset(self, k):
        return C.c_void_p(self.ptr.value + 8 * int(k))


def mat_create_csr(m, n, ai, aj, aa):
    hx, _ = load()
    ai = np.ascontiguousarray(ai)
    aj = np.ascontiguousarray(aj, dtype=np.int32)
    aa = np.ascontiguousarray(aa, dtype=np.float64)
    A = C.c_void_p()
    if ai.dtype == np.int64:
        chk(hx.hipxMatCreateCSR64(int(m), int(n), ai.ctypes.data_as(C.c_void_p), aj.ctypes.data_as(C.c_void_p), aa.ctypes.data_as(C.c_void_p), C.byref(A)))
    else:
        ai = ai.astype(np.int32, copy=False)
        chk(hx.hipxMatCreateCSR(int(m), int(n), ai.ctypes.data_as(C.c_void_p), aj.ctypes.data_as(C.c_void_p), aa.ctypes.data_as(C.c_void_p), C.byref(A)))
    return A


def mat_create_cprow(m, n, nrows, ci, ridx, aj, aa):
    hx, _ = load()
    ci = np.ascontiguousarray(ci, dtype=np.int32)
    ridx = np.ascontiguousarray(ridx, dtype=np.int32)
    aj = np.ascontiguousarray(aj, dtype=np.int32)
    aa = np.ascontiguousarray(aa, dtype=np.float64)
    A = C.c_void_p()
    chk(hx.hipxMatCreateCSRCompressedRow(int(m), int(n), int(nrows), ci.ctypes.data_as(C.c_void_p), ridx.ctypes.data_as(C.c_void_p),
                                         aj.ctypes.data_as(C.c_void_p), aa.ctypes.data_as(C.c_void_p), C.byref(A)))
    return A


def mat_destroy(A):
    hx, _ = load()
    chk(hx.hipxMatDestroy(C.byref(A)))
