"""petsc_amd -- MI355X-native Krylov inner loop (CSR SpMV + Vec BLAS-1 + PCJACOBI/PCSOR) behind PETSc's
Vec/Mat/PC plugin API.  The product is native: petsc_amd/lib/libhipx.so (HIP kernels, C ABI
include/hipx.h), libhipxksp.so (C host layer) and libpetschipx.so (PETSc plugin).  This Python package
is only the ctypes loader used by tests and bench.py; it has no numerical fallback of its own."""
from . import _lib  # noqa: F401
from ._lib import load, HipxError  # noqa: F401
