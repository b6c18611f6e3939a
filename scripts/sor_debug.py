"""GPU debugging aid: one MatSOR call per schedule on a small stencil matrix, with HIPX_SOR_DEBUG progress lines."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402  (debug script: the oracle is the checker)
from petsc_amd import _lib  # noqa: E402

kind, n, mode, flag = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
os.environ["HIPX_SOR_MODE"] = mode
os.environ["HIPX_SOR_DEBUG"] = "1"
hx = _lib.init(0)
ai, aj, aa = orc.stencil(kind, n)
N = len(ai) - 1
rng = np.random.default_rng(7)
b, x0 = rng.standard_normal(N), rng.standard_normal(N)
A = _lib.mat_create_csr(N, N, ai, aj, aa)
B, X = _lib.DVec(N, b), _lib.DVec(N, x0)
ierr = hx.hipxMatSOR(A, B.ptr, 1.0, flag, 0.0, 1, 1, X.ptr)
print("ierr", ierr, hx.hipxGetErrorString().decode() if ierr else "")
xo = np.array(x0)
orc.lib().orc_MatSOR_SeqAIJ(N, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(b), C.c_double(1.0), flag, C.c_double(0.0), 1, 1, orc.P(xo))
g = X.get()
print(kind, n, mode, flag, "bit-exact", np.array_equal(g, xo), "max diff", np.abs(g - xo).max(), "first bad", int(np.argmax(g != xo)) if not np.array_equal(g, xo) else -1)
