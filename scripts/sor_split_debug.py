"""Localize a split-kernel mismatch: 27-pt, n = 24, omega 1.3, sweeps one by one (debug helper)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc
from petsc_amd import _lib
import test_gpu_sor as T
hx = _lib.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ai, aj, aa = orc.stencil("27pt", n)
N = len(ai) - 1
rng = np.random.default_rng(7)
b = rng.standard_normal(N); x0 = rng.standard_normal(N)
for omega in (1.0, 1.3):
    for flag, its, name in ((T.FWD | T.ZERO, 1, "fwd zero (KIND 0)"), (T.BWD | T.ZERO, 1, "bwd zero (KIND 2)"), (T.SYM | T.ZERO, 1, "sym zero (KIND 0+1)"), (T.SYM | T.ZERO, 2, "sym zero its 2 (+3,4)"), (T.FWD, 1, "fwd (3)"), (T.BWD, 1, "bwd (4)")):
        g = T.sor_gpu(hx, ai, aj, aa, b, omega, flag, 0.0, its, 1, x0, mode="strand")
        o = T.sor_cpu(ai, aj, aa, b, omega, flag, 0.0, its, 1, x0)
        bad = np.where(g != o)[0]
        print("omega %.1f %-24s max diff %.3e  mismatching rows %d %s" % (omega, name, np.abs(g - o).max(), len(bad), bad[:12]))
