"""Worker for tests/test_gpu_sor.py::test_strand_kernel_variants_bit_exact: the strand SOR kernels read their variant switches
(HIPX_SOR_SPLIT, HIPX_SOR_STAGGER, HIPX_SOR_WG_PER_CU) once per process, so every combination runs in its own process.
27-point grids with several panels per plane and several planes (stagger applies), aligned (n % 8 == 0) and not; every sweep
kind the split kernel serves (forward / backward zero guess, symmetric with omega != 1: the C wave needs the old value) plus the
general-guess kinds that share the tables."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402
from petsc_amd import _lib  # noqa: E402
import test_gpu_sor as T  # noqa: E402



def box27(nx, ny, nz):
    """27-point operator on an nx x ny x nz box (x fastest), diagonal 26 + small row-dependent term, off-diagonals -1 .. -1.3
    by direction: CSR with sorted columns."""
    rows, cols, vals = [], [], []
    idx = np.arange(nx * ny * nz)
    x, y, z = idx % nx, (idx // nx) % ny, idx // (nx * ny)
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ok = (x + dx >= 0) & (x + dx < nx) & (y + dy >= 0) & (y + dy < ny) & (z + dz >= 0) & (z + dz < nz)
                rows.append(idx[ok])
                cols.append((idx + dx + nx * dy + nx * ny * dz)[ok])
                vals.append(np.full(ok.sum(), 26.5 if (dx, dy, dz) == (0, 0, 0) else -1.0 - 0.1 * abs(dx) - 0.05 * abs(dy) - 0.15 * abs(dz)))
    r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    o = np.lexsort((c, r))
    r, c, v = r[o], c[o], v[o]
    ai = np.zeros(nx * ny * nz + 1, np.int32)
    np.add.at(ai, r + 1, 1)
    return np.cumsum(ai).astype(np.int32), c.astype(np.int32), v.astype(np.float64)


hx = _lib.init(0)
bad = 0
# boxes whose planes are not whole panels (80 and 200 lines: panels straddle planes / staggered boundaries with ragged ends), strands of 16
# and 24 rows: the lockstep C wave (HIPX_SOR_LOCKSTEP=1) is eligible here (far strands are >= 64 lines away); 3 and 5 planes
for (nx, ny, nz) in ((16, 80, 3), (24, 200, 5), (8, 65, 2)):
    ai, aj, aa = box27(nx, ny, nz)
    N = len(ai) - 1
    rng = np.random.default_rng(5)
    b, x0 = rng.standard_normal(N), rng.standard_normal(N)
    for omega, flag, its in ((1.0, T.FWD | T.ZERO, 1), (1.3, T.SYM | T.ZERO, 2), (1.3, T.SYM, 1)):
        g = T.sor_gpu(hx, ai, aj, aa, b, omega, flag, 0.0, its, 1, x0, mode="strand")
        o = T.sor_cpu(ai, aj, aa, b, omega, flag, 0.0, its, 1, x0)
        if not np.array_equal(g, o):
            bad += 1
            print("MISMATCH box %dx%dx%d omega %.1f flag %d its %d: max diff %.3e, first at %d" % (nx, ny, nz, omega, flag, its, np.abs(g - o).max(), int(np.flatnonzero(g != o)[0])))
for n in (24, 128):  # 24^3: one panel per plane, no stagger; 128^3: two panels per plane, 128 planes, staggered boundaries
    ai, aj, aa = orc.stencil("27pt", n)
    N = len(ai) - 1
    m = None
    rng = np.random.default_rng(11)
    b, x0 = rng.standard_normal(N), rng.standard_normal(N)
    for omega in ((1.0, 1.3) if n < 100 else (1.3,)):
        for flag, its in (((T.FWD | T.ZERO, 1), (T.BWD | T.ZERO, 1), (T.SYM | T.ZERO, 1), (T.SYM | T.ZERO, 2), (T.FWD, 1), (T.BWD, 1)) if n < 100 else ((T.SYM | T.ZERO, 1), (T.SYM, 1))):
            g = T.sor_gpu(hx, ai, aj, aa, b, omega, flag, 0.0, its, 1, x0, mode="strand")
            o = T.sor_cpu(ai, aj, aa, b, omega, flag, 0.0, its, 1, x0)
            if not np.array_equal(g, o):
                bad += 1
                print("MISMATCH n %d m %s omega %.1f flag %d its %d: max diff %.3e" % (n, m, omega, flag, its, np.abs(g - o).max()))
print("SOR_VARIANT_OK" if bad == 0 else "SOR_VARIANT_FAILED %d" % bad)
