"""Plane-march PCSOR: the step time and the hop between chunks, measured directly.  Boxes of ONE block of lines (ny + nz - 2 < 64) and 1, 2, 4 ... 16
chunks of four planes: a sweep takes T steps + (chunks - 1) hops, so the difference between two chunk counts is the hop and the one-chunk time / T is
the step.  "blocks": ONE chunk (four planes) and 1, 2, 4, 8 blocks of lines instead: the hop between blocks (124 steps of skew + its hand-off).
python scripts/sor_box_hop.py [nx] [blocks]        (HIPX_LIBDIR / HIPX_SORBOX_* select the variant)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from petsc_amd import _lib  # noqa: E402

hx = _lib.init(0)
LSYM, ZERO = 12, 16
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
blocks = "blocks" in sys.argv


def box27(nx, ny, nz):
    N = nx * ny * nz
    idx = np.arange(N, dtype=np.int64)
    i, j, k = idx % nx, (idx // nx) % ny, idx // (nx * ny)
    rows, cols, vals = [], [], []
    for dk in (-1, 0, 1):
        for dj in (-1, 0, 1):
            for di in (-1, 0, 1):
                ok = (i + di >= 0) & (i + di < nx) & (j + dj >= 0) & (j + dj < ny) & (k + dk >= 0) & (k + dk < nz)
                rows.append(idx[ok])
                cols.append(idx[ok] + di + nx * dj + nx * ny * dk)
                vals.append(np.full(int(ok.sum()), 26.0 if (di, dj, dk) == (0, 0, 0) else -1.0))
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    o = np.lexsort((cols, rows))
    ai = np.zeros(N + 1, np.int32)
    ai[1:] = np.cumsum(np.bincount(rows, minlength=N))
    return N, ai, cols[o].astype(np.int32), vals[o]


e0, e1 = C.c_void_p(), C.c_void_p()
_lib.chk(hx.hipxEventCreate(C.byref(e0)))
_lib.chk(hx.hipxEventCreate(C.byref(e1)))
T = (nx + 3 + 126 + 12 + 3) & ~3
res = []
for ny, nz in (((30, 4), (94, 4), (222, 4), (478, 4)) if blocks else ((40, 4), (40, 8), (40, 16), (30, 32))):
    N, ai, aj, aa = box27(nx, ny, nz)
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    b = 1.0 + (np.arange(N) % 17) / 17.0
    B, X = _lib.DVec(N, b), _lib.DVec(N)
    os.environ["HIPX_SOR_MODE"] = "box"
    for _ in range(2):
        _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, LSYM | ZERO, 0.0, 1, 1, X.ptr))
    used = C.c_int(-2)
    _lib.chk(hx.hipxMatGetSORMode(A, C.byref(used)))
    reps = 10
    _lib.chk(hx.hipxEventRecord(e0))
    for _ in range(reps):
        _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, LSYM | ZERO, 0.0, 1, 1, X.ptr))
    _lib.chk(hx.hipxEventRecord(e1))
    ms = C.c_float()
    _lib.chk(hx.hipxEventElapsedMs(e0, e1, C.byref(ms)))
    us = ms.value / reps / 2 * 1e3  # one sweep
    units = (ny + nz - 2) // 64 + 1 if blocks else nz // 4
    res.append((units, us))
    print("27-pt %d x %d x %d  (mode %d, %d %s, T = %d steps): %.1f us per sweep" % (nx, ny, nz, used.value, units, "blocks" if blocks else "chunks", T, us), flush=True)
    B.free()
    X.free()
    _lib.mat_destroy(A)
c1 = res[0][1]
print("step (one chunk, incl. launch + fill): %.0f ns;  hop between %s: %s us%s" % (c1 / T * 1e3, "blocks" if blocks else "chunks", ", ".join("%.2f" % ((us - c1) / (n - 1)) for n, us in res[1:]),
      "  (of it 124 steps of skew: %.1f us)" % (124 * c1 / T) if blocks else ""))
