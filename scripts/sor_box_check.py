"""GPU check of the plane-march PCSOR schedule (petsc_amd/csrc/hipx_sorbox.hip): zero-guess sweeps on 7- and 27-point boxes against the oracle's
MatSOR_SeqAIJ restatement (small boxes) and against the level-ordered schedule of the same library (large ones), with a map of where the first
differences sit; then timings against the strand schedule.   python scripts/sor_box_check.py [quick]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402  (checker)
from petsc_amd import _lib  # noqa: E402

hx = _lib.init(0)
_, ks = _lib.load()
FWD, BWD, LFWD, LBWD, LSYM, ZERO = 1, 2, 4, 8, 12, 16


def box_csr(nx, ny, nz, full27, seed=3):
    """constant-coefficient box stencil, natural ordering: 27-point (values by |d| class, all couplings negative, distinct per position) or 7-point"""
    rng = np.random.default_rng(seed)
    N = nx * ny * nz
    I, Jg, K = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    i, j, k = I.ravel(order="F"), Jg.ravel(order="F"), K.ravel(order="F")  # x fastest
    rows, cols, vals = [], [], []
    for dk in (-1, 0, 1):
        for dj in (-1, 0, 1):
            for di in (-1, 0, 1):
                nd = abs(di) + abs(dj) + abs(dk)
                if not full27 and nd > 1:
                    continue
                ok = (i + di >= 0) & (i + di < nx) & (j + dj >= 0) & (j + dj < ny) & (k + dk >= 0) & (k + dk < nz)
                r = (i + nx * (j + ny * k))[ok]
                c = r + di + nx * dj + nx * ny * dk
                v = 7.5 + rng.random() if nd == 0 else -(0.125 + 0.5 * rng.random())
                rows.append(r)
                cols.append(c)
                vals.append(np.full(len(r), v))
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    o = np.lexsort((cols, rows))
    rows, cols, vals = rows[o], cols[o], vals[o]
    ai = np.zeros(N + 1, np.int32)
    ai[1:] = np.cumsum(np.bincount(rows, minlength=N))
    return ai, cols.astype(np.int32), vals


def sor(A, N, b, flag, omega, shift, mode):
    B, X = _lib.DVec(N, b), _lib.DVec(N, np.zeros(N))
    old = os.environ.pop("HIPX_SOR_MODE", None)
    os.environ["HIPX_SOR_MODE"] = mode
    try:
        _lib.chk(hx.hipxMatSOR(A, B.ptr, omega, flag, shift, 1, 1, X.ptr))
    finally:
        os.environ.pop("HIPX_SOR_MODE", None)
        if old:
            os.environ["HIPX_SOR_MODE"] = old
    used = C.c_int(-2)
    _lib.chk(hx.hipxMatGetSORMode(A, C.byref(used)))
    x = X.get()
    B.free()
    X.free()
    return x, used.value


def where(d, nx, ny, nz):
    bad = np.nonzero(d)[0]
    i, j, k = bad % nx, (bad // nx) % ny, bad // (nx * ny)
    return "%d rows differ; first (i, j, k) = %s; planes %s..%s, lines %s..%s, i %s..%s" % (len(bad), list(zip(i[:4], j[:4], k[:4])), k.min(), k.max(), j.min(), j.max(), i.min(), i.max())


if "stats" in sys.argv:  # HIPX_SORBOX_STATS=1 python scripts/sor_box_check.py stats: one application on 27-pt 256^3 and on config 3's slab, wait statistics on stderr
    import bench
    for dims, planes in (((256, 256, 256), None), ((512, 512, 512), 64)):
        n = dims[0]
        N = n * n * (planes or n)
        ai, aj, aa = bench.assemble(ks, 27, dims, 0, N)
        if planes:
            keep = aj < N
            rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(ai))[keep]
            aj, aa = aj[keep], aa[keep]
            ai = np.zeros(N + 1, np.int32)
            ai[1:] = np.cumsum(np.bincount(rows, minlength=N))
        A = _lib.mat_create_csr(N, N, ai, aj, aa)
        b = 1.0 + (np.arange(N) % 17) / 17.0
        sor(A, N, b, LSYM | ZERO, 1.0, 0.0, "box")
        sor(A, N, b, LSYM | ZERO, 1.0, 0.0, "box")
        _lib.mat_destroy(A)
    sys.exit(0)

nfail = 0
small = [(8, 70, 6, True), (8, 70, 6, False), (16, 66, 9, False), (6, 130, 3, True), (12, 12, 12, True), (10, 5, 70, True), (4, 64, 4, True), (32, 40, 5, True), (64, 64, 64, True), (64, 64, 64, False)]
if "timeonly" in sys.argv:
    small = []
for nx, ny, nz, full in small:
    ai, aj, aa = box_csr(nx, ny, nz, full)
    N = nx * ny * nz
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    b = np.random.default_rng(5).standard_normal(N)
    for flag, omega, shift in ((LSYM | ZERO, 1.0, 0.0), (FWD | ZERO, 1.0, 0.0), (BWD | ZERO, 1.0, 0.0), (LSYM | ZERO, 1.3, 0.25), (LFWD | ZERO, 0.8, 0.0), (LBWD | ZERO, 1.3, 0.0)):
        try:
            xg, used = sor(A, N, b, flag, omega, shift, "box")
        except Exception as e:  # noqa: BLE001
            print("box %dx%dx%d %s flag %d: ERROR %s" % (nx, ny, nz, "27pt" if full else "7pt", flag, str(e)[:200]))
            nfail += 1
            break
        xc = np.zeros(N)
        orc.lib().orc_MatSOR_SeqAIJ_dispatch(N, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(b), C.c_double(omega), flag, C.c_double(shift), 1, 1, orc.P(xc), 0)
        ok = np.array_equal(xg, xc)
        print("box %3dx%3dx%3d %4s flag %2d omega %.1f shift %.2f mode %d: %s" % (nx, ny, nz, "27pt" if full else "7pt", flag, omega, shift, used, "bit-identical to the oracle" if ok else "DIFFERENT " + where(xg != xc, nx, ny, nz)), flush=True)
        nfail += 0 if ok else 1
    _lib.mat_destroy(A)
    if nfail > 3:
        print("stopping: too many failures")
        sys.exit(1)
if nfail:
    sys.exit(1)

import bench  # noqa: E402
e0, e1 = C.c_void_p(), C.c_void_p()
_lib.chk(hx.hipxEventCreate(C.byref(e0)))
_lib.chk(hx.hipxEventCreate(C.byref(e1)))
big = [("27-pt 128^3", 27, (128, 128, 128), None), ("7-pt 128^3", 7, (128, 128, 128), None), ("7-pt 512x256x96", 7, (512, 256, 96), None)]
if "first256" in sys.argv:  # (the 27-pt 256^3 case first and last: its time depends on what the process has allocated and freed before)
    big = [("27-pt 256^3", 27, (256, 256, 256), None)] + big + [("27-pt 256^3", 27, (256, 256, 256), None)]
if "quick" not in sys.argv:
    big += [("27-pt 256^3", 27, (256, 256, 256), None), ("7-pt 256^3", 7, (256, 256, 256), None), ("27-pt 512x512x64 (config 3's slab)", 27, (512, 512, 512), 64)]
for name, st, dims, planes in big:
    t0 = time.perf_counter()
    if planes:  # the first `planes` planes of the cube, columns cut at the slab: one rank's diagonal block
        n = dims[0]
        N = n * n * planes
        ai, aj, aa = bench.assemble(ks, st, dims, 0, N)
        keep = aj < N
        rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(ai))[keep]
        aj, aa = aj[keep], aa[keep]
        ai = np.zeros(N + 1, np.int32)
        ai[1:] = np.cumsum(np.bincount(rows, minlength=N))
        del rows, keep
        shp = (n, n, planes)
    else:
        N = dims[0] * dims[1] * dims[2]
        ai, aj, aa = bench.assemble(ks, st, dims, 0, N)
        shp = dims
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    nnz = int(ai[-1])
    del ai, aj, aa
    b = 1.0 + (np.arange(N) % 17) / 17.0
    res = {}
    for mode in (("box",) if "timeonly" in sys.argv else ("box", "strand", "dep")):
        if mode == "dep" and N > 40e6:
            continue
        try:
            x, used = sor(A, N, b, LSYM | ZERO, 1.0, 0.0, mode)
        except Exception as e:  # noqa: BLE001
            print("%s mode %s: ERROR %s" % (name, mode, str(e)[:200]), flush=True)
            continue
        B, X = _lib.DVec(N, b), _lib.DVec(N)
        os.environ["HIPX_SOR_MODE"] = mode
        reps = 5 if mode != "dep" else 2
        _lib.chk(hx.hipxEventRecord(e0))
        for _ in range(reps):
            _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, LSYM | ZERO, 0.0, 1, 1, X.ptr))
        _lib.chk(hx.hipxEventRecord(e1))
        ms = C.c_float()
        _lib.chk(hx.hipxEventElapsedMs(e0, e1, C.byref(ms)))
        os.environ.pop("HIPX_SOR_MODE", None)
        B.free()
        X.free()
        res[mode] = (x, ms.value / reps, used)
    ref = "dep" if "dep" in res else "strand"
    line = "%-36s rows %10d:" % (name, N)
    for mode, (x, ms, used) in res.items():
        line += "  %s %.3f ms (mode %d)" % (mode, ms, used)
    if "box" in res and ref in res:
        ok = np.array_equal(res["box"][0], res[ref][0])
        line += "  box vs %s: %s" % (ref, "bit-identical" if ok else "DIFFERENT " + where(res["box"][0] != res[ref][0], *shp))
    if "box" in res:
        line += "  [box: %.0f GB/s on the 32 B/row it moves]" % (32.0 * N / (res["box"][1] * 1e-3) / 1e9)
    print(line + "  (set-up + runs %.1f s)" % (time.perf_counter() - t0), flush=True)
    _lib.mat_destroy(A)
