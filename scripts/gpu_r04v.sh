#!/bin/bash
# round-4 run V: how many workgroups the cooperative inode sweep wants (pollers against publishers)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1500 python - <<'PY'
import ctypes as C, os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from surrogates import flan_surrogate_spd
from petsc_amd import _lib
hx = _lib.init(0)
ai, aj, aa = flan_surrogate_spd()
N = len(ai) - 1
A = _lib.mat_create_csr(N, N, ai, aj, aa)
B, X = _lib.DVec(N, np.random.default_rng(1).standard_normal(N)), _lib.DVec(N)
ref = None
for blocks in (256, 32, 64, 96, 128, 192, 256, 320, 384):
    os.environ["HIPX_SOR_INODE_COOP_BLOCKS"] = str(blocks)
    for k in range(2):
        _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 16 | 12, 0.0, 1, 1, X.ptr))
    _lib.chk(hx.hipxDeviceSynchronize())
    t0 = time.perf_counter()
    for _ in range(10):
        _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 16 | 12, 0.0, 1, 1, X.ptr))
    _lib.chk(hx.hipxDeviceSynchronize())
    x = X.get()
    if ref is None: ref = x
    print("blocks %4d: %.2f ms per symmetric sweep  same bits %s" % (blocks, (time.perf_counter() - t0) / 10 * 1e3, np.array_equal(x, ref)), flush=True)
PY
