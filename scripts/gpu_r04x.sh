#!/bin/bash
# round-4 run X: the cooperative form of the point dependency-driven sweep (sor_dep_coop_kernel): parity, then timings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_sor.py tests/test_gpu_inode.py -x -q -m gpu 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -4
timeout 1200 python - <<'PY'
import json, os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import bench
from petsc_amd import _lib
hx = _lib.init(0)
_, ks = _lib.load()
for coop in ("1", "0"):
    os.environ["HIPX_SOR_DEP_COOP"] = coop
    r = bench.leg_sor_arbitrary_values(hx, _lib, ks)
    print("27-pt 256^3 arbitrary values, HIPX_SOR_DEP_COOP=%s: strand %.2f ms, level-ordered %.2f ms, same bits %s" % (coop, r["strand_streamed_coefficients_ms"], r["level_ordered_ms"], r["bit_identical_to_level_ordered"]), flush=True)
os.environ["HIPX_MAT_NO_INODE"] = "1"
from surrogates import flan_surrogate_spd
ai, aj, aa = flan_surrogate_spd()
N = len(ai) - 1
A = _lib.mat_create_csr(N, N, ai, aj, aa)
B, X = _lib.DVec(N, np.random.default_rng(1).standard_normal(N)), _lib.DVec(N)
ref = None
for coop in ("1", "0"):
    os.environ["HIPX_SOR_DEP_COOP"] = coop
    for k in range(2):
        _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 16 | 12, 0.0, 1, 1, X.ptr))
    _lib.chk(hx.hipxDeviceSynchronize())
    t0 = time.perf_counter()
    for _ in range(5):
        _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 16 | 12, 0.0, 1, 1, X.ptr))
    _lib.chk(hx.hipxDeviceSynchronize())
    x = X.get()
    if ref is None: ref = x
    print("elasticity stand-in as a POINT matrix (-mat_no_inode), coop=%s: %.2f ms per symmetric sweep, same bits %s" % (coop, (time.perf_counter() - t0) / 5 * 1e3, np.array_equal(x, ref)), flush=True)
PY
