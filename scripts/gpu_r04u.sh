#!/bin/bash
# round-4 run U: the cooperative inode kernel (16 lanes per node): parity, then A/B of the PCSOR application on the config-4 stand-in
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_inode.py -x -q -m gpu 2>&1 | tail -4
HIPX_SOR_INODE_COOP=0 timeout 900 python -m pytest tests/test_gpu_inode.py -x -q -m gpu -k "bit_exact and 900" 2>&1 | tail -2
timeout 1500 python - <<'PY'
import ctypes as C, os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import numpy as np
from surrogates import flan_surrogate_spd
from petsc_amd import _lib
hx = _lib.init(0)
ai, aj, aa = flan_surrogate_spd()
N = len(ai) - 1
A = _lib.mat_create_csr(N, N, ai, aj, aa)
b = np.random.default_rng(1).standard_normal(N)
B, X = _lib.DVec(N, b), _lib.DVec(N)
def timeit(reps=10):
    _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 16 | 12, 0.0, 1, 1, X.ptr))
    _lib.chk(hx.hipxDeviceSynchronize())
    t0 = time.perf_counter()
    for _ in range(reps):
        _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 16 | 12, 0.0, 1, 1, X.ptr))
    _lib.chk(hx.hipxDeviceSynchronize())
    return (time.perf_counter() - t0) / reps * 1e3
t0 = time.perf_counter()
ms = timeit()
print("coop (16 waves/CU): %.2f ms per symmetric sweep (first call incl. set-up %.1f s)" % (ms, time.perf_counter() - t0))
ref = X.get()
PY
for w in 4 8 32; do HIPX_SOR_INODE_COOP_WAVES_PER_CU=$w timeout 600 python - <<'PY'
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
for k in range(2):
    _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 16 | 12, 0.0, 1, 1, X.ptr))
_lib.chk(hx.hipxDeviceSynchronize())
t0 = time.perf_counter()
for _ in range(10):
    _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 16 | 12, 0.0, 1, 1, X.ptr))
_lib.chk(hx.hipxDeviceSynchronize())
print("coop waves/CU %s: %.2f ms" % (os.environ.get("HIPX_SOR_INODE_COOP_WAVES_PER_CU"), (time.perf_counter() - t0) / 10 * 1e3))
PY
done
