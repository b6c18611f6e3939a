#!/bin/bash
# round-4 run Z10: the RUN form with 4 waves per run and many more runs resident
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
HIPX_SOR_INODE_RUN=1 timeout 100 python -m pytest tests/test_gpu_inode.py -x -q -m gpu -k "bit_exact and 900" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -2
timeout 200 python - <<'PY'
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
for runmode, blocks in (("0", "0"), ("1", "2048"), ("1", "1792"), ("1", "1024"), ("1", "512")):
    os.environ["HIPX_SOR_INODE_RUN"] = runmode
    os.environ["HIPX_SOR_INODE_RUN_BLOCKS"] = blocks
    for k in range(2):
        rc = hx.hipxMatSOR(A, B.ptr, 1.0, 16 | 12, 0.0, 1, 1, X.ptr)
    _lib.chk(hx.hipxDeviceSynchronize())
    t0 = time.perf_counter()
    for _ in range(3):
        rc = hx.hipxMatSOR(A, B.ptr, 1.0, 16 | 12, 0.0, 1, 1, X.ptr)
    _lib.chk(hx.hipxDeviceSynchronize())
    x = X.get()
    if ref is None: ref = x
    print("HIPX_SOR_INODE_RUN=%s blocks %s: %.2f ms per symmetric sweep  same bits %s rc %d" % (runmode, blocks, (time.perf_counter() - t0) / 3 * 1e3, np.array_equal(x, ref), rc), flush=True)
PY
