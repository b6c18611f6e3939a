#!/bin/bash
# round-4 run Z2: the poll back-off of the cooperative inode sweep; the driver's own bench command (--steps 20 --warmup 5)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1200 python - <<'PY'
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
for sl in ("8", "4", "2", "1", "0", "8"):
    os.environ["HIPX_SOR_COOP_SLEEP"] = sl
    for k in range(2):
        _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 16 | 12, 0.0, 1, 1, X.ptr))
    _lib.chk(hx.hipxDeviceSynchronize())
    t0 = time.perf_counter()
    for _ in range(10):
        _lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 16 | 12, 0.0, 1, 1, X.ptr))
    _lib.chk(hx.hipxDeviceSynchronize())
    x = X.get()
    if ref is None: ref = x
    print("poll back-off s_sleep %s: %.2f ms per symmetric sweep  same bits %s" % (sl, (time.perf_counter() - t0) / 10 * 1e3, np.array_equal(x, ref)), flush=True)
PY
for rep in 1 2; do
python bench.py --gpus 1 --steps 20 --warmup 5 --quick 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 20 warmup 5: %.1f it/s  %.4f ms/it  product %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
python bench.py --gpus 1 --quick 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 200 warmup 20: %.1f it/s  %.4f ms/it  product %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
