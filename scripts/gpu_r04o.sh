#!/bin/bash
# round-4 run O: SELL-64 with one column code per run of three (FEM rows): bit-exactness and same-box A/B on the config-4 stand-in
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r04o
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_mat.py "tests/test_gpu_scale_parity.py::test_config4_surrogate_full_vector_bit_exact" -m gpu -q --timeout 600 -p no:cacheprovider -rf -k "sell or surrogate or long" > gpurun_out/${T}_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/${T}_pytest.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/${T}_pytest.log | head
cat > /tmp/sell_ab.py <<'PY'
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
from petsc_amd import _lib
from surrogates import flan_surrogate
hx = _lib.init(0)
ai, aj, aa = flan_surrogate()
N, nnz = len(ai) - 1, int(ai[-1])
A = _lib.mat_create_csr(N, N, ai, aj, aa)
X, Y = _lib.DVec(N, 1.0 + (np.arange(N) % 17) / 17.0), _lib.DVec(N)
kb = C.create_string_buffer(256); _lib.chk(hx.hipxMatGetSpMVKernel(A, kb, 256))
for _ in range(5): _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
_lib.chk(hx.hipxProfileSpMV(1))
for _ in range(100): _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
cnt, tot = C.c_int(), C.c_double(); _lib.chk(hx.hipxProfileSpMVGet(C.byref(cnt), C.byref(tot)))
ms = tot.value / cnt.value; b = 12 * nnz + 4 * (N + 1) + 16 * N
print("%s  %.4f ms  %.2f TB/s on CSR bytes = %.3f of 8 TB/s" % (kb.value.decode()[:40], ms, b / ms / 1e9, b / ms / 1e9 / 8.0))
PY
for rep in 1 2; do
echo "triple-run codes:"; python /tmp/sell_ab.py 2>/dev/null
echo "plain codes:"; HIPX_SELL_NOTRI=1 python /tmp/sell_ab.py 2>/dev/null
done
echo "total ${SECONDS}s"
