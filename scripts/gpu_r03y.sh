#!/bin/bash
# round-3 run Y: SELL-64 with values in pairs (16-byte loads): tests + the config-4 stand-in's SpMV and solve.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03y
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_mat.py tests/test_gpu_scale_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -k "sell or surrogate or variant or long" > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s: $(tail -1 gpurun_out/${T}_pytest.log)"
python - <<'PY'
import sys, json
sys.path.insert(0, '.')
import bench
from petsc_amd import _lib
hx = _lib.init(0)
print(json.dumps(bench.leg_surrogate_spmv(hx, _lib)["roofline_longrow"]))
PY
echo "total ${SECONDS}s"
