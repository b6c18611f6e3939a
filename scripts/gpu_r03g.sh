#!/bin/bash
# round-3 run G: strand SOR with streamed coefficients (arbitrary values on a stencil pattern): parity tests, timings.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03g
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_sor.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "variable or default_schedule" > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
tail -15 gpurun_out/${T}_pytest.log | cut -c1-400
{
timeout 300 python scripts/sor_var_timing.py 7 256
timeout 300 python scripts/sor_var_timing.py 27 256
HIPX_SOR_VAR_RING=4 timeout 300 python scripts/sor_var_timing.py 27 256
HIPX_SOR_VAR_RING=4 timeout 300 python scripts/sor_var_timing.py 7 256
HIPX_SOR_VAR_RING=16 timeout 300 python scripts/sor_var_timing.py 7 256
timeout 300 python scripts/sor_var_timing.py 27 512 64
HIPX_SOR_VAR_RING=4 timeout 300 python scripts/sor_var_timing.py 27 512 64
} > gpurun_out/${T}_timing.log 2>&1
cat gpurun_out/${T}_timing.log | cut -c1-300
echo "total ${SECONDS}s"
