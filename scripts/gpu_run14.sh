#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sor.py -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/pytest14.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest14.log
timeout 600 python scripts/gmres_sor_timing.py 128 27 > gpurun_out/gmres_sor14_27_128.log 2>&1
timeout 600 python scripts/gmres_sor_timing.py 192 7 > gpurun_out/gmres_sor14_7_192.log 2>&1
tail -5 gpurun_out/pytest14.log; grep -v amdgpu gpurun_out/gmres_sor14_27_128.log gpurun_out/gmres_sor14_7_192.log
