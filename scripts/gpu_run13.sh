#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_halo.py tests/test_gpu_plugin.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest13.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest13.log
timeout 900 python scripts/gmres_sor_timing.py 128 27 > gpurun_out/gmres_sor_27_128.log 2>&1
timeout 900 python scripts/gmres_sor_timing.py 192 7 > gpurun_out/gmres_sor_7_192.log 2>&1
tail -3 gpurun_out/pytest13.log; grep -v amdgpu gpurun_out/gmres_sor_27_128.log gpurun_out/gmres_sor_7_192.log
