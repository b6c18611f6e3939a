#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 90 python -m pytest tests/test_gpu_sor.py -m gpu -q --timeout 30 -p no:cacheprovider -x > gpurun_out/pytest18.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest18.log
tail -3 gpurun_out/pytest18.log
bash scripts/gpu_run17.sh
