#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 python -m pytest tests/test_gpu_halo.py tests/test_gpu_vec.py -m gpu -q --timeout 60 -p no:cacheprovider -x > gpurun_out/pytest24.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest24.log
tail -12 gpurun_out/pytest24.log
