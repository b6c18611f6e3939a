#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 python -m pytest tests/test_gpu_mat.py -m gpu -q --timeout 60 -p no:cacheprovider -x > gpurun_out/pytest28.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest28.log
tail -3 gpurun_out/pytest28.log
timeout 120 python scripts/spmv_variants.py 256 7 2>&1 | grep variant
timeout 120 python scripts/spmv_variants.py 128 27 2>&1 | grep variant
