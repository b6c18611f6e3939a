#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mat.py -m gpu -q --timeout 600 -p no:cacheprovider -x > gpurun_out/pytest10.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest10.log
timeout 600 python scripts/spmv_variants.py 256 7 > gpurun_out/variants10_7pt_256.log 2>&1
timeout 600 python scripts/spmv_variants.py 128 27 > gpurun_out/variants10_27pt_128.log 2>&1
tail -3 gpurun_out/pytest10.log; cat gpurun_out/variants10_7pt_256.log gpurun_out/variants10_27pt_128.log | grep -v amdgpu
