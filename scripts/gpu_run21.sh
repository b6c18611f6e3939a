#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_ksp.py tests/test_gpu_mat.py -m gpu -q --timeout 60 -p no:cacheprovider -x > gpurun_out/pytest22.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest22.log
tail -3 gpurun_out/pytest22.log
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench22.log 2>&1; tail -1 gpurun_out/bench22.log | cut -c1-260
timeout 200 python bench.py --no-cpu-baseline --fused 0 > gpurun_out/bench22_unfused.log 2>&1; tail -1 gpurun_out/bench22_unfused.log | cut -c1-260
