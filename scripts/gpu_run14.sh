#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_halo.py tests/test_gpu_ksp.py tests/test_gpu_vec.py -q -m gpu 2>&1 | tail -25 > gpurun_out/ksp_tests.log
cat gpurun_out/ksp_tests.log
