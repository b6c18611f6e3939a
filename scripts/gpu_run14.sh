#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_mat.py -q -m gpu 2>&1 | tail -3 > gpurun_out/mat_tests.log
HIPX_VD_RPT=4 timeout 300 python -m pytest tests/test_gpu_mat.py -q -m gpu 2>&1 | tail -3 >> gpurun_out/mat_tests.log
HIPX_VD_RPT=1 timeout 300 python -m pytest tests/test_gpu_mat.py -q -m gpu 2>&1 | tail -3 >> gpurun_out/mat_tests.log
timeout 200 python scripts/spmv_variants.py 256 7 25,3025,23 2>&1 | grep "variant" > gpurun_out/probe7.log
timeout 200 python scripts/spmv_variants.py 160 27 25,3025,22 2>&1 | grep "variant" >> gpurun_out/probe7.log
HIPX_VD_RPT=1 timeout 200 python scripts/spmv_variants.py 256 7 25 2>&1 | grep "variant" >> gpurun_out/probe7.log
HIPX_VD_RPT=4 timeout 200 python scripts/spmv_variants.py 256 7 25 2>&1 | grep "variant" >> gpurun_out/probe7.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_vd.log 2>&1
cat gpurun_out/mat_tests.log gpurun_out/probe7.log; tail -1 gpurun_out/bench_vd.log | cut -c1-330
