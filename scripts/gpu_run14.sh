#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python scripts/spmv_variants.py 256 7 25,1025,2025,3025,23 2>&1 | grep "variant" > gpurun_out/probe7.log
timeout 300 python scripts/spmv_variants.py 160 27 25,1025,2025,3025 2>&1 | grep "variant" > gpurun_out/probe27.log
cat gpurun_out/probe7.log gpurun_out/probe27.log
