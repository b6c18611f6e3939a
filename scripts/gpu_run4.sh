#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x > gpurun_out/pytest5.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest5.log
timeout 600 python scripts/spmv_variants.py 256 7 > gpurun_out/variants5_7pt_256.log 2>&1
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench5.log 2>&1
timeout 300 python bench.py --steps 100 --warmup 10 --fused 1 --no-cpu-baseline > gpurun_out/bench5_fused.log 2>&1
tail -3 gpurun_out/pytest5.log; cat gpurun_out/variants5_7pt_256.log; tail -1 gpurun_out/bench5.log | cut -c1-300; tail -1 gpurun_out/bench5_fused.log | cut -c1-300
