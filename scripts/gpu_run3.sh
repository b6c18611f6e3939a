#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
R="$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x > gpurun_out/pytest3.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest3.log
timeout 600 python scripts/spmv_variants.py 256 7 > gpurun_out/variants3_7pt_256.log 2>&1
timeout 600 python scripts/spmv_variants.py 128 27 > gpurun_out/variants3_27pt_128.log 2>&1
timeout 600 python scripts/spmv_variants.py 200 27 > gpurun_out/variants3_27pt_200.log 2>&1
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench3.log 2>&1
timeout 300 python bench.py --steps 100 --warmup 10 --fused 1 --no-cpu-baseline > gpurun_out/bench3_fused.log 2>&1
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc3_fetch" -o pmc -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/pmc3_fetch.log" 2>&1
cd "$R"
tail -4 gpurun_out/pytest3.log; cat gpurun_out/variants3_7pt_256.log gpurun_out/variants3_27pt_128.log gpurun_out/variants3_27pt_200.log; tail -1 gpurun_out/bench3.log; tail -1 gpurun_out/bench3_fused.log
