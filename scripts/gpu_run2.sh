#!/bin/bash
# second GPU pass: full parity suite, SpMV geometry sweep, rocprof kernel stats (CSV) and PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
R="$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest2.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest2.log
timeout 600 python scripts/spmv_variants.py 256 7 > gpurun_out/variants_7pt_256.log 2>&1
timeout 600 python scripts/spmv_variants.py 128 27 > gpurun_out/variants_27pt_128.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof2" -o stats -- python "$R/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > "$R/gpurun_out/rocprof2.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_fetch" -o pmc -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_write" -o pmc -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/pmc_write.log" 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_l2" -o pmc -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/pmc_l2.log" 2>&1
cd "$R"
find gpurun_out -name "*.csv" | head -30
tail -4 gpurun_out/pytest2.log; cat gpurun_out/variants_7pt_256.log
