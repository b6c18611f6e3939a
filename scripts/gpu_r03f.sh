#!/bin/bash
# round-3 run F: the 64-bit-PetscInt flavour (tests + one operator beyond 2^31 nonzeros through the drop-in), the final default bench
# line, rocprofv3 stats of the headline configuration alone.  Usage: bash scripts/gpu_r03f.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03f
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_plugin_int64.py tests/test_gpu_scale_parity.py tests/test_gpu_plugin.py -m gpu -q --timeout 900 -p no:cacheprovider -rf > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
S0=$SECONDS
timeout 900 bash scripts/int64_beyond_2g.sh 432 > gpurun_out/${T}_int64.log 2>&1
echo "int64 beyond 2^31: $((SECONDS - S0)) s" >> gpurun_out/${T}_int64.log
S0=$SECONDS
timeout 1200 python bench.py > gpurun_out/${T}_bench.log 2>gpurun_out/${T}_bench.err
echo "default bench: $((SECONDS - S0)) s" >> gpurun_out/${T}_bench.err
bash scripts/gpu_profile.sh ${T} > gpurun_out/${T}_profile.log 2>&1
tail -5 gpurun_out/${T}_pytest.log | cut -c1-300
cat gpurun_out/${T}_int64.log | tail -8
tail -1 gpurun_out/${T}_bench.err; tail -1 gpurun_out/${T}_bench.log | cut -c1-300
head -12 gpurun_out/${T}_stats/stats_kernel_stats.csv | cut -c1-220
echo "total ${SECONDS}s"
