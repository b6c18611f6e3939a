#!/bin/bash
# round-3 run X: the evidence run with the march form of the template SpMV -- full GPU suite, smoke, the default bench line (all legs), rocprofv3 kernel
# stats of the same command's timed legs (the SOR kernels did not change since run W2: their slab profile is not repeated).  Usage: bash scripts/gpu_r03x.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03x
SECONDS=0
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rf > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/${T}_smoke.log
S0=$SECONDS
timeout 1200 python bench.py > gpurun_out/${T}_bench.log 2>gpurun_out/${T}_bench.err
echo "default bench: $((SECONDS - S0)) s" >> gpurun_out/${T}_bench.err
R="$GRAFT_REPO_ROOT"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${T}_stats" -o stats -- python "$R/bench.py" --no-traffic --no-plugin --no-cpu-baseline --no-other --no-general > "$R/gpurun_out/${T}_stats.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${T}_stats_all" -o stats -- python "$R/bench.py" --no-traffic --no-plugin --no-cpu-baseline > "$R/gpurun_out/${T}_stats_all.log" 2>&1
cd "$R"
find gpurun_out/${T}_stats gpurun_out/${T}_stats_all -name "*kernel_trace.csv" -delete 2>/dev/null
tail -5 gpurun_out/${T}_pytest.log | cut -c1-300; tail -2 gpurun_out/${T}_smoke.log; tail -1 gpurun_out/${T}_bench.err
tail -1 gpurun_out/${T}_bench.log | cut -c1-600
f=$(find gpurun_out/${T}_stats -name "*kernel_stats.csv" | head -1); echo "== $f"; head -7 "$f" | cut -c1-220
echo "total ${SECONDS}s"
