#!/bin/bash
# cg_fused_kernel with four elements per stream in flight: tests + per-kernel averages of the headline CG loop (rocprofv3 stats)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_ksp.py tests/test_gpu_vec.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r03z_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s: $(tail -1 gpurun_out/r03z_pytest.log)"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03z_stats" -o stats -- python "$GRAFT_REPO_ROOT/bench.py" --no-traffic --no-plugin --no-cpu-baseline --no-other --no-general > "$GRAFT_REPO_ROOT/gpurun_out/r03z_stats.log" 2>&1)
find gpurun_out/r03z_stats -name "*kernel_trace.csv" -delete
f=$(find gpurun_out/r03z_stats -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-60,200-330
tail -1 gpurun_out/r03z_stats.log | cut -c1-200
echo "total ${SECONDS}s"
