#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats of the default bench, then FETCH_SIZE / WRITE_SIZE passes (separate runs,
# --kernel-trace only).  Usage: bash scripts/gpu_profile.sh <tag>   (outputs under gpurun_out/<tag>_*)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; T=${1:-prof}
mkdir -p gpurun_out
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${T}_stats" -o stats -- python "$R/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > "$R/gpurun_out/${T}_stats.log" 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/${T}_fetch" -o pmc -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/${T}_fetch.log" 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/${T}_write" -o pmc -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/${T}_write.log" 2>&1
cd "$R"
rm -f gpurun_out/${T}_*/stats_kernel_trace.csv gpurun_out/${T}_*/pmc_kernel_trace.csv   # large, not needed
head -8 gpurun_out/${T}_stats/stats_kernel_stats.csv | cut -c1-200
