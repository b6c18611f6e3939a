#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats of the default bench (timed legs only: the plugin / PMC / CPU-baseline legs start
# their own processes), kernel stats + FETCH_SIZE / WRITE_SIZE of the SOR sweeps on the config-3 slab (separate passes,
# --kernel-trace only).  Usage: bash scripts/gpu_profile.sh <tag>   (outputs under gpurun_out/<tag>_*)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; T=${1:-prof}
mkdir -p gpurun_out
cd /tmp
# the headline configuration alone (the legs for the other configurations and the general-kernel leg launch the same kernel names on
# other operators: their launches would be averaged in)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${T}_stats" -o stats -- python "$R/bench.py" --no-traffic --no-plugin --no-cpu-baseline --no-other --no-general > "$R/gpurun_out/${T}_stats.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${T}_stats_all" -o stats -- python "$R/bench.py" --no-traffic --no-plugin --no-cpu-baseline > "$R/gpurun_out/${T}_stats_all.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${T}_sor_stats" -o stats -- python "$R/scripts/config3_slab_proxy.py" > "$R/gpurun_out/${T}_sor_stats.log" 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/${T}_sor_fetch" -o pmc -- python "$R/scripts/config3_slab_proxy.py" > "$R/gpurun_out/${T}_sor_fetch.log" 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/${T}_sor_write" -o pmc -- python "$R/scripts/config3_slab_proxy.py" > "$R/gpurun_out/${T}_sor_write.log" 2>&1
cd "$R"
python scripts/pmc_summary.py --sor "gpurun_out/${T}_sor_fetch" "gpurun_out/${T}_sor_write" > "gpurun_out/${T}_sor_traffic.json" 2> "gpurun_out/${T}_sor_traffic.err"
find gpurun_out/${T}_stats gpurun_out/${T}_stats_all gpurun_out/${T}_sor_stats gpurun_out/${T}_sor_fetch gpurun_out/${T}_sor_write -name "*kernel_trace.csv" -delete 2>/dev/null  # large, not needed
for d in stats sor_stats; do f=$(find gpurun_out/${T}_$d -name "*kernel_stats.csv" | head -1); echo "== $f"; head -8 "$f" | cut -c1-200; done
cat "gpurun_out/${T}_sor_traffic.json" | cut -c1-600
