#!/bin/bash
# Round 2, GPU call A: new kernels first (SOR strands, row-template SpMV) with -x so a systematic failure stops early, then the
# whole GPU suite, the slab proxy timings, the kernel-variant timings, the default bench and a rocprofv3 kernel-stats pass.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
rm -f "$O/parity_measured.json"
rocm-smi --showproductname 2>/dev/null | head -8 > "$O/r2a_device.log"; nproc >> "$O/r2a_device.log"; lscpu | head -20 >> "$O/r2a_device.log"
echo "== sor tests"; timeout 900 python -m pytest tests/test_gpu_sor.py -x -q --timeout=300 -p no:cacheprovider > "$O/r2a_sor.log" 2>&1; tail -5 "$O/r2a_sor.log"
echo "== mat tests"; timeout 600 python -m pytest tests/test_gpu_mat.py -x -q --timeout=300 -p no:cacheprovider > "$O/r2a_mat.log" 2>&1; tail -3 "$O/r2a_mat.log"
echo "== slab proxy"; timeout 600 python scripts/config3_slab_proxy.py > "$O/r2a_slab.log" 2>&1; tail -12 "$O/r2a_slab.log"
echo "== spmv variants"; timeout 300 python scripts/spmv_variants.py 256 7 0,25,23,22,1 > "$O/r2a_spmv7.log" 2>&1; tail -12 "$O/r2a_spmv7.log"
timeout 300 python scripts/spmv_variants.py 160 27 0,25,22 > "$O/r2a_spmv27.log" 2>&1; tail -5 "$O/r2a_spmv27.log"
echo "== rest of the suite"; timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider --deselect tests/test_gpu_sor.py --deselect tests/test_gpu_mat.py > "$O/r2a_pytest.log" 2>&1; tail -15 "$O/r2a_pytest.log"
echo "== bench"; timeout 900 python bench.py > "$O/r2a_bench.json" 2> "$O/r2a_bench.err"; cat "$O/r2a_bench.json" | cut -c1-1500
echo "== gmres+sor bench"; timeout 600 python bench.py --ksp gmres --pc sor --stencil 27 --grid 256 --steps 60 --warmup 5 --quick > "$O/r2a_bench_gmres_sor.json" 2> "$O/r2a_bench_gmres_sor.err"; cat "$O/r2a_bench_gmres_sor.json" | cut -c1-600
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/r2a_stats" -o stats -- python "$R/bench.py" --steps 50 --warmup 5 --quick > "$O/r2a_stats.log" 2>&1
rm -f "$O"/r2a_stats/*/stats_kernel_trace.csv "$O"/r2a_stats/stats_kernel_trace.csv
find "$O/r2a_stats" -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-220
