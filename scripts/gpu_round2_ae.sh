#!/bin/bash
# Round 2, GPU call AE: split kernel for the forward sweep only (default) vs all vs none; SOR tests in the default mode and with HIPX_SOR_SPLIT=2.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
echo "== sor tests (default)"; timeout 900 python -m pytest tests/test_gpu_sor.py -x -q --timeout=300 -p no:cacheprovider > "$O/r2ae_sor.log" 2>&1; tail -2 "$O/r2ae_sor.log" | cut -c1-300
echo "== sor tests (split 2)"; HIPX_SOR_SPLIT=2 timeout 900 python -m pytest tests/test_gpu_sor.py -x -q --timeout=300 -p no:cacheprovider -k "strand or scale or slab" > "$O/r2ae_sor2.log" 2>&1; tail -2 "$O/r2ae_sor2.log" | cut -c1-300
for m in 1 2 0; do echo "== slab proxy HIPX_SOR_SPLIT=$m"; HIPX_SOR_SPLIT=$m timeout 600 python scripts/config3_slab_proxy.py 2>&1 | grep "SOR local symmetric sweep \[strand\|bit for bit"; done
echo "== gmres+sor 27pt 256"; timeout 600 python bench.py --ksp gmres --pc sor --stencil 27 --grid 256 --steps 60 --warmup 5 --quick 2>/dev/null | tee "$O/r2ae_bench_gmres_sor.json" | cut -c1-200
