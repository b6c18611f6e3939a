#!/bin/bash
# Round 2, GPU call S: SOR compute wave sleeps when idle, raised priority; 1 vs 2 workgroups per CU.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
echo "== sor tests"; timeout 900 python -m pytest tests/test_gpu_sor.py -x -q --timeout=300 -p no:cacheprovider > "$O/r2s_sor.log" 2>&1; tail -2 "$O/r2s_sor.log" | cut -c1-300
echo "== slab proxy (2 per CU)"; timeout 600 python scripts/config3_slab_proxy.py 2>&1 | grep -v amdgpu.ids | tee "$O/r2s_slab.log" | grep SOR
echo "== slab proxy (1 per CU)"; HIPX_SOR_WG_PER_CU=1 timeout 600 python scripts/config3_slab_proxy.py 2>&1 | grep -v amdgpu.ids | tee "$O/r2s_slab1.log" | grep "SOR local symmetric sweep \[strand"
HIPX_SOR_DEBUG=1 HIPX_SOR_DEBUG_DUMP="$O/r2s_sorpanels" timeout 300 python scripts/config3_slab_proxy.py 2>&1 | grep "hipx sor\]   per panel\|hipx sor\] strand KIND . done" | head -4 | cut -c1-600 | tee "$O/r2s_sorstats.log"
echo "== 7pt 256 sor"; timeout 300 python bench.py --ksp gmres --pc sor --stencil 7 --grid 256 --steps 30 --warmup 3 --quick 2>&1 | grep "^{" | tail -1 | cut -c1-300 | tee "$O/r2s_sor7.log"
