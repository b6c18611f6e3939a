#!/bin/bash
# Round 2, GPU call AB: staging bounded by the slowest consumer.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
echo "== sor tests"; timeout 900 python -m pytest tests/test_gpu_sor.py -x -q --timeout=300 -p no:cacheprovider > "$O/r2ab_sor.log" 2>&1; tail -2 "$O/r2ab_sor.log" | cut -c1-300
echo "== slab proxy"; timeout 600 python scripts/config3_slab_proxy.py 2>&1 | grep -v amdgpu.ids | tee "$O/r2ab_slab.log" | grep SOR
HIPX_SOR_DEBUG=1 HIPX_SOR_DEBUG_DUMP="$O/r2ab_sorpanels" timeout 300 python scripts/config3_slab_proxy.py 2>&1 | grep "hipx sor\]   per panel\|hipx sor\] strand KIND . done" | head -4 | cut -c1-400 | tee "$O/r2ab_sorstats.log"
