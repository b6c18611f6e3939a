#!/bin/bash
# Round 2, GPU call AC: split SOR kernel (C / F / loader waves).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
echo "== slab proxy (split)"; timeout 600 python scripts/config3_slab_proxy.py 2>&1 | grep -v amdgpu.ids | tee "$O/r2ac_slab.log" | grep "SOR local symmetric sweep \[strand\|bit for bit"
HIPX_SOR_DEBUG=1 HIPX_SOR_DEBUG_DUMP="$O/r2ac_sorpanels" timeout 300 python scripts/config3_slab_proxy.py 2>&1 | grep "hipx sor\]   per panel\|hipx sor\] strand KIND . done\|F wave" | head -8 | cut -c1-400 | tee "$O/r2ac_sorstats.log"
