#!/bin/bash
# Round 2, GPU call T: far polls at system scope (experiment).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
echo "== agent"; timeout 600 python scripts/config3_slab_proxy.py 2>&1 | grep "SOR local symmetric sweep \[strand\|bit for bit"
echo "== system"; HIPX_SOR_POLL=sys timeout 600 python scripts/config3_slab_proxy.py 2>&1 | grep "SOR local symmetric sweep \[strand\|bit for bit"
HIPX_SOR_POLL=sys HIPX_SOR_DEBUG=1 HIPX_SOR_DEBUG_DUMP="$O/r2t_sorpanels" timeout 300 python scripts/config3_slab_proxy.py 2>&1 | grep "hipx sor\] strand KIND . done" | head -2 | cut -c1-300
