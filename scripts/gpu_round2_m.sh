#!/bin/bash
# Round 2, GPU call M: per-row trace of one SOR panel (plane 5, y-block 3) and its loader.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
HIPX_SOR_DEBUG=1 HIPX_SOR_TRACE_PANEL=43 HIPX_SOR_DEBUG_DUMP="$O/r2m_sorpanels" timeout 300 python scripts/config3_slab_proxy.py 2>&1 | grep "hipx sor\]   per panel\|hipx sor\] strand KIND . done" | head -4 | cut -c1-600 | tee "$O/r2m_sorstats.log"
ls -la $O/r2m_*
