#!/bin/bash
# Round 2, GPU call Y: per-iteration log of lane 0 of SOR panel 8.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
for l in 0; do
HIPX_SOR_DEBUG=1 HIPX_SOR_TRACE_PANEL=8 HIPX_SOR_TRACE_ROWS=0 HIPX_SOR_TRACE_LANE=$l HIPX_SOR_TRACE_IT0=500 HIPX_SOR_DEBUG_DUMP="$O/r2y_l${l}" timeout 300 python scripts/config3_slab_proxy.py 2>&1 | grep "hipx sor\] strand KIND 0 done" | head -1 | cut -c1-200
rm -f "$O/r2y_l${l}_trace1.bin" "$O/r2y_l${l}_kind1.txt"
done
