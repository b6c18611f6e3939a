#!/bin/bash
# Round 2, GPU call Z: per-iteration log of lane 0 of SOR panels 8 / 16 / 43, looking for the stalled iterations.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
for p in 8 16 43 24; do
HIPX_SOR_DEBUG=1 HIPX_SOR_TRACE_PANEL=$p HIPX_SOR_TRACE_ROWS=0 HIPX_SOR_TRACE_LANE=0 HIPX_SOR_TRACE_IT0=400 HIPX_SOR_DEBUG_DUMP="$O/r2z_p${p}" timeout 300 python scripts/config3_slab_proxy.py 2>&1 | grep "hipx sor\] strand KIND 0 done" | head -1 | cut -c1-120
rm -f "$O/r2z_p${p}_trace1.bin" "$O/r2z_p${p}_kind1.txt"
done
