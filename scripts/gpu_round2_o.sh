#!/bin/bash
# Round 2, GPU call O: per-row traces of SOR panels 16 (plane 2, y-block 0) and 8.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
for p in 16 8; do
HIPX_SOR_DEBUG=1 HIPX_SOR_TRACE_PANEL=$p HIPX_SOR_DEBUG_DUMP="$O/r2o_p${p}" timeout 300 python scripts/config3_slab_proxy.py 2>&1 | grep "hipx sor\] strand KIND 0 done" | head -1 | cut -c1-200
rm -f "$O/r2o_p${p}_trace1.bin" "$O/r2o_p${p}_kind1.txt"
done
ls -la $O/r2o_*
