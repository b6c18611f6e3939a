#!/bin/bash
# Round 2, GPU call AA: one workgroup per CU: are there slow panels?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
HIPX_SOR_WG_PER_CU=1 HIPX_SOR_DEBUG=1 HIPX_SOR_DEBUG_DUMP="$O/r2aa_one" timeout 300 python scripts/config3_slab_proxy.py 2>&1 | grep "hipx sor\] strand KIND 0 done" | head -1 | cut -c1-200
