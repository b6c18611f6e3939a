#!/bin/bash
# How the strand SOR kernel was profiled from the inside (round 2; run on the GPU box: gpurun -- 'bash scripts/sor_trace.sh 16').
#   HIPX_SOR_DEBUG=1            per-launch statistics on stderr: iterations, lane-iterations waiting (operands / dependencies),
#                               fallback loads, loader passes, shader clocks per phase, F-wave statistics of the split kernel
#   HIPX_SOR_DEBUG_DUMP=prefix  <prefix>_kind<K>.txt: per panel start / first row of lane 0 / first row of lane 63 / end (us)
#   HIPX_SOR_TRACE_PANEL=p      <prefix>_trace<K>.bin for panel p: completion time of every row (HIPX_SOR_TRACE_ROWS=0 turns
#                               that part off: it costs a scattered store per row), the loader's passes as seen by lane
#                               HIPX_SOR_TRACE_LANE (progress, staging front, rows asked for), a per-iteration log of that lane of
#                               the compute wave from iteration HIPX_SOR_TRACE_IT0 on (phase clocks, which entries are late)
#   HIPX_SOR_MODE=strand|dep|levels, HIPX_SOR_SPLIT=0|1|2, HIPX_SOR_WG_PER_CU=1|2, HIPX_SOR_POLL=sys
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O="$GRAFT_REPO_ROOT/gpurun_out"; mkdir -p "$O"
P=${1:-16}
HIPX_SOR_DEBUG=1 HIPX_SOR_TRACE_PANEL=$P HIPX_SOR_TRACE_ROWS=${2:-1} HIPX_SOR_TRACE_LANE=${3:-0} HIPX_SOR_TRACE_IT0=${4:-400} HIPX_SOR_DEBUG_DUMP="$O/sortrace_p$P" \
  timeout 300 python scripts/config3_slab_proxy.py 2>&1 | grep "hipx sor\]" | cut -c1-400
ls -la "$O"/sortrace_p${P}_*
