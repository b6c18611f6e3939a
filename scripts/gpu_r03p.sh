#!/bin/bash
# round-3 run P: does the template kernel's time per row depend on the grid size being a power of two (8 XCD slabs 2^24 bytes apart)?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
SECONDS=0
for n in 240 250 256 264 272 288; do
  for mode in "HIPX_TMPL_NOPAIR=1" "HIPX_TMPL_PERSIST=1"; do
    env $mode timeout 600 python bench.py --quick --stencil 7 --grid $n 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); rows=$n**3; ms=d['roofline']['avg_launch_ms']
    print('7-pt %d^3 [$mode]: spmv %.4f ms = %.3f ns per 1000 rows, %.1f it/s  %s' % ($n, ms, ms*1e6/rows*1000, d['value'], d['roofline']['kernel'][:18]))
except Exception as e: print('$n $mode failed', e)"
  done
done
echo "total ${SECONDS}s"
