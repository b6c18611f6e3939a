#!/bin/bash
# template SpMV (pair form) against the number of persistent workgroups (HIPX_TMPL_BLOCKS): latency-bound (time ~ 1 / workgroups) or throughput-bound (flat)?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for b in 512 768 1024 1280 1536 2048; do
  HIPX_TMPL_BLOCKS=$b timeout 600 python bench.py --quick --stencil 7 --grid 256 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blocks $b: spmv %.4f ms  %.1f it/s' % (d['roofline']['avg_launch_ms'], d['value']))"
done
