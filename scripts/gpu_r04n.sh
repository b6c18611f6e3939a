#!/bin/bash
# round-4 run N: non-temporal x loads / stores in the fused product kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
q() { python bench.py --quick --steps 400 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%8.1f it/s  %.4f ms/it  product %.4f ms' % (d['value'] or -1, d['ms_per_step'], r['avg_launch_ms']))"; }
for rep in 1 2; do
echo "default (NT load w in C):";  q
echo "+ NT load x:";  HIPX_MARCH_NT_X=1 q
echo "+ NT store x:";  HIPX_MARCH_NT_X=2 q
echo "+ both:";  HIPX_MARCH_NT_X=3 q
done
