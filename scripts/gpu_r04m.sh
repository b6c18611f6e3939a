#!/bin/bash
# round-4 run M: non-temporal loads / stores around the update kernel inside the CG loop
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
q() { python bench.py --quick --steps 400 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%8.1f it/s  %.4f ms/it  product %.4f ms' % (d['value'] or -1, d['ms_per_step'], r['avg_launch_ms']))"; }
for rep in 1 2; do
echo "default:";  q
echo "NT load w:";  HIPX_CG_FUSED_NT=2 q
echo "NT load w, r:";  HIPX_CG_FUSED_NT=3 q
echo "NT store y:";  HIPX_MARCH_NT_STORE=1 q
echo "NT store y + NT load w:";  HIPX_MARCH_NT_STORE=1 HIPX_CG_FUSED_NT=2 q
done
