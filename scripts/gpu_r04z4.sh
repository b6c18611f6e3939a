#!/bin/bash
# round-4 run Z4: strand SOR with one workgroup per CU instead of two (fewer pollers), config 3's solver
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
q() { python bench.py --quick "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f it/s  %.4f ms/it  sor %s' % (d['value'] or -1, d['ms_per_step'], [r.get('sor_ms') for r in d.get('per_rank', [])] or d.get('roofline_sor', {}).get('avg_call_ms')))"; }
for rep in 1 2; do
echo "27pt 256 gmres+sor, 2 WG/CU:"; q --ksp gmres --pc sor --stencil 27 --grid 256 --steps 60 --warmup 5
echo "27pt 256 gmres+sor, 1 WG/CU:"; HIPX_SOR_WG_PER_CU=1 q --ksp gmres --pc sor --stencil 27 --grid 256 --steps 60 --warmup 5
done
echo "7pt 256 gmres+sor, 2 WG/CU:"; q --ksp gmres --pc sor --stencil 7 --grid 256 --steps 60 --warmup 5
echo "7pt 256 gmres+sor, 1 WG/CU:"; HIPX_SOR_WG_PER_CU=1 q --ksp gmres --pc sor --stencil 7 --grid 256 --steps 60 --warmup 5
