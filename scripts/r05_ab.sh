#!/bin/bash
# same-box A/B pairs of round 5: bash scripts/r05_ab.sh [repeats]
q() { python bench.py --quick "$@" > /dev/null 2>&1; python -c "
import json,sys
d=json.load(open('bench_detail.json'))
r=d.get('roofline',{})
print('  it/s %.1f ms %.4f %s kernels %s setup %s' % (d['value'] or -1, d['ms_per_step'], r.get('kernel','')[:30], [(k['match'], round(k['avg_launch_us'],1)) for k in r.get('by_kernel',[])], {k: round(v, 3) for k, v in (d.get('setup_split') or {}).items()}))"; }
for i in $(seq 1 ${1:-1}); do
for g in 8192 4096 2048 1024 512; do
echo "one-shot update kernel, two-level tickets, G = $g"; HIPX_CG_FUSED_OS_G=$g q --steps 400 --warmup 40
done
echo "HIPX_CG_FUSED_OS=0 (persistent reduction grid)"; HIPX_CG_FUSED_OS=0 q --steps 400 --warmup 40
done
