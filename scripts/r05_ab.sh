#!/bin/bash
# same-box A/B pairs of round 5 that the library still has switches for: bash scripts/r05_ab.sh [repeats]
# (run on the GPU box: gpurun -- 'bash scripts/gpu_run.sh <tag> "sh:bash scripts/r05_ab.sh 2"'; results of the round: profiles/r05_experiments.txt)
q() { python bench.py --quick "$@" > /dev/null 2>&1; python -c "
import json,sys
d=json.load(open('bench_detail.json'))
r=d.get('roofline',{})
print('  it/s %.1f ms %.4f %s kernels %s setup %s' % (d['value'] or -1, d['ms_per_step'], r.get('kernel','')[:30], [(k['match'], round(k['avg_launch_us'],1)) for k in r.get('by_kernel',[])], {k: round(v, 3) for k, v in (d.get('setup_split') or {}).items()}))"; }
for i in $(seq 1 ${1:-2}); do
echo "fold of the dot partials in the product kernel (default)"; q --steps 400 --warmup 40
echo "HIPX_MARCH_NOFOLD=1: the separate fold kernel"; HIPX_MARCH_NOFOLD=1 q --steps 400 --warmup 40
done
echo "7-pt 200^3: march2 on the whole tiles + remainder kernel"; q --grid 200 --steps 400 --warmup 40
echo "7-pt 200^3: HIPX_MARCH2_NOREM=1 (first march kernel)"; HIPX_MARCH2_NOREM=1 q --grid 200 --steps 400 --warmup 40
echo "27-pt 200^3"; q --grid 200 --stencil 27 --steps 200 --warmup 20
echo "27-pt 200^3: HIPX_MARCH2_NOREM=1"; HIPX_MARCH2_NOREM=1 q --grid 200 --stencil 27 --steps 200 --warmup 20
