#!/bin/bash
# round-4 run F: interleaved add chains in march2; sweep form of the update kernel; alternating repetitions against box noise
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=r04f
q() { python bench.py --quick --steps 400 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%8.1f it/s  %.4f ms/it  spmv %.4f ms  %s' % (d['value'] or -1, d['ms_per_step'], r['avg_launch_ms'], r['kernel'][:24]))"; }
for rep in 1 2 3; do
echo "fused AB + sweep C:";       q
echo "no CG fusion:";             HIPX_NO_CGFUSE=1 q
echo "fused AB, chunk C:";        HIPX_CG_FUSED_CHUNK=1 q
done
echo "march1, chunk C, no fusion (round 3):"; HIPX_MARCH1=1 HIPX_CG_FUSED_CHUNK=1 HIPX_CG_FUSED_U2=1 q
echo "27pt 256 march2:";           q --stencil 27 --grid 256 --steps 100
echo "27pt 256 march1:";           HIPX_MARCH1=1 q --stencil 27 --grid 256 --steps 100
timeout 600 python -m pytest tests/test_gpu_mat.py tests/test_gpu_ksp.py -m gpu -q --timeout 600 -p no:cacheprovider -x 2>&1 | tail -2
