#!/bin/bash
# round-4 run L: 2048-row tiles by 512 threads (4 waves per SIMD) against 256 threads (2 waves per SIMD), 7-pt 256^3 and 512^3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=r04l
q() { python bench.py --quick --steps 400 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%8.1f it/s  %.4f ms/it  spmv %.4f ms  %s' % (d['value'] or -1, d['ms_per_step'], r['avg_launch_ms'], r['kernel'][:24]))"; }
for rep in 1 2; do
echo "256 threads, fused:";  q
echo "512 threads, fused:";  HIPX_MARCH_NT512=1 q
echo "256 threads, SpMV alone:";  HIPX_NO_CGFUSE=1 q
echo "512 threads, SpMV alone:";  HIPX_MARCH_NT512=1 HIPX_NO_CGFUSE=1 q
done
HIPX_MARCH_NT512=1 timeout 600 python -m pytest tests/test_gpu_mat.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "march2 or prologue or fullsize" 2>&1 | tail -2
