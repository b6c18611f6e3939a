#!/bin/bash
# round-4 run C: march2 after the register work (27-point class), reduction grid 256 vs 512, small-tile / higher-occupancy variants
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r04c
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_mat.py tests/test_gpu_vec.py tests/test_gpu_exact.py -m gpu -q --timeout 600 -p no:cacheprovider -rf -x > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
grep -E "passed|failed|Error|assert" gpurun_out/${T}_pytest.log | tail -8
q() { python bench.py --quick "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%8.1f it/s  %.4f ms/it  spmv %.4f ms  %s' % (d['value'] or -1, d['ms_per_step'], r['avg_launch_ms'], r['kernel'][:40]))"; }
echo "7pt 256 march2:";            q
echo "7pt 256 march2 RED 512:";    HIPX_RED_BLOCKS=512 q
echo "7pt 256 march2 (again):";    q
echo "7pt 256 march2 RED 512:";    HIPX_RED_BLOCKS=512 q
echo "7pt 256 march2 L=1024:";     HIPX_TMPL_MARCH_L=1024 q
echo "7pt 256 march2 L=1024 units 1024:";     HIPX_TMPL_MARCH_L=1024 HIPX_TMPL_MARCH_UNITS=1024 q
echo "7pt 256 march2 units 256:";  HIPX_TMPL_MARCH_UNITS=256 q
echo "27pt 256 march2:";           q --stencil 27 --grid 256 --steps 100
echo "27pt 256 march1:";           HIPX_MARCH1=1 q --stencil 27 --grid 256 --steps 100
echo "27pt 256 march2 L=1024:";    HIPX_TMPL_MARCH_L=1024 q --stencil 27 --grid 256 --steps 100
echo "total ${SECONDS}s"
