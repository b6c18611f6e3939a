#!/bin/bash
# round-4 run B: the second-generation march kernel (spmv_march2_kernel): bit-exactness, same-box A/B against the first one, the wider fused update loop
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r04b
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_mat.py tests/test_gpu_ksp.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -p no:cacheprovider -rf -x > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
grep -E "passed|failed|Error|assert" gpurun_out/${T}_pytest.log | tail -8
q() { python bench.py --quick "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%8.1f it/s  %.4f ms/it  spmv %.4f ms  %s' % (d['value'] or -1, d['ms_per_step'], r['avg_launch_ms'], r['kernel'][:40]))"; }
echo "7pt 256 march2:";            q
echo "7pt 256 march2 (again):";    q
echo "7pt 256 march1:";            HIPX_MARCH1=1 q
echo "7pt 256 march2, fused U2:";  HIPX_CG_FUSED_U2=1 q
echo "27pt 256 march2:";           q --stencil 27 --grid 256 --steps 100
echo "27pt 256 march1:";           HIPX_MARCH1=1 q --stencil 27 --grid 256 --steps 100
echo "7pt 512 march2:";            q --grid 512 --steps 50
echo "7pt 512 march1:";            HIPX_MARCH1=1 q --grid 512 --steps 50
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${T}_prof -o q -- python $GRAFT_REPO_ROOT/bench.py --quick > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200
echo "total ${SECONDS}s"
