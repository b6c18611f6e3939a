#!/bin/bash
# round-4 run K: 512-thread march form for 1024-point lines (config 5's share), retests of run J's failures
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r04k
SECONDS=0
timeout 1500 python -m pytest tests/test_gpu_mat.py \
  "tests/test_gpu_scale_parity.py::test_single_reduction_cg_follows_the_reference_in_exact_mode" \
  "tests/test_gpu_plugin.py::test_matmulttranspose_on_the_device_bit_exact" "tests/test_gpu_plugin_mpi.py::test_matmulttranspose_mpiaijhipx_bit_exact_vs_cpu_mpi" \
  -m gpu -q --timeout 900 -p no:cacheprovider -rf > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
grep -E "passed|failed" gpurun_out/${T}_pytest.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/${T}_pytest.log | head -20
q() { python bench.py --quick "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%8.1f it/s  %.4f ms/it  spmv %.4f ms  %s' % (d['value'] or -1, d['ms_per_step'], r['avg_launch_ms'], r['kernel'][:24]))"; }
C5="--grid 1024 --scaling weak --pc none --steps 50 --warmup 5"
echo "config5 share, march2 512 threads + fused prologue:"; q $C5
echo "config5 share, no CG fusion:"; HIPX_NO_CGFUSE=1 q $C5
echo "config5 share, pair form (round 3):"; HIPX_MARCH1=1 q $C5
echo "config5 share again:"; q $C5
echo "total ${SECONDS}s"
