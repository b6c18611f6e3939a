#!/bin/bash
# round-4 run D: the CG direction update as the march2 kernel's prologue (two kernels per iteration): parity + same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r04d
SECONDS=0
timeout 1200 python -m pytest tests/test_gpu_mat.py tests/test_gpu_ksp.py tests/test_gpu_exact.py tests/test_gpu_fullsize.py tests/test_gpu_plugin.py \
  "tests/test_gpu_scale_parity.py::test_config2_cg_jacobi_256_history_vs_reference" \
  "tests/test_gpu_scale_parity.py::test_config2_exact_mode_history_equals_the_reference_with_exact_blas_bit_for_bit" \
  "tests/test_gpu_scale_parity.py::test_large_configurations_follow_the_committed_exact_histories" \
  -m gpu -q --timeout 900 -p no:cacheprovider -rf > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
grep -E "passed|failed|Error|assert" gpurun_out/${T}_pytest.log | tail -12
q() { python bench.py --quick "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%8.1f it/s  %.4f ms/it  spmv %.4f ms  %s' % (d['value'] or -1, d['ms_per_step'], r['avg_launch_ms'], r['kernel'][:40]))"; }
echo "7pt 256 fused AB:";            q
echo "7pt 256 no CG fusion:";        HIPX_NO_CGFUSE=1 q
echo "7pt 256 fused AB (again):";    q
echo "7pt 256 march1:";              HIPX_MARCH1=1 q
echo "7pt 256 fused AB L=1024:";     HIPX_TMPL_MARCH_L=1024 q
echo "7pt 256 no fusion L=1024:";    HIPX_TMPL_MARCH_L=1024 HIPX_NO_CGFUSE=1 q
echo "7pt 512 fused AB:";            q --grid 512 --steps 50
echo "7pt 512 no fusion:";           HIPX_NO_CGFUSE=1 q --grid 512 --steps 50
echo "7pt 256 exact mode fused:";    HIPX_REDUCTIONS=exact q
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${T}_prof -o q -- python $GRAFT_REPO_ROOT/bench.py --quick > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print(r['Name'][:80].ljust(80), r['Calls'], '%.1f us'%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
echo "total ${SECONDS}s"
