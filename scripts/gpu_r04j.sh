#!/bin/bash
# round-4 run J: single-reduction CG, device MatMultTranspose, the relaxed fast-mode / exact-mode cross-path tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r04j
SECONDS=0
timeout 1500 python -m pytest tests/test_gpu_bench_multi.py tests/test_gpu_plugin.py tests/test_gpu_plugin_mpi.py tests/test_gpu_plugin_int64.py tests/test_gpu_ksp.py \
  "tests/test_gpu_scale_parity.py::test_single_reduction_cg_bit_identical_to_the_reference_in_exact_mode" \
  "tests/test_gpu_scale_parity.py::test_config2_cg_jacobi_256_history_vs_reference" \
  -m gpu -q --timeout 900 -p no:cacheprovider -rf > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
grep -E "passed|failed" gpurun_out/${T}_pytest.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/${T}_pytest.log | head -20
q() { python bench.py --quick --steps 400 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%8.1f it/s  %.4f ms/it  spmv %.4f ms  %s' % (d['value'] or -1, d['ms_per_step'], r['avg_launch_ms'], r['kernel'][:24]))"; }
echo "standard fused CG:"; q
echo "single-reduction CG (--pipeline 3):"; q --pipeline 3
echo "total ${SECONDS}s"
