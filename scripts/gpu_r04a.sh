#!/bin/bash
# round-4 run A: the exact reduction mode (new tests), PCNONE through the fused loop, the touched multi-rank paths; quick bench + per-kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r04a
SECONDS=0
timeout 1200 python -m pytest tests/test_gpu_exact.py tests/test_gpu_halo.py tests/test_gpu_vec.py tests/test_gpu_ksp.py tests/test_gpu_bench_multi.py tests/test_gpu_multirank.py tests/test_gpu_sf.py \
  "tests/test_gpu_scale_parity.py::test_config2_exact_mode_history_equals_the_reference_with_exact_blas_bit_for_bit" \
  "tests/test_gpu_scale_parity.py::test_config3_solver_gmres30_sor_exact_mode_within_1e12" \
  "tests/test_gpu_scale_parity.py::test_pipelined_and_single_reduction_cg_exact_mode_within_1e12" \
  "tests/test_gpu_scale_parity.py::test_large_configurations_follow_the_committed_exact_histories" \
  -m gpu -q --timeout 900 -p no:cacheprovider -rf > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
grep -E "passed|failed|Error|assert" gpurun_out/${T}_pytest.log | tail -12
S0=$SECONDS
echo "total ${SECONDS}s"
