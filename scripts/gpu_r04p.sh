#!/bin/bash
# round-4 run P: cghipx self-driven (launch-ahead under the plugin) -- plugin test tiers + its/s through the drop-in; SELL run-code test fix
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r04p
SECONDS=0
timeout 1500 python -m pytest tests/test_gpu_plugin.py tests/test_gpu_plugin_mpi.py tests/test_gpu_plugin_int64.py tests/test_gpu_plugin_kats.py tests/test_gpu_vs_reference.py \
  "tests/test_gpu_scale_parity.py::test_config2_cg_jacobi_256_history_vs_reference" "tests/test_gpu_scale_parity.py::test_config2_exact_mode_history_equals_the_reference_with_exact_blas_bit_for_bit" \
  "tests/test_gpu_scale_parity.py::test_single_reduction_cg_follows_the_reference_in_exact_mode" \
  "tests/test_gpu_mat.py::test_sell_triple_run_column_codes_for_three_unknowns_per_node" \
  -m gpu -q --timeout 900 -p no:cacheprovider -rf > gpurun_out/${T}_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/${T}_pytest.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/${T}_pytest.log | head
D=oracle/_ref/bin/ref_driver; P="-dll_prepend petsc_amd/lib/libpetschipx.so -vec_type hipx -mat_type aijhipx"
A="-stencil 7 -n 256 -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 400 -ksp_norm_type preconditioned"
for rep in 1 2; do
echo "cghipx self-driven:"; HIPX_NO_TORCH=1 $D $A -ksp_type cghipx $P | grep iterations
echo "cghipx stepwise:";    HIPX_NO_TORCH=1 HIPX_CGHIPX_STEPWISE=1 $D $A -ksp_type cghipx $P | grep iterations
echo "stock cg:";           HIPX_NO_TORCH=1 $D $A -ksp_type cg $P | grep iterations
done
echo "total ${SECONDS}s"
