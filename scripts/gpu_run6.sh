#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
R="$GRAFT_REPO_ROOT"
timeout 1800 python -m pytest tests/test_gpu_plugin.py -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/pytest7.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest7.log
B=$R/oracle/_ref/bin
P="-dll_prepend $R/petsc_amd/lib/libpetschipx.so -vec_type hipx -mat_type aijhipx"
# end-to-end drop-in at the BASELINE size through the reference's own KSPSolve_CG
MKL_NUM_THREADS=1 OMP_NUM_THREADS=1 timeout 600 $B/ref_driver -stencil 7 -n 256 -ksp_type cg -pc_type jacobi -ksp_max_it 200 -ksp_rtol 1e-50 -matmult_its 50 $P > gpurun_out/plugin7_256.log 2>&1
timeout 600 $B/ref_driver -stencil 7 -n 256 -ksp_type cg -pc_type jacobihipx -ksp_max_it 200 -ksp_rtol 1e-50 $P -log_view > gpurun_out/plugin7_256_logview.log 2>&1
# CPU reference leg (1 core), bounded
MKL_NUM_THREADS=1 OMP_NUM_THREADS=1 timeout 600 $B/ref_driver -stencil 7 -n 256 -ksp_type cg -pc_type jacobi -ksp_max_it 20 -ksp_rtol 1e-50 -matmult_its 5 > gpurun_out/cpu7_ref_256.log 2>&1
tail -6 gpurun_out/pytest7.log; cat gpurun_out/plugin7_256.log | tail -3; grep -E "^(MatMult|VecTDot|VecNorm|VecAXPY|VecAYPX|VecPointwiseMult|KSPSolve|PCApply) " gpurun_out/plugin7_256_logview.log | cut -c1-100; tail -3 gpurun_out/cpu7_ref_256.log
