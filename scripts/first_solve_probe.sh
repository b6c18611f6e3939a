#!/bin/bash
# where the first KSPSolve of the drop-in spends the time the second one does not: -log_view of the reference's executable + plugin (stock KSPCG, 400 iterations,
# then the same solve again: -resolve), events with their max time
R=${GRAFT_REPO_ROOT:-/root/repo}
A="-stencil 7 -n 256 -pc_type jacobi -ksp_rtol 1e-50 -ksp_norm_type preconditioned -dll_prepend $R/petsc_amd/lib/libpetschipx.so -vec_type hipx -mat_type aijhipx -ksp_type cg -ksp_max_it 400 -resolve"
export HIPX_NO_TORCH=1 MKL_NUM_THREADS=1 OMP_NUM_THREADS=1
$R/oracle/_ref/bin/ref_driver $A $@ -log_view 2>&1 | grep -E "iterations|second_solve|^(Vec|Mat|KSP|PC)[A-Za-z]+ +[0-9]" | cut -c1-120
