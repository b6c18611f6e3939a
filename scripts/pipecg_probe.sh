#!/bin/bash
# iterations/s of the CG family through the drop-in (reference executable + plugin) on BASELINE config 2's system: stock loops over the hipx types and the hipx KSP types,
# first and second solve; plus the host layer's loops through bench.py's own legs.  Usage: bash scripts/pipecg_probe.sh [extra options]
R=${GRAFT_REPO_ROOT:-/root/repo}
export HIPX_NO_TORCH=1 MKL_NUM_THREADS=1 OMP_NUM_THREADS=1
for K in cg cghipx pipecg pipecghipx groppcg groppcghipx pipecr; do
  A="-stencil 7 -n 256 -pc_type jacobi -ksp_rtol 1e-50 -ksp_norm_type preconditioned -dll_prepend $R/petsc_amd/lib/libpetschipx.so -vec_type hipx -mat_type aijhipx -ksp_type $K -ksp_max_it 400 -resolve"
  echo "== $K $@"
  $R/oracle/_ref/bin/ref_driver $A $@ 2>&1 | grep -E "iterations|second_solve" | cut -c1-160
done
