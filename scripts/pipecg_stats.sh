#!/bin/bash
# rocprofv3 kernel statistics of the drop-in's pipelined CG runs (reference executable + plugin, 7-pt 256^3, 400 iterations): -> gpurun_out/<tag>/pipecg_<ksp>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r06}
O=$R/gpurun_out/$T
mkdir -p $O
export HIPX_NO_TORCH=1 MKL_NUM_THREADS=1 OMP_NUM_THREADS=1 TMPDIR=/tmp
shift
for K in ${@:-pipecghipx pipecg groppcg groppcghipx}; do
  A="-stencil 7 -n 256 -pc_type jacobi -ksp_rtol 1e-50 -ksp_norm_type preconditioned -dll_prepend $R/petsc_amd/lib/libpetschipx.so -vec_type hipx -mat_type aijhipx -ksp_type $K -ksp_max_it 400"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$K -o s -- $R/oracle/_ref/bin/ref_driver $A > $O/stats_$K.out 2>&1)
  f=$(find $O/prof_$K -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/pipecg_${K}_kernel_stats.csv && head -8 "$f" | cut -c1-200
  rm -rf $O/prof_$K
done
