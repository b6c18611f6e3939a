#!/bin/bash
# the reference's executable + plugin on the headline configuration: KSPSolve time at two iteration counts (slope = time per iteration, the rest is set-up
# inside KSPSolve), lazy-fusion counts; arg "stats": rocprofv3 kernel stats of the stock-CG run
R=${GRAFT_REPO_ROOT:-/root/repo}
A="-stencil 7 -n 256 -pc_type jacobi -ksp_rtol 1e-50 -ksp_norm_type preconditioned -dll_prepend $R/petsc_amd/lib/libpetschipx.so -vec_type hipx -mat_type aijhipx"
export HIPX_NO_TORCH=1 MKL_NUM_THREADS=1 OMP_NUM_THREADS=1
for extra in "-ksp_type cg -hipx_lazy_view" "-ksp_type cg -hipx_lazy_fusion 0" "-ksp_type cg -hipx_lazy_fusion 0 -hipx_reduction_cache 0" "-ksp_type cghipx"; do
  for its in 400 1200; do
    echo "== $extra -ksp_max_it $its"
    $R/oracle/_ref/bin/ref_driver $A $extra -ksp_max_it $its 2>&1 | grep -E "iterations|lazy" | cut -c1-300
  done
done
if [ "$1" = "stats" ]; then
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pcg -o s -- $R/oracle/_ref/bin/ref_driver $A -ksp_type cg -ksp_max_it 400 > /dev/null 2>&1
  f=$(find /tmp/pcg -name "*kernel_stats.csv" | head -1)
  python - "$f" <<PY
import csv,sys
for r in list(csv.reader(open(sys.argv[1])))[1:12]:
    print(r[0].replace('(anonymous namespace)::','').replace('void ','')[:70], r[1], round(float(r[3])/1e3,1),'us')
PY
fi
