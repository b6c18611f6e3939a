#!/bin/bash
# round-3 run R: fused Chebyshev (plugin + host layer), MatLoad test (inode), pair-form tests; Chebyshev timing rows.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03r
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_plugin.py tests/test_gpu_mat.py -m gpu -q --timeout 600 -p no:cacheprovider -k "chebyshev or matload or pair or auto_variant or templates" -rf > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
tail -12 gpurun_out/${T}_pytest.log | cut -c1-300
for ksp in chebyshev chebyshevhipx; do
  HIPX_NO_TORCH=1 oracle/_ref/bin/ref_driver -stencil 7 -n 256 -ksp_type $ksp -pc_type jacobi -ksp_norm_type none -ksp_max_it 400 -ksp_chebyshev_eigenvalues 0.1,2.0 -dll_prepend petsc_amd/lib/libpetschipx.so -vec_type hipx -mat_type aijhipx 2>&1 | tail -1
done
echo "total ${SECONDS}s"
