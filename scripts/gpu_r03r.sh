#!/bin/bash
# fused Chebyshev (plugin + host layer): bit-identical solutions, timing rows (device drained inside the timed region).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03r
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_plugin.py tests/test_gpu_mat.py -m gpu -q --timeout 600 -p no:cacheprovider -k "chebyshev or pair" -rf > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s: $(tail -1 gpurun_out/${T}_pytest.log)"
for ksp in chebyshev chebyshevhipx; do
  for st in 7 27; do
    echo "$ksp $st-pt 256^3: $(HIPX_NO_TORCH=1 oracle/_ref/bin/ref_driver -stencil $st -n 256 -ksp_type $ksp -pc_type jacobi -ksp_norm_type none -ksp_max_it 400 -ksp_chebyshev_eigenvalues 0.1,2.0 -dll_prepend petsc_amd/lib/libpetschipx.so -vec_type hipx -mat_type aijhipx 2>&1 | tail -1)"
  done
done
echo "total ${SECONDS}s"
