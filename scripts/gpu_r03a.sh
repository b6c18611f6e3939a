#!/bin/bash
# round-3 run A: validates the round's infrastructure on the MI355X -- the GPU suite (incl. the multi-rank bench, the PetscSF type,
# the exact-BLAS yardstick tests), the default bench line with its new legs, the SOR work-in-progress switches, the template-kernel
# timing probes.  Usage: bash scripts/gpu_r03a.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03a
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rf -x --deselect tests/test_gpu_sor.py > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
S0=$SECONDS
timeout 900 python bench.py > gpurun_out/${T}_bench.log 2>gpurun_out/${T}_bench.err
echo "default bench: $((SECONDS - S0)) s" >> gpurun_out/${T}_bench.err
# SOR: the line stores of t that went in unmeasured (default), and the two work-in-progress switches
for v in "" "HIPX_SOR_TFLUSH=1" "HIPX_SOR_LOCKSTEP_BWD=1" "HIPX_SOR_TFLUSH=1 HIPX_SOR_LOCKSTEP_BWD=1"; do
  echo "=== slab proxy: ${v:-default}" >> gpurun_out/${T}_slab.log
  env $v timeout 300 python scripts/config3_slab_proxy.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> gpurun_out/${T}_slab.log
done
# template SpMV: timing probes (wrong results by construction; time only)
for p in 0 1 2 3; do
  echo "=== HIPX_TMPL_PROBE=$p" >> gpurun_out/${T}_tmpl_probe.log
  HIPX_TMPL_PROBE=$p timeout 200 python scripts/spmv_variants.py 256 7 26 2>&1 | grep -v amdgpu.ids | head -3 >> gpurun_out/${T}_tmpl_probe.log
done
tail -5 gpurun_out/${T}_pytest.log
tail -2 gpurun_out/${T}_bench.err
tail -1 gpurun_out/${T}_bench.log | cut -c1-600
cat gpurun_out/${T}_slab.log | grep -i "===\|sor\|symmetric" | cut -c1-160
cat gpurun_out/${T}_tmpl_probe.log
echo "total ${SECONDS}s"
