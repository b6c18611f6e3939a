#!/bin/bash
# round-3 run D: the second form of the template kernel (static round robin + drift throttle, late-issued prefetches, optional prefetch
# wave) against the first, stand-alone and inside CG; the pattern-template kernel (variant 29) against the packed CSR kernels.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03d
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_mat.py tests/test_gpu_ksp.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -p no:cacheprovider -rf -x > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
for v in "HIPX_TMPL_V2=0" "HIPX_TMPL_V2=1 HIPX_TMPL_LAG=0" "HIPX_TMPL_V2=1 HIPX_TMPL_LAG=1" "HIPX_TMPL_V2=1 HIPX_TMPL_LAG=2" "HIPX_TMPL_V2=1 HIPX_TMPL_LAG=1 HIPX_TMPL_PFW=1" "HIPX_TMPL_V2=1 HIPX_TMPL_LAG=2 HIPX_TMPL_PFW=1" "HIPX_TMPL_V2=1 HIPX_TMPL_LAG=0 HIPX_TMPL_PFW=1"; do
  echo "=== $v" >> gpurun_out/${T}_tmpl.log
  env $v timeout 200 python scripts/spmv_variants.py 256 7 26 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/${T}_tmpl.log
  env $v timeout 300 python bench.py --quick --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  bench --quick: %.1f it/s  %.4f ms/it  spmv %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))" >> gpurun_out/${T}_tmpl.log
done
for v in "HIPX_TMPL_V2=0" "HIPX_TMPL_V2=1 HIPX_TMPL_LAG=1" "HIPX_TMPL_V2=1 HIPX_TMPL_LAG=1 HIPX_TMPL_PFW=1"; do
  echo "=== 27-pt 160^3 $v" >> gpurun_out/${T}_tmpl.log
  env $v timeout 200 python scripts/spmv_variants.py 160 27 26 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/${T}_tmpl.log
done
echo "=== pattern templates (29) vs packed CSR (23 / 22)" >> gpurun_out/${T}_tmpl.log
timeout 200 python scripts/spmv_variants.py 256 7 23,29 2>&1 | grep -v amdgpu.ids | head -2 >> gpurun_out/${T}_tmpl.log
timeout 200 python scripts/spmv_variants.py 160 27 22,29 2>&1 | grep -v amdgpu.ids | head -2 >> gpurun_out/${T}_tmpl.log
tail -4 gpurun_out/${T}_pytest.log | cut -c1-300
cat gpurun_out/${T}_tmpl.log | cut -c1-200
echo "total ${SECONDS}s"
