#!/bin/bash
# round-3 run C: the tests touched since run B (PetscSF route of MATMPIAIJHIPX, launch-ahead CG on several ranks, MatAXPY, pbjacobi),
# the template kernel with the cached short template (SHORT) against the previous form, inside CG and stand-alone.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03c
SECONDS=0
timeout 1200 python -m pytest tests/test_gpu_sf.py tests/test_gpu_bench_multi.py tests/test_gpu_multirank.py tests/test_gpu_plugin.py tests/test_gpu_plugin_mpi.py tests/test_gpu_ksp.py tests/test_gpu_mat.py tests/test_gpu_halo.py -m gpu -q --timeout 900 -p no:cacheprovider -rf > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
for v in "" "HIPX_TMPL_NOSHORT=1"; do
  echo "=== template kernel ${v:-SHORT (default)}" >> gpurun_out/${T}_tmpl.log
  env $v timeout 200 python scripts/spmv_variants.py 256 7 26 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/${T}_tmpl.log
  env $v timeout 300 python bench.py --quick --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench --quick: %.1f it/s  %.4f ms/it  spmv %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))" >> gpurun_out/${T}_tmpl.log
done
echo "=== 2 ranks (one GPU, IPC): launch-ahead vs host-synchronised" >> gpurun_out/${T}_tmpl.log
for p in 1 2; do
  timeout 300 python bench.py --gpus 2 --grid 128 --steps 100 --warmup 10 --quick --pipeline $p 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline $p: %.1f it/s  parity %s' % (d['value'], d['parity_gate'].get('max_rel_diff')), [ (r['spmv_ms'], r.get('halo_ms'), r.get('allreduce_ms')) for r in d['per_rank']])" >> gpurun_out/${T}_tmpl.log
done
tail -6 gpurun_out/${T}_pytest.log | cut -c1-300
cat gpurun_out/${T}_tmpl.log
echo "total ${SECONDS}s"
