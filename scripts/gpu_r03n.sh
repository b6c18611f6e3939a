#!/bin/bash
# round-3 run N: pair form, non-persistent (one workgroup per chunk, no tickets / barriers / prefetch loads in the vmcnt queue) vs persistent.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03n
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_mat.py tests/test_gpu_ksp.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
tail -8 gpurun_out/${T}_pytest.log | cut -c1-300
run() {  # label, env assignments..., bench args
  local label=$1; shift
  env "$@" timeout 600 python bench.py --quick $ARGS 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label: %.1f it/s  ms/step %.4f  spmv %.4f ms  %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'][:22]))
except Exception as e: print('$label: failed', e)"
}
for st in 7 27; do
  ARGS="--stencil $st --grid 256"
  run "stencil $st non-persistent" A=1
  run "stencil $st persistent    " HIPX_TMPL_PERSIST=1
  run "stencil $st no pair form  " HIPX_TMPL_NOPAIR=1
done
ARGS="--stencil 7 --grid 512"
run "7-pt 512^3 non-persistent" A=1
run "7-pt 512^3 no pair form  " HIPX_TMPL_NOPAIR=1
bash scripts/pmc_sq.sh ${T}_np 0 > gpurun_out/${T}_sq_np.txt 2>&1
sed -n 1,40p gpurun_out/${T}_sq_np.txt | cut -c1-100
echo "total ${SECONDS}s"
