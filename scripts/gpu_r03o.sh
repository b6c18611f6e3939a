#!/bin/bash
# round-3 run O: persistent pair form with a prefetch wave (no HBM miss in a compute wave's vmcnt queue).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03o
SECONDS=0
HIPX_TMPL_PERSIST=1 timeout 900 python -m pytest tests/test_gpu_mat.py -m gpu -q --timeout 600 -p no:cacheprovider -k "pair or stencil_spmv or templates" > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
tail -5 gpurun_out/${T}_pytest.log | cut -c1-300
run() {
  local label=$1; shift
  env "$@" timeout 600 python bench.py --quick $ARGS 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label: %.1f it/s  ms/step %.4f  spmv %.4f ms  %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'][:22]))
except Exception as e: print('$label: failed', e)"
}
for st in 7 27; do
  ARGS="--stencil $st --grid 256"
  run "stencil $st persistent + prefetch wave" HIPX_TMPL_PERSIST=1
  run "stencil $st persistent, NOPF          " HIPX_TMPL_PERSIST=1 HIPX_TMPL_NOPF=1
  run "stencil $st no pair form              " HIPX_TMPL_NOPAIR=1
done
ARGS="--stencil 7 --grid 512"
run "7-pt 512^3 persistent + prefetch wave" HIPX_TMPL_PERSIST=1
run "7-pt 512^3 no pair form              " HIPX_TMPL_NOPAIR=1
echo "total ${SECONDS}s"
