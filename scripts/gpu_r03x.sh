#!/bin/bash
# round-3 run X: pair form with a control wave (prefetch + tickets in a fifth wave): tests, timing against the four-wave kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03x
SECONDS=0
for tg in 1 2 4; do
HIPX_TMPL_TG=$tg timeout 600 python -m pytest tests/test_gpu_mat.py -m gpu -q --timeout 300 -p no:cacheprovider -k "pair or stencil_spmv or templates or auto_variant" > gpurun_out/${T}_pytest_tg$tg.log 2>&1
echo "tg $tg pytest exit $? after ${SECONDS}s: $(tail -1 gpurun_out/${T}_pytest_tg$tg.log)"
done
run() {
  local label=$1; shift
  env "$@" timeout 600 python bench.py --quick $ARGS 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label: %.1f it/s  ms/step %.4f  spmv %.4f ms  %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'][:22]))
except Exception as e: print('$label: failed', e)"
}
ARGS="--stencil 7 --grid 256"
run "7-pt 256^3 control wave tg 1" HIPX_TMPL_TG=1
run "7-pt 256^3 control wave tg 2" HIPX_TMPL_TG=2
run "7-pt 256^3 control wave tg 4" HIPX_TMPL_TG=4
run "7-pt 256^3 four waves       " HIPX_TMPL_CTRL=0
ARGS="--stencil 7 --grid 512"
run "7-pt 512^3 control wave tg 2" HIPX_TMPL_TG=2
run "7-pt 512^3 control wave tg 4" HIPX_TMPL_TG=4
run "7-pt 512^3 four waves       " HIPX_TMPL_CTRL=0
ARGS="--stencil 27 --grid 256"
run "27-pt 256^3 control wave tg 2" HIPX_TMPL_TG=2
run "27-pt 256^3 four waves       " HIPX_TMPL_CTRL=0
echo "total ${SECONDS}s"
