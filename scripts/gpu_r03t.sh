#!/bin/bash
# round-3 run T: pair kernel with the edge elements on the vector path: tests, trace, A/B.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03t
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_mat.py -m gpu -q --timeout 600 -p no:cacheprovider -k "pair or stencil_spmv or templates or auto_variant" > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
tail -4 gpurun_out/${T}_pytest.log | cut -c1-300
HIPX_TMPL_TRACE=1 timeout 300 python bench.py --spmv-only 8 --stencil 7 --grid 256 > gpurun_out/${T}_trace.log 2>&1
grep "tmpl trace" gpurun_out/${T}_trace.log | awk '$4==8' | head -12 | cut -c1-200
grep "tmpl trace" gpurun_out/${T}_trace.log | awk '$4==1032' | head -8 | cut -c1-200
run() {
  local label=$1; shift
  env "$@" timeout 600 python bench.py --quick $ARGS 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label: %.1f it/s  ms/step %.4f  spmv %.4f ms  %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'][:22]))
except Exception as e: print('$label: failed', e)"
}
ARGS="--stencil 7 --grid 256"
run "7-pt 256^3 pair form   " A=1
run "7-pt 256^3 no pair form" HIPX_TMPL_NOPAIR=1
run "7-pt 256^3 pair form   " A=1
ARGS="--stencil 7 --grid 512"
run "7-pt 512^3 pair form   " A=1
run "7-pt 512^3 no pair form" HIPX_TMPL_NOPAIR=1
echo "total ${SECONDS}s"
