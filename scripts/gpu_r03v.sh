#!/bin/bash
# round-3 run V: pair kernel for every sub-template matrix up to 16 pairs (edges of all pairs through one vector load): tests, timing incl. 27-pt.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03v
SECONDS=0
for tg in 2; do
HIPX_TMPL_TG=$tg timeout 900 python -m pytest tests/test_gpu_mat.py -m gpu -q --timeout 600 -p no:cacheprovider -k "pair or stencil_spmv or templates or auto_variant" > gpurun_out/${T}_pytest_tg$tg.log 2>&1
echo "tg $tg pytest exit $? after ${SECONDS}s: $(tail -1 gpurun_out/${T}_pytest_tg$tg.log)"
done
HIPX_TMPL_TRACE=1 timeout 300 python bench.py --spmv-only 8 --stencil 7 --grid 256 > gpurun_out/${T}_trace.log 2>&1
grep "tmpl trace" gpurun_out/${T}_trace.log | awk '$5==8' | sed -n 2,9p | cut -c1-200
grep "tmpl trace" gpurun_out/${T}_trace.log | awk '$5==1032' | sed -n 2,7p | cut -c1-200
run() {
  local label=$1; shift
  env "$@" timeout 600 python bench.py --quick $ARGS 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label: %.1f it/s  ms/step %.4f  spmv %.4f ms  %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'][:22]))
except Exception as e: print('$label: failed', e)"
}
for g in 256; do
ARGS="--stencil 7 --grid $g"
run "7-pt $g^3 pair tg 1" HIPX_TMPL_TG=1
run "7-pt $g^3 pair tg 2" HIPX_TMPL_TG=2
run "7-pt $g^3 pair tg 4" HIPX_TMPL_TG=4
run "7-pt $g^3 no pair  " HIPX_TMPL_NOPAIR=1
done
ARGS="--stencil 27 --grid 256"
run "27-pt 256^3 pair (9 pairs, NP 12)" A=1
run "27-pt 256^3 pair tg 1          " HIPX_TMPL_TG=1
run "27-pt 256^3 no pair            " HIPX_TMPL_NOPAIR=1
ARGS="--stencil 27 --grid 512"
run "27-pt 512^3 pair               " A=1
run "27-pt 512^3 no pair            " HIPX_TMPL_NOPAIR=1
echo "total ${SECONDS}s"
