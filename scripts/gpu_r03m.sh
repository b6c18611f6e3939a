#!/bin/bash
# round-3 run M: pair form with several ticket counters per XCD (HIPX_TMPL_NQ = 1, 2, 4, 8): parity + timing.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03m
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_mat.py tests/test_gpu_ksp.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
tail -8 gpurun_out/${T}_pytest.log | cut -c1-300
for st in 7 27; do
  for nq in 1 2 4 8; do
    HIPX_TMPL_NQ=$nq timeout 600 python bench.py --stencil $st --grid 256 --quick > gpurun_out/${T}_bench_${st}_nq${nq}.json 2> gpurun_out/${T}_bench_${st}_nq${nq}.err
    python - "$st" "$nq" "gpurun_out/${T}_bench_${st}_nq${nq}.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print("stencil %s nq %s: %.1f it/s  ms/step %.4f  spmv %.4f ms  %s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], r.get("avg_launch_ms", -1), r.get("kernel", "")[:24]))
except Exception as e:
    print("stencil %s nq %s: failed %s" % (sys.argv[1], sys.argv[2], e))
PY
  done
done
HIPX_TMPL_NOPAIR=1 timeout 600 python bench.py --stencil 7 --grid 256 --quick 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stencil 7 nopair: %.1f it/s spmv %.4f ms' % (d['value'], d['roofline']['avg_launch_ms']))"
echo "total ${SECONDS}s"
