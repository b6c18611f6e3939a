#!/bin/bash
# round-3 run L: pair form of the template SpMV (two consecutive rows per thread, 16-byte pair loads, neighbour lanes for the +-1 entries): parity + timing A/B.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03l
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_mat.py tests/test_gpu_ksp.py -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
tail -6 gpurun_out/${T}_pytest.log | cut -c1-400
for st in 7 27; do
  for nopair in 0 1; do
    if [ $nopair = 1 ]; then export HIPX_TMPL_NOPAIR=1; else unset HIPX_TMPL_NOPAIR; fi
    timeout 600 python bench.py --stencil $st --grid 256 --quick > gpurun_out/${T}_bench_${st}_nopair${nopair}.json 2> gpurun_out/${T}_bench_${st}_nopair${nopair}.err
    python - "$st" "$nopair" "gpurun_out/${T}_bench_${st}_nopair${nopair}.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print("stencil %s nopair %s: %.1f it/s  ms/step %.4f  spmv %.4f ms  parity %s %s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], r.get("avg_launch_ms", -1), d.get("parity_gate", {}).get("pass"), d.get("parity_gate", {}).get("max_rel_diff")))
except Exception as e:
    print("stencil %s nopair %s: failed %s" % (sys.argv[1], sys.argv[2], e))
PY
  done
done
unset HIPX_TMPL_NOPAIR
bash scripts/pmc_sq.sh ${T}_tmpl 0 > gpurun_out/${T}_sq_tmpl.txt 2>&1
sed -n 1,40p gpurun_out/${T}_sq_tmpl.txt | cut -c1-100
echo "total ${SECONDS}s"
