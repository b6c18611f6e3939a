#!/bin/bash
# round-3 run B: full GPU suite (incl. PetscSF hipx, SELL, pbjacobihipx, multi-rank bench), SELL-64 timings against the CSR kernels
# on the three operator classes, the template-kernel structure probes.  Usage: bash scripts/gpu_r03b.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03b
SECONDS=0
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rf > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
# SELL-64 (variant 28) vs the packed CSR kernels (23 row-parallel, 22 staged) and auto: 7-pt 256^3, 27-pt 160^3 with DISTINCT values is
# what matters for general matrices; spmv_variants uses the constant-coefficient operators (dictionary off for 22/23/28)
for cfg in "256 7" "160 27"; do
  for u in 8 4; do
    echo "=== spmv_variants $cfg HIPX_SELL_U=$u" >> gpurun_out/${T}_sell.log
    HIPX_SELL_U=$u timeout 300 python scripts/spmv_variants.py $cfg 23,22,28,0 2>&1 | grep -v amdgpu.ids | head -5 >> gpurun_out/${T}_sell.log
  done
done
echo "=== surrogate (Flan-like)" >> gpurun_out/${T}_sell.log
timeout 300 python scripts/config4_surrogate.py 0,22,23,28 2>&1 | grep -v amdgpu.ids | head -12 >> gpurun_out/${T}_sell.log
HIPX_SELL_U=4 timeout 300 python scripts/config4_surrogate.py 28 2>&1 | grep -v amdgpu.ids | grep -i "variant\|ms" | head -4 >> gpurun_out/${T}_sell.log
for p in 0 4 5 6; do
  echo "=== HIPX_TMPL_PROBE=$p" >> gpurun_out/${T}_tmpl_probe.log
  HIPX_TMPL_PROBE=$p timeout 200 python scripts/spmv_variants.py 256 7 26 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/${T}_tmpl_probe.log
done
# the general leg of the bench with SELL as the general kernel
timeout 600 python bench.py --general-variant 28 --no-other --no-plugin --no-cpu-baseline > gpurun_out/${T}_bench_sell.log 2>gpurun_out/${T}_bench_sell.err
tail -6 gpurun_out/${T}_pytest.log | cut -c1-300
cat gpurun_out/${T}_sell.log | cut -c1-220
cat gpurun_out/${T}_tmpl_probe.log | cut -c1-200
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r03b_bench_sell.log").read().strip().splitlines()[-1])
    g = d["roofline_general"]
    print("bench general leg:", g["kernel"], g["avg_launch_ms"], g["frac"], g.get("frac_counter_bytes"), g["iterations_per_s"], g.get("traffic"))
except Exception as e:
    print("bench sell leg failed", e)
PY
echo "total ${SECONDS}s"
