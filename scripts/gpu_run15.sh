#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sor.py -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/pytest15.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest15.log
tail -3 gpurun_out/pytest15.log
for w in 1 2 4 8 16; do
  echo "== waves/CU $w"; SOR_ONLY=1 HIPX_SOR_WAVES_PER_CU=$w timeout 300 python scripts/gmres_sor_timing.py 192 7 2>&1 | grep -E "sweep|Error|error" | tail -2
  SOR_ONLY=1 HIPX_SOR_WAVES_PER_CU=$w timeout 300 python scripts/gmres_sor_timing.py 128 27 2>&1 | grep -E "sweep|Error|error" | tail -2
done
echo "== levels"; SOR_ONLY=1 HIPX_SOR_MODE=levels timeout 300 python scripts/gmres_sor_timing.py 192 7 2>&1 | grep -E "sweep" | tail -1
