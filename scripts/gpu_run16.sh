#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 90 python -m pytest tests/test_gpu_sor.py -m gpu -q --timeout 30 -p no:cacheprovider -x > gpurun_out/pytest16.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest16.log
tail -3 gpurun_out/pytest16.log
for w in 2 8; do
  echo "== waves/CU $w"; SOR_ONLY=1 HIPX_SOR_WAVES_PER_CU=$w timeout 40 python scripts/gmres_sor_timing.py 96 7 2>&1 | grep -E "sweep|Error|error" | tail -2
done
echo "== levels"; SOR_ONLY=1 HIPX_SOR_MODE=levels timeout 40 python scripts/gmres_sor_timing.py 96 7 2>&1 | grep -E "sweep" | tail -1
