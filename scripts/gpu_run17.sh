#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for w in 2 4; do
  echo "== waves/CU $w"
  SOR_ONLY=1 HIPX_SOR_WAVES_PER_CU=$w timeout 60 python scripts/gmres_sor_timing.py 192 7 2>&1 | grep -E "sweep|Error|error" | tail -1
  SOR_ONLY=1 HIPX_SOR_WAVES_PER_CU=$w timeout 60 python scripts/gmres_sor_timing.py 128 27 2>&1 | grep -E "sweep|Error|error" | tail -1
done
echo "== levels"; SOR_ONLY=1 HIPX_SOR_MODE=levels timeout 60 python scripts/gmres_sor_timing.py 128 27 2>&1 | grep -E "sweep" | tail -1
