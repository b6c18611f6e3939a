#!/bin/bash
# round-4 run E: the fused update kernel's shape (stand-alone timings)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=r04e
for v in "" "HIPX_CG_FUSED_U2=1" "HIPX_CG_FUSED_BLOCKS=512" "HIPX_CG_FUSED_SWEEP=1" "HIPX_CG_FUSED_SWEEP=1 HIPX_CG_FUSED_BLOCKS=512" "HIPX_CG_FUSED_BLOCKS=128" "HIPX_RED_BLOCKS=512"; do
  echo "== $v"; env $v python scripts/cg_kernels_timing.py 2>&1 | grep -v "^$" | head -3
done
python scripts/cg_kernels_timing.py
