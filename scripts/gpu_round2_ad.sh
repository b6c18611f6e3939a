#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python scripts/sor_split_debug.py 24 2>&1 | grep -v amdgpu.ids | tail -14
echo "== no split"; HIPX_SOR_SPLIT=0 timeout 300 python scripts/sor_split_debug.py 24 2>&1 | grep -v amdgpu.ids | tail -14
