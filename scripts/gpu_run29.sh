#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python scripts/config3_slab_proxy.py 512 8 3 > gpurun_out/config3_slab.log 2>&1
grep -v amdgpu gpurun_out/config3_slab.log | tail -12
