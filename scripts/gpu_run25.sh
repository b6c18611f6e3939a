#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_plugin_kats.py -m gpu -q --timeout 60 -p no:cacheprovider > gpurun_out/pytest25.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest25.log
tail -40 gpurun_out/pytest25.log | cut -c1-220
