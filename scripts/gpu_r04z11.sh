#!/bin/bash
# round-4 run Z11: the library as committed at the end (after the RUN-form experiment was taken out): smoke, the inode and SOR tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 python -m pytest tests/test_gpu_inode.py tests/test_gpu_sor.py -x -q -m gpu 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -2
