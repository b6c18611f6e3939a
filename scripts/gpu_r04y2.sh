#!/bin/bash
# round-4 run Y2: the MPI plugin tests after the inode state is declared at every device-matrix creation (COO path included)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_plugin_mpi.py tests/test_gpu_plugin.py tests/test_gpu_plugin_int64.py -x -q -m gpu 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -5
