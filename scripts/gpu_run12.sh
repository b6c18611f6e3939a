#!/bin/bash
# MPI-rank plugin tests (ranks share the one GPU)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
which mpiexec; ls /opt/conda/bin/mpiexec
timeout 900 python -m pytest tests/test_gpu_plugin_mpi.py -q -m gpu -x 2>&1 | tail -30 | tee gpurun_out/mpi_tests.log
