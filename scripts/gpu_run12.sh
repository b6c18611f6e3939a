#!/bin/bash
# MPI-rank plugin tests (ranks share the one GPU) + the single-rank plugin tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_plugin_mpi.py tests/test_gpu_plugin.py tests/test_gpu_plugin_kats.py -q -m gpu 2>&1 | tail -40 | tee gpurun_out/mpi_tests.log
