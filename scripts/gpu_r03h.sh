#!/bin/bash
# round-3 run H: COO assembly with entries travelling between ranks on the device (MPI plugin), ex123 suite 4, SOR timings incl. the
# constant-coefficient reference point.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03h
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_plugin_mpi.py -m gpu -q --timeout 600 -p no:cacheprovider -k "coo or ex123" -rf > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
tail -40 gpurun_out/${T}_pytest.log | cut -c1-600
{
timeout 300 python scripts/sor_var_timing.py 7 256
timeout 300 python scripts/sor_var_timing.py 27 256
} 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_timing.log
cat gpurun_out/${T}_timing.log | cut -c1-300
echo "total ${SECONDS}s"
