#!/bin/bash
# round-3 run I: row-granular coefficient staging for the 27-point class (ring 4 with two workgroups per CU vs ring 8 with one);
# SQ counters of the template SpMV kernel (and of the pattern-template kernel beside it).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03i
SECONDS=0
timeout 600 python -m pytest tests/test_gpu_sor.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "variable" > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
HIPX_SOR_VAR_RING=4 timeout 600 python -m pytest tests/test_gpu_sor.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "variable and 27" >> gpurun_out/${T}_pytest.log 2>&1
echo "pytest (ring 4) exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
tail -8 gpurun_out/${T}_pytest.log | cut -c1-400
{
echo "# ring 8"; timeout 300 python scripts/sor_var_timing.py 27 256
echo "# ring 4"; HIPX_SOR_VAR_RING=4 timeout 300 python scripts/sor_var_timing.py 27 256
echo "# slab ring 8"; timeout 300 python scripts/sor_var_timing.py 27 512 64
echo "# slab ring 4"; HIPX_SOR_VAR_RING=4 timeout 300 python scripts/sor_var_timing.py 27 512 64
} 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_timing.log
cat gpurun_out/${T}_timing.log | cut -c1-300
bash scripts/pmc_sq.sh ${T}_tmpl 0 > gpurun_out/${T}_sq_tmpl.txt 2>&1
bash scripts/pmc_sq.sh ${T}_tp 29 > gpurun_out/${T}_sq_tp.txt 2>&1
cat gpurun_out/${T}_sq_tmpl.txt gpurun_out/${T}_sq_tp.txt | cut -c1-200
echo "total ${SECONDS}s"
