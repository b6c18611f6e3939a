#!/bin/bash
# round-3 run W2: the evidence run after the last kernel refactor -- full GPU suite, smoke, the default bench line (all legs), rocprofv3 kernel stats of the same
# command's timed legs, SOR slab stats + PMC.  Usage: bash scripts/gpu_r03e.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03w2
SECONDS=0
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rf > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/${T}_smoke.log
S0=$SECONDS
timeout 1200 python bench.py > gpurun_out/${T}_bench.log 2>gpurun_out/${T}_bench.err
echo "default bench: $((SECONDS - S0)) s" >> gpurun_out/${T}_bench.err
timeout 400 python scripts/config3_slab_proxy.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 > gpurun_out/${T}_config3_slab.log
bash scripts/gpu_profile.sh ${T} > gpurun_out/${T}_profile.log 2>&1
tail -5 gpurun_out/${T}_pytest.log | cut -c1-300; tail -2 gpurun_out/${T}_smoke.log; tail -1 gpurun_out/${T}_bench.err
tail -1 gpurun_out/${T}_bench.log | cut -c1-400
cat gpurun_out/${T}_config3_slab.log
tail -24 gpurun_out/${T}_profile.log | cut -c1-200
echo "total ${SECONDS}s"
