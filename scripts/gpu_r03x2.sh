#!/bin/bash
# round-3 run X2: the last check of the round on the final code -- full GPU suite, smoke, the default bench line (no rocprofv3 passes: run X has them)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03x2
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rf > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
grep -E "passed|failed" gpurun_out/${T}_pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/${T}_smoke.log
tail -2 gpurun_out/${T}_smoke.log
S0=$SECONDS
timeout 1200 python bench.py > gpurun_out/${T}_bench.log 2>gpurun_out/${T}_bench.err
echo "default bench: $((SECONDS - S0)) s" >> gpurun_out/${T}_bench.err
tail -1 gpurun_out/${T}_bench.err
tail -1 gpurun_out/${T}_bench.log | cut -c1-400
echo "total ${SECONDS}s"
