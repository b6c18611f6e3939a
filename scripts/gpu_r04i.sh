#!/bin/bash
# round-4 run I: the whole GPU suite + smoke + the default bench line on the current tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r04i
SECONDS=0
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rf > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
grep -E "passed|failed" gpurun_out/${T}_pytest.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/${T}_pytest.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/${T}_smoke.log
tail -2 gpurun_out/${T}_smoke.log
S0=$SECONDS
HIPX_BENCH_KEEP_PROFILES=$GRAFT_REPO_ROOT/gpurun_out/${T}_pmc timeout 1500 python bench.py > gpurun_out/${T}_bench.log 2>gpurun_out/${T}_bench.err
echo "default bench: rc $? $((SECONDS - S0)) s"
tail -1 gpurun_out/${T}_bench.log | cut -c1-300
echo "total ${SECONDS}s"
