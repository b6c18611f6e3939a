#!/bin/bash
# round-4 run Q: multi-rank GMRES in exact mode (pairs through the MDot all-reduce), the multi-rank tiers
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r04q
SECONDS=0
timeout 1500 python -m pytest tests/test_gpu_bench_multi.py tests/test_gpu_multirank.py tests/test_gpu_halo.py -m gpu -q --timeout 900 -p no:cacheprovider -rf > gpurun_out/${T}_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/${T}_pytest.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/${T}_pytest.log | head
grep -E "^E  " gpurun_out/${T}_pytest.log | head -12
echo "total ${SECONDS}s"
