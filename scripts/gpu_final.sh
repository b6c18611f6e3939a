#!/bin/bash
# round-end style validation: full GPU test suite, smoke, default bench (with CPU baseline), variants, rocprofv3 evidence
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-final}
timeout 900 python -m pytest tests -m gpu -q --timeout 180 -p no:cacheprovider > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/${T}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/${T}_smoke.log
timeout 600 python bench.py > gpurun_out/${T}_bench.log 2>&1
timeout 300 python bench.py --variant 23 --no-cpu-baseline > gpurun_out/${T}_bench_v23.log 2>&1
timeout 300 python bench.py --fused 0 --variant 23 --no-cpu-baseline > gpurun_out/${T}_bench_unfused_v23.log 2>&1
timeout 300 python bench.py --stencil 27 --grid 160 --no-cpu-baseline > gpurun_out/${T}_bench_27.log 2>&1
timeout 300 python bench.py --grid 512 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_7pt512.log 2>&1
bash scripts/gpu_profile.sh ${T} > gpurun_out/${T}_profile.log 2>&1
tail -3 gpurun_out/${T}_pytest.log; tail -2 gpurun_out/${T}_smoke.log
for f in bench bench_v23 bench_unfused_v23 bench_27 bench_7pt512; do tail -1 gpurun_out/${T}_$f.log | cut -c1-260; done
tail -8 gpurun_out/${T}_profile.log | cut -c1-160
