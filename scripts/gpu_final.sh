#!/bin/bash
# round-end style validation: full GPU test suite, smoke, default bench (with plugin / PMC / CPU-baseline legs), the other
# single-GPU configurations, rocprofv3 evidence.  Usage: bash scripts/gpu_final.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-final}
SECONDS=0
timeout 1800 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rf > gpurun_out/${T}_pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/${T}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/${T}_smoke.log
S0=$SECONDS
timeout 900 python bench.py > gpurun_out/${T}_bench.log 2>gpurun_out/${T}_bench.err
echo "default bench: $((SECONDS - S0)) s" >> gpurun_out/${T}_bench.err
if [ -z "$HIPX_FINAL_SHORT" ]; then
timeout 300 python bench.py --stencil 27 --grid 160 --quick > gpurun_out/${T}_bench_27.log 2>&1
timeout 300 python bench.py --grid 512 --steps 50 --warmup 5 --quick > gpurun_out/${T}_bench_7pt512.log 2>&1
fi
timeout 400 python bench.py --ksp gmres --pc sor --stencil 27 --grid 256 --steps 60 --warmup 5 --quick > gpurun_out/${T}_bench_gmres_sor27.log 2>&1
timeout 400 python bench.py --ksp gmres --pc sor --stencil 7 --grid 256 --steps 60 --warmup 5 --quick > gpurun_out/${T}_bench_gmres_sor7.log 2>&1
timeout 400 python scripts/config3_slab_proxy.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_config3_slab.log
timeout 60 scripts/diag/ta_probe > gpurun_out/${T}_ta_probe.txt 2>&1
timeout 60 scripts/diag/wave_placement > gpurun_out/${T}_wave_placement.txt 2>&1
bash scripts/gpu_profile.sh ${T} > gpurun_out/${T}_profile.log 2>&1
tail -3 gpurun_out/${T}_pytest.log; tail -2 gpurun_out/${T}_smoke.log; tail -1 gpurun_out/${T}_bench.err
for f in bench bench_27 bench_7pt512 bench_gmres_sor27 bench_gmres_sor7; do [ -f gpurun_out/${T}_$f.log ] && tail -1 gpurun_out/${T}_$f.log | cut -c1-300; done
cat gpurun_out/${T}_config3_slab.log | cut -c1-200
tail -24 gpurun_out/${T}_profile.log | cut -c1-200
