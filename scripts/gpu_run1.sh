#!/bin/bash
# first GPU pass: parity tests, smoke, bench variants, rocprof kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx" | head -4 > gpurun_out/device.log 2>&1
nproc >> gpurun_out/device.log; lscpu | grep -E "Model name|Socket|Core|Thread" >> gpurun_out/device.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/bench_v1.log 2>&1
timeout 300 python bench.py --steps 100 --warmup 10 --variant 2 --no-cpu-baseline > gpurun_out/bench_v2nt.log 2>&1
timeout 300 python bench.py --steps 100 --warmup 10 --fused 1 --no-cpu-baseline > gpurun_out/bench_fused.log 2>&1
timeout 300 python bench.py --steps 100 --warmup 10 --fused 1 --variant 2 --no-cpu-baseline > gpurun_out/bench_fused_nt.log 2>&1
timeout 300 python bench.py --steps 30 --warmup 5 --stencil 27 --grid 128 --no-cpu-baseline > gpurun_out/bench_27.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof1" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/rocprof1.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find gpurun_out/prof1 -name "*stats*" | head; ls -la gpurun_out/prof1/* | head -20
tail -5 gpurun_out/pytest.log; cat gpurun_out/smoke.log | tail -3; cat gpurun_out/bench_v1.log | tail -2
