#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
R="$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests -m gpu -q --timeout 120 -p no:cacheprovider > gpurun_out/pytest20.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest20.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke20.log 2>&1
timeout 600 python bench.py > gpurun_out/bench20.log 2>&1
timeout 300 python bench.py --fused 0 --no-cpu-baseline > gpurun_out/bench20_unfused.log 2>&1
timeout 300 python bench.py --variant 1 --no-cpu-baseline > gpurun_out/bench20_v1.log 2>&1
timeout 300 python bench.py --stencil 27 --grid 160 --no-cpu-baseline > gpurun_out/bench20_27.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof20" -o stats -- python "$R/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > "$R/gpurun_out/rocprof20.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc20_fetch" -o pmc -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/pmc20_fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc20_write" -o pmc -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/pmc20_write.log" 2>&1
cd "$R"
B=$R/oracle/_ref/bin
P="-dll_prepend $R/petsc_amd/lib/libpetschipx.so -vec_type hipx -mat_type aijhipx"
timeout 600 $B/ref_driver -stencil 7 -n 256 -ksp_type cg -pc_type jacobi -ksp_max_it 200 -ksp_rtol 1e-50 -matmult_its 50 $P > gpurun_out/plugin20_256.log 2>&1
tail -3 gpurun_out/pytest20.log; tail -1 gpurun_out/smoke20.log; for f in bench20 bench20_unfused bench20_v1 bench20_27; do tail -1 gpurun_out/$f.log | cut -c1-200; done; tail -2 gpurun_out/plugin20_256.log
