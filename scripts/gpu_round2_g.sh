#!/bin/bash
# Round 2, GPU call G: SOR compute wave with per-row precompute (ME entries), deterministic per-chunk dot partials, MPI cghipx.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
echo "== kernel tests"; timeout 1500 python -m pytest tests/test_gpu_sor.py tests/test_gpu_mat.py -x -q --timeout=300 -p no:cacheprovider > "$O/r2g_kern.log" 2>&1; tail -4 "$O/r2g_kern.log" | cut -c1-300
echo "== slab proxy"; timeout 600 python scripts/config3_slab_proxy.py 2>&1 | grep -v amdgpu.ids | tee "$O/r2g_slab.log" | grep SOR
HIPX_SOR_DEBUG=1 timeout 300 python scripts/config3_slab_proxy.py 2>&1 | grep "hipx sor\] strand KIND . done" | head -2 | cut -c1-400 | tee "$O/r2g_sorstats.log"
echo "== 7pt 256 sor"; HIPX_SOR_DEBUG=1 timeout 300 python bench.py --ksp gmres --pc sor --stencil 7 --grid 256 --steps 30 --warmup 3 --quick 2>&1 | grep "strand KIND . done\|^{" | tail -3 | cut -c1-400 | tee "$O/r2g_sor7.log"
echo "== tests"; timeout 2400 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_plugin_mpi.py tests/test_gpu_plugin_kats.py tests/test_gpu_scale_parity.py -q --timeout=900 -p no:cacheprovider -rf > "$O/r2g_pytest.log" 2>&1; tail -14 "$O/r2g_pytest.log" | cut -c1-300
echo "== gmres+sor bench"; timeout 600 python bench.py --ksp gmres --pc sor --stencil 27 --grid 256 --steps 60 --warmup 5 --quick 2>/dev/null | tee "$O/r2g_bench_gmres_sor.json" | cut -c1-330
