#!/bin/bash
# Round 2, GPU call F: SOR compute wave with hand-issued LDS bursts, template SpMV with id prefetch, multi-rank (IPC) bench path.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
echo "== kernel tests"; timeout 1500 python -m pytest tests/test_gpu_mat.py tests/test_gpu_sor.py tests/test_gpu_vec.py -x -q --timeout=300 -p no:cacheprovider > "$O/r2f_kern.log" 2>&1; tail -4 "$O/r2f_kern.log" | cut -c1-300
echo "== slab proxy"; timeout 600 python scripts/config3_slab_proxy.py 2>&1 | grep -v amdgpu.ids | tee "$O/r2f_slab.log" | tail -9
HIPX_SOR_DEBUG=1 timeout 300 python scripts/config3_slab_proxy.py 2>&1 | grep "hipx sor\] strand KIND . done" | head -2 | cut -c1-400 | tee "$O/r2f_sorstats.log"
echo "== tmpl"
for cfg in 1 2; do echo "cfg $cfg: $(HIPX_TMPL_CFG=$cfg timeout 200 python bench.py --quick --steps 100 --warmup 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f it/s  spmv %.4f ms" % (1e3/d["ms_per_step"], d["roofline"]["avg_launch_ms"]))')"; done 2>&1 | tee "$O/r2f_tmpl.log"
timeout 200 python scripts/spmv_variants.py 256 7 0,25 2>&1 | grep "spmv_" | tee -a "$O/r2f_tmpl.log"
echo "== multirank + other tests"; timeout 2400 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_halo.py tests/test_gpu_plugin_mpi.py tests/test_gpu_plugin_kats.py tests/test_gpu_plugin.py tests/test_gpu_scale_parity.py -q --timeout=900 -p no:cacheprovider -rf > "$O/r2f_pytest.log" 2>&1; tail -14 "$O/r2f_pytest.log" | cut -c1-300
echo "== gmres+sor bench"; timeout 600 python bench.py --ksp gmres --pc sor --stencil 27 --grid 256 --steps 60 --warmup 5 --quick 2>/dev/null | tee "$O/r2f_bench_gmres_sor.json" | cut -c1-330
