#!/bin/bash
# Round 2, GPU call E: restructured SOR compute wave, queued template SpMV, wide MDot/MAXPY, COO / IPC tests, bench.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
echo "== kernel tests"; timeout 1500 python -m pytest tests/test_gpu_mat.py tests/test_gpu_vec.py tests/test_gpu_sor.py -x -q --timeout=300 -p no:cacheprovider > "$O/r2e_kern.log" 2>&1; tail -4 "$O/r2e_kern.log" | cut -c1-300
echo "== slab proxy"; timeout 600 python scripts/config3_slab_proxy.py 2>&1 | grep -v amdgpu.ids | tee "$O/r2e_slab.log" | tail -9
HIPX_SOR_DEBUG=1 timeout 300 python scripts/config3_slab_proxy.py 2>&1 | grep "hipx sor\] strand KIND . done" | head -4 | cut -c1-400 | tee "$O/r2e_sorstats.log"
echo "== tmpl sweep"
for cfg in 0 1 2; do for blocks in 2048 1024 512; do
  echo "cfg $cfg blocks $blocks: $(HIPX_TMPL_CFG=$cfg HIPX_TMPL_BLOCKS=$blocks timeout 200 python bench.py --quick --steps 100 --warmup 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f it/s  spmv %.4f ms" % (1e3/d["ms_per_step"], d["roofline"]["avg_launch_ms"]))')"
done; done 2>&1 | tee "$O/r2e_tmpl_sweep.log"
echo "== other tests"; timeout 2400 python -m pytest tests/test_gpu_halo.py tests/test_gpu_plugin_mpi.py tests/test_gpu_plugin_kats.py tests/test_gpu_plugin.py tests/test_gpu_scale_parity.py tests/test_gpu_ksp.py tests/test_gpu_vs_reference.py tests/test_gpu_fullsize.py -q --timeout=900 -p no:cacheprovider -rf > "$O/r2e_pytest.log" 2>&1; tail -12 "$O/r2e_pytest.log" | cut -c1-300
echo "== bench"; timeout 900 python bench.py > "$O/r2e_bench.json" 2> "$O/r2e_bench.err"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2e_bench.json').read())
print({k:d[k] for k in ('value','ms_per_step')}); print(json.dumps(d['parity_gate'])[:300]); print(json.dumps(d['plugin'])[:400]); print(json.dumps(d['cpu_baseline'])[:700])
r=d['roofline']; print({k:r[k] for k in ('kernel','achieved','frac','traffic','avg_launch_ms','effective_gbps')})
g=d['roofline_general']; print({k:g.get(k) for k in ('kernel','achieved','frac','avg_launch_ms','iterations_per_s','traffic','frac_counter_bytes')})
PY
echo "== gmres+sor bench"; timeout 600 python bench.py --ksp gmres --pc sor --stencil 27 --grid 256 --steps 60 --warmup 5 --quick 2>/dev/null | tee "$O/r2e_bench_gmres_sor.json" | cut -c1-400
