#!/bin/bash
# Round 2, GPU call C: SOR (all schedules), template-SpMV geometry sweep, slab proxy, whole suite, bench.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
rm -f "$O/parity_measured.json"
echo "== sor tests"; timeout 1200 python -m pytest tests/test_gpu_sor.py -x -q --timeout=300 -p no:cacheprovider > "$O/r2c_sor.log" 2>&1; tail -5 "$O/r2c_sor.log"
echo "== slab proxy"; timeout 600 python scripts/config3_slab_proxy.py > "$O/r2c_slab.log" 2>&1; grep -v amdgpu.ids "$O/r2c_slab.log" | tail -12
echo "== tmpl sweep"
for cfg in 0 1 2 3 4; do for blocks in 2048 1024; do
  echo "cfg $cfg blocks $blocks: $(HIPX_TMPL_CFG=$cfg HIPX_TMPL_BLOCKS=$blocks timeout 200 python bench.py --quick --steps 100 --warmup 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f it/s  spmv %.4f ms" % (d["steps"]/ (d["ms_per_step"]*d["steps"]/1e3), d["roofline"]["avg_launch_ms"]))')"
done; done 2>&1 | tee "$O/r2c_tmpl_sweep.log"
for cfg in 0 2 4; do echo "27pt cfg $cfg: $(HIPX_TMPL_CFG=$cfg timeout 200 python scripts/spmv_variants.py 160 27 0 2>&1 | grep spmv_tmpl)"; done 2>&1 | tee -a "$O/r2c_tmpl_sweep.log"
echo "== rest of the suite"; timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider --deselect tests/test_gpu_sor.py -rf > "$O/r2c_pytest.log" 2>&1; tail -40 "$O/r2c_pytest.log" | cut -c1-300
echo "== bench"; timeout 900 python bench.py > "$O/r2c_bench.json" 2> "$O/r2c_bench.err"; cat "$O/r2c_bench.json" | cut -c1-3000
echo "== gmres+sor bench"; timeout 600 python bench.py --ksp gmres --pc sor --stencil 27 --grid 256 --steps 60 --warmup 5 --quick > "$O/r2c_bench_gmres_sor.json" 2> "$O/r2c_bench_gmres_sor.err"; cat "$O/r2c_bench_gmres_sor.json" | cut -c1-700; tail -3 "$O/r2c_bench_gmres_sor.err"
