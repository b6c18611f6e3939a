#!/bin/bash
# Round 2, GPU call H: template SpMV first-touch prefetch (on/off, distances); per-panel time stamps of the strand SOR.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
echo "== tmpl parity"; timeout 600 python -m pytest tests/test_gpu_mat.py -x -q --timeout=300 -p no:cacheprovider -k "templates or stencil_spmv" 2>&1 | tail -2
one() { python bench.py --quick --steps 100 --warmup 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f it/s  spmv %.4f ms" % (1e3/d["ms_per_step"], d["roofline"]["avg_launch_ms"]))'; }
echo "== tmpl prefetch"
{ echo "nopf: $(HIPX_TMPL_NOPF=1 one)"; echo "pf default: $(one)"
  for d in 32 64 128 512; do echo "pf dist $d: $(HIPX_TMPL_PFDIST=$d one)"; done
  echo "pf cfg2: $(HIPX_TMPL_CFG=2 one)"; echo "pf blocks 4096: $(HIPX_TMPL_BLOCKS=4096 one)"; } 2>&1 | tee "$O/r2h_tmpl.log"
timeout 200 python scripts/spmv_variants.py 256 7 0 2>&1 | grep "spmv_" | tee -a "$O/r2h_tmpl.log"
HIPX_TMPL_NOPF=1 timeout 200 python scripts/spmv_variants.py 256 7 0 2>&1 | grep "spmv_" | tee -a "$O/r2h_tmpl.log"
echo "== sor panel times"
HIPX_SOR_DEBUG=1 HIPX_SOR_DEBUG_DUMP="$O/r2h_sorpanels" timeout 300 python scripts/config3_slab_proxy.py 2>&1 | grep "hipx sor\] strand KIND . done" | head -2 | cut -c1-400 | tee "$O/r2h_sorstats.log"
ls -la "$O"/r2h_sorpanels* 2>&1 | head
