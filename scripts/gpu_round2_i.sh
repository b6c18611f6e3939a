#!/bin/bash
# Round 2, GPU call I: template SpMV with the template cached in scalar registers and grouped gathers.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
echo "== tmpl parity"; timeout 600 python -m pytest tests/test_gpu_mat.py -x -q --timeout=300 -p no:cacheprovider -k "templates or stencil_spmv" 2>&1 | tail -2
one() { python bench.py --quick --steps 100 --warmup 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f it/s  spmv %.4f ms" % (1e3/d["ms_per_step"], d["roofline"]["avg_launch_ms"]))'; }
echo "== tmpl cfgs"
{ for c in 1 5 6; do echo "cfg $c: $(HIPX_TMPL_CFG=$c one)"; done
  echo "cfg 5 blocks 1024: $(HIPX_TMPL_BLOCKS=1024 one)"; echo "cfg 5 blocks 4096: $(HIPX_TMPL_BLOCKS=4096 one)"; echo "cfg 5 nopf: $(HIPX_TMPL_NOPF=1 one)"; } 2>&1 | tee "$O/r2i_tmpl.log"
for c in 1 5 6; do HIPX_TMPL_CFG=$c timeout 200 python scripts/spmv_variants.py 256 7 0 2>&1 | grep "spmv_" | tee -a "$O/r2i_tmpl.log"; done
for c in 1 5 6; do HIPX_TMPL_CFG=$c timeout 200 python scripts/spmv_variants.py 256 27 0 2>&1 | grep "spmv_" | tee -a "$O/r2i_tmpl.log"; done
