#!/bin/bash
# Round 2, GPU call AF: template SpMV with adjacent row pairs (16-byte gathers).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
echo "== tmpl parity cfg 7"; HIPX_TMPL_CFG=7 timeout 600 python -m pytest tests/test_gpu_mat.py -x -q --timeout=300 -p no:cacheprovider -k "templates or stencil_spmv" 2>&1 | tail -2
one() { python bench.py --quick --steps 100 --warmup 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f it/s  spmv %.4f ms" % (1e3/d["ms_per_step"], d["roofline"]["avg_launch_ms"]))'; }
{ for c in 1 7; do echo "cfg $c: $(HIPX_TMPL_CFG=$c one)"; done; echo "cfg 7 blocks 1792: $(HIPX_TMPL_CFG=7 HIPX_TMPL_BLOCKS=1792 one)"; } 2>&1 | tee "$O/r2af_tmpl.log"
for c in 1 7; do HIPX_TMPL_CFG=$c timeout 200 python scripts/spmv_variants.py 256 7 0 2>&1 | grep "spmv_" | tee -a "$O/r2af_tmpl.log"; done
for c in 1 7; do HIPX_TMPL_CFG=$c timeout 200 python scripts/spmv_variants.py 256 27 0 2>&1 | grep "spmv_" | tee -a "$O/r2af_tmpl.log"; done
