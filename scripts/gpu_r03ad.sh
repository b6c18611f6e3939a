#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_mat.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "march" 2>&1 | tail -3 | cut -c1-300
run() { timeout 300 python bench.py --quick $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('$1', round(d['value'],1), round(d['ms_per_step'],4), r.get('kernel','')[:20], r.get('avg_launch_ms'))"; }
run march7 "--stencil 7"
run march27 "--stencil 27"
run march7_512 "--stencil 7 --grid 512"
HIPX_TMPL_NOMARCH=1 run pair7_512 "--stencil 7 --grid 512"
