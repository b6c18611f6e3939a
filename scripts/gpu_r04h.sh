#!/bin/bash
# round-4 run H: where did the GMRES(30)+SOR iteration's extra 5 ms come from?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=r04h
q() { python bench.py --quick "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%8.1f it/s  %.4f ms/it  spmv %.4f ms  %s' % (d['value'] or -1, d['ms_per_step'], r['avg_launch_ms'], r['kernel'][:24]))"; }
echo "gmres+sor 27pt 256:"; q --ksp gmres --pc sor --stencil 27 --grid 256 --steps 60 --warmup 5
echo "gmres+sor 27pt 256 march1:"; HIPX_MARCH1=1 q --ksp gmres --pc sor --stencil 27 --grid 256 --steps 60 --warmup 5
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${T}_prof -o q -- python $GRAFT_REPO_ROOT/bench.py --quick --ksp gmres --pc sor --stencil 27 --grid 256 --steps 60 --warmup 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r['Name'][:90].ljust(90), r['Calls'], '%.1f us'%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
