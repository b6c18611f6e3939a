#!/bin/bash
# config-3 leg (GMRES(30)+PCSOR, 27-pt 256^3) under rocprofv3 --kernel-trace --stats with the environment given as arguments: top kernels, us per launch
R=${GRAFT_REPO_ROOT:-/root/repo}
for envs in "$@"; do
  echo "== $envs"
  rm -rf /tmp/c3p; (cd /tmp && env $envs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3p -o s -- python $R/bench.py --quick --ksp gmres --pc sor --stencil 27 --grid 256 --steps 60 --warmup 5 > /tmp/c3p.out 2>&1)
  tail -1 /tmp/c3p.out | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print('  it/s', d['value'], 'ms', d['ms_per_step'])
except Exception as e: print('  no line', e)"
  f=$(find /tmp/c3p -name "*kernel_stats.csv" | head -1)
  python - "$f" <<PY
import csv,sys
for r in list(csv.reader(open(sys.argv[1])))[1:8]:
    print('  ', r[0].replace('(anonymous namespace)::','').replace('void ','')[:64], r[1], round(float(r[3])/1e3,1),'us')
PY
done
