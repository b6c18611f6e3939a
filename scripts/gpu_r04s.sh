#!/bin/bash
# round-4 run S: single ds_read_b64 operand reads in the 27-entry march kernels (A/B against a paired-reads build in petsc_amd/lib/alt),
# and the config-4 CG + PCSOR leg on the unstructured stand-in
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
q() { python bench.py --quick "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%8.1f it/s  %.4f ms/it  product %.4f ms  %s' % (d['value'] or -1, d['ms_per_step'], r['avg_launch_ms'], r['kernel'][:24]))"; }
timeout 600 python -m pytest tests/test_gpu_mat.py -x -q -m gpu -k "march or templ or 27" 2>&1 | tail -3
for rep in 1 2; do
for g in 256 512; do
echo "27pt $g solo reads:";   q --stencil 27 --grid $g --steps 100
echo "27pt $g paired reads:"; HIPX_LIBDIR=$PWD/petsc_amd/lib/alt q --stencil 27 --grid $g --steps 100
echo "27pt $g solo, pc none (plain product + dot):";   q --stencil 27 --grid $g --steps 100 --pc none --pipeline 0
echo "27pt $g paired, pc none:"; HIPX_LIBDIR=$PWD/petsc_amd/lib/alt q --stencil 27 --grid $g --steps 100 --pc none --pipeline 0
done
done
timeout 900 python - <<'PY'
import json, sys, time
sys.path.insert(0, '.')
import torch, bench
from petsc_amd import _lib
hx = _lib.init(0)
def sync():
    _lib.chk(hx.hipxDeviceSynchronize())
t0 = time.time()
cfg = bench.config4_cfg(); cfg.pc = "sor"
r = bench.leg_matrix_solver(cfg, 30, 3, sync, torch, parity_its=5)
print("config4 cg+sor leg: %.1f s wall" % (time.time() - t0))
print(json.dumps(r)[:3000])
PY
