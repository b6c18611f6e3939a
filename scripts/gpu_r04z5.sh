#!/bin/bash
# round-4 run Z5: strand SOR, one vs two workgroups per CU on config 3's per-rank slab (27-pt 512 x 512 x 64) and with streamed coefficients
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for w in 2 1 2 1; do echo "slab, $w WG/CU:"; HIPX_SOR_WG_PER_CU=$w python scripts/config3_slab_proxy.py 2>/dev/null | grep -i "sor\|sweep" | head -3; done
python - <<'PY'
import os, subprocess, sys
for w in ("2", "1"):
    out = subprocess.run([sys.executable, "-c", """
import sys; sys.path.insert(0, '.')
import bench
from petsc_amd import _lib
hx = _lib.init(0); _, ks = _lib.load()
r = bench.leg_sor_arbitrary_values(hx, _lib, ks)
print('arbitrary values 27-pt 256^3: strand %.2f ms' % r['strand_streamed_coefficients_ms'])
"""], env=dict(os.environ, HIPX_SOR_WG_PER_CU=w), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    print("WG/CU", w, out.strip().splitlines()[-1] if out.strip() else "no output")
PY
