#!/bin/bash
# round-4 run T: PCSOR on matrices with inodes (MatSOR_SeqAIJ_Inode on the device): parity tests, the plugin, the config-4 CG + PCSOR leg
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_inode.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_plugin.py -x -q -m gpu -k "inodes or matload" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_sor.py -x -q -m gpu 2>&1 | tail -3
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
print(json.dumps({k: r[k] for k in ("iterations_per_s", "ms_per_step", "parity", "setup_seconds", "setup_split", "sor_schedule", "roofline_sor")}))
PY
