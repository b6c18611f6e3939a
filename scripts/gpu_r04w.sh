#!/bin/bash
# round-4 run W: MatMult of matrices with inodes in MatMult_SeqAIJ_Inode's order; the tests it touches; the config-4 legs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_inode.py -x -q -m gpu 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_mat.py -x -q -m gpu 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_plugin.py -x -q -m gpu 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_scale_parity.py -x -q -m gpu -k "config4" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_halo.py tests/test_gpu_ksp.py tests/test_gpu_sor.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python - <<'PY'
import json, sys, time
sys.path.insert(0, '.')
import torch, bench
from petsc_amd import _lib
hx = _lib.init(0)
_, ks = _lib.load()
def sync():
    _lib.chk(hx.hipxDeviceSynchronize())
r = bench.leg_surrogate_spmv(hx, _lib)
print(json.dumps({k: r[k] for k in ("kernel", "sampled_rows_bit_identical")}), r["roofline_longrow"]["avg_launch_ms"], r["roofline_longrow"]["frac"])
import tempfile
tmp = tempfile.mkdtemp()
for pc, steps in (("jacobi", 100), ("sor", 30)):
    cfg = bench.config4_cfg(); cfg.pc = pc
    r = bench.leg_matrix_solver(cfg, steps, 3, sync, torch, parity_its=10 if pc == "jacobi" else 5, tmpdir=tmp)
    print(pc, json.dumps({k: r.get(k) for k in ("iterations_per_s", "ms_per_step", "parity", "spmv_kernel", "sor_schedule")}), r["roofline_spmv"]["avg_launch_ms"], (r.get("roofline_sor") or {}).get("avg_call_ms"))
PY
