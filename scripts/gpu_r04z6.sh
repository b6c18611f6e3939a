#!/bin/bash
# round-4 run Z6: lane-per-row dependency-driven sweep with 16 instead of 8 entries per chunk (27-point rows in one chunk); full-size inode parity test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sor.py -x -q -m gpu -k "dep or levels or both_forms or unsymmetric" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -3
python - <<'PY'
import sys; sys.path.insert(0, '.')
import bench
from petsc_amd import _lib
hx = _lib.init(0); _, ks = _lib.load()
r = bench.leg_sor_arbitrary_values(hx, _lib, ks)
print('arbitrary values 27-pt 256^3: strand %.2f ms, level-ordered (lane per row, CH 16) %.2f ms, same bits %s' % (r['strand_streamed_coefficients_ms'], r['level_ordered_ms'], r['bit_identical_to_level_ordered']))
PY
timeout 900 python -m pytest tests/test_gpu_scale_parity.py -x -q -m gpu -k "config4" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -3
