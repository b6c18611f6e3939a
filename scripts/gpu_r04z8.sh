#!/bin/bash
# round-4 run Z8: sanity of the last change (the fused CG-direction product refuses matrices with inodes): its tests, the headline loop
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mat.py tests/test_gpu_inode.py tests/test_gpu_exact.py -x -q -m gpu -k "prologue or inode or exact" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -3
python bench.py --quick 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline quick: %.1f it/s  %.4f ms/it  product %.4f ms  parity %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], (d.get('parity_gate') or {}).get('max_rel_diff')))"
