#!/bin/bash
# a short check after a host-side change: the Mat / KSP / halo GPU tests, smoke, the timed legs of the bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mat.py tests/test_gpu_ksp.py tests/test_gpu_halo.py -m gpu -q --timeout 600 -p no:cacheprovider -rf 2>&1 | grep -E "passed|failed|FAILED|Error" | head -10 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --quick 2>/dev/null | tail -1 | cut -c1-200
