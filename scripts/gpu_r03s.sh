#!/bin/bash
# round-3 run S: per-pass timing of the pair kernel (HIPX_TMPL_TRACE): where do the ~7 us per chunk go?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
HIPX_TMPL_TRACE=1 timeout 300 python bench.py --spmv-only 8 --stencil 7 --grid 256 > gpurun_out/r03s_trace.log 2>&1
grep "tmpl trace" gpurun_out/r03s_trace.log | head -70 | cut -c1-200
