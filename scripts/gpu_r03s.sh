#!/bin/bash
# per-pass / per-workgroup timing of the pair kernel (HIPX_TMPL_TRACE)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
HIPX_TMPL_TRACE=1 timeout 300 python bench.py --spmv-only 8 --stencil 7 --grid 256 > gpurun_out/r03s_trace.log 2>&1
grep "tmpl trace\] XCD" gpurun_out/r03s_trace.log | head -8
grep "tmpl trace\] wg " gpurun_out/r03s_trace.log | grep passes, | head -64 | cut -c1-120
