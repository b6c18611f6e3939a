#!/bin/bash
# round-4 run G: the default bench line with the complete roofline evidence (by-kernel times, counter passes for every leg)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r04g
SECONDS=0
HIPX_BENCH_KEEP_PROFILES=$GRAFT_REPO_ROOT/gpurun_out/${T}_pmc timeout 1500 python bench.py > gpurun_out/${T}_bench.log 2>gpurun_out/${T}_bench.err
echo "default bench: rc $? ${SECONDS} s"
tail -3 gpurun_out/${T}_bench.err
tail -1 gpurun_out/${T}_bench.log | cut -c1-600
